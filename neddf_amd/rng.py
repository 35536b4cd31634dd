"""Jump-ahead for torch's CPU generator (MT19937), so that ray-sharded rendering can replay the reference's uniforms.

The reference draws every stratified / importance uniform with `torch.rand` on the CPU default generator
(nerf_render.py:137, base_neural_render.py:75), chunk by chunk inside render_image (nerf_render.py:237-244), so "same
seed => same samples" makes the position of a ray's uniforms in the generator's output stream part of the contract.  A
rank that renders only a slab of the frame must therefore start its draws `skip` outputs into the stream.  Drawing and
discarding costs the same on every rank (0.5 GB of floats per 800x800 frame), i.e. it does not shrink with the number of
GPUs; this module advances the generator by an arbitrary distance in a few milliseconds instead:

  * MT19937 is a linear recurrence over GF(2) on a 19937-bit state; advancing by J steps is multiplication by t^J in
    GF(2)[t] / phi(t), phi = the characteristic polynomial of the recurrence (Haramoto, Matsumoto, Nishimura, Panneton,
    L'Ecuyer: "Efficient jump ahead for F2-linear random number generators", 2008);
  * phi is recovered once per process with Berlekamp-Massey from 2 x 19937 output bits (it is primitive, so any non-zero
    output bit stream has phi as its minimal polynomial), g(t) = t^J mod phi by square-and-multiply on Python integers
    (cached per J: a frame geometry needs two values);
  * the jumped state's words are x[J+m] = XOR_{i : g_i = 1} x[i+m], evaluated on 33 blocks generated from the current
    state (numpy), which needs no matrix and no 19937-step Horner loop.

`torch.rand` (float32, CPU) consumes exactly one 32-bit output per element (24-bit mantissa path), which is what lets a
distance in elements be a distance in generator outputs; tests/test_host.py checks both facts against torch itself.
"""
from functools import lru_cache

import numpy as np
import torch

_N, _M = 624, 397
_DEG = 19937
_UPPER, _LOWER, _MATRIX_A = np.uint32(0x80000000), np.uint32(0x7FFFFFFF), np.uint32(0x9908B0DF)


def _twist(mt: np.ndarray) -> np.ndarray:
    """Next block of 624 untempered words (the in-place regeneration of the reference implementation, vectorised in the
    three dependency-free segments it decomposes into)."""
    new = np.empty(_N, np.uint32)

    def mix(hi, lo, far):
        y = (hi & _UPPER) | (lo & _LOWER)
        return far ^ (y >> np.uint32(1)) ^ ((y & np.uint32(1)) * _MATRIX_A)

    a = _N - _M                                     # 227
    new[:a] = mix(mt[:a], mt[1:a + 1], mt[_M:])
    new[a:2 * a] = mix(mt[a:2 * a], mt[a + 1:2 * a + 1], new[:a])
    new[2 * a:_N - 1] = mix(mt[2 * a:_N - 1], mt[2 * a + 1:], new[a:_N - 1 - a])
    new[_N - 1] = mix(mt[_N - 1:], new[:1], new[_M - 1:_M])[0]
    return new


def _sequence(block: np.ndarray, n_words: int) -> np.ndarray:
    out = [block]
    while _N * len(out) < n_words:
        out.append(_twist(out[-1]))
    return np.concatenate(out)


@lru_cache(maxsize=1)
def _char_poly() -> int:
    """phi(t) as an integer (bit i = coefficient of t^i), via Berlekamp-Massey on the low bit of 2*19937 outputs."""
    seed = np.arange(1, _N + 1, dtype=np.uint32) * np.uint32(2654435761)
    bits = (_sequence(seed, 2 * _DEG + _N)[_N:_N + 2 * _DEG] & np.uint32(1)).tolist()
    c, b, L, m, window = 1, 1, 0, 1, 0          # connection polynomials as integers; window bit j = s[n-j]
    for n, s in enumerate(bits):
        window = (window << 1) | s
        if (c & window).bit_count() & 1:
            t = c
            c ^= b << m
            if 2 * L <= n:
                L, b, m = n + 1 - L, t, 1
            else:
                m += 1
        else:
            m += 1
    if L != _DEG:
        raise RuntimeError("MT19937 minimal polynomial has degree %d" % L)
    # reciprocal of the connection polynomial: phi_i = c_{L-i}
    return int(format(c, "0%db" % (L + 1))[::-1], 2)


@lru_cache(maxsize=1)
def _reduction_table():
    """tab[b] = the multiple of phi (degree < 19937 + 8) whose top byte (bits 19937..19944) is b: lets the reduction
    clear eight bits per step."""
    phi = _char_poly()
    tab = [0] * 256
    for b in range(1, 256):
        acc = 0
        for bit in range(7, -1, -1):
            if ((acc >> (_DEG + bit)) ^ (b >> bit)) & 1:
                acc ^= phi << bit
        tab[b] = acc
    return tab


def _reduce(r: int) -> int:
    tab = _reduction_table()
    while True:
        top = r.bit_length() - 1
        if top < _DEG:
            return r
        sh = max(top - 7 - _DEG, 0)             # the table's byte sits on bits DEG+sh .. DEG+sh+7, which include `top`
        r ^= tab[(r >> (_DEG + sh)) & 0xFF] << sh


@lru_cache(maxsize=64)
def jump_poly(steps: int) -> int:
    """t^steps mod phi(t)."""
    if steps < 0:
        raise ValueError("cannot jump backwards")
    r = 1
    for bit in format(steps, "b"):
        r = _reduce(int(format(r, "b"), 4))     # squaring over GF(2) = spreading the bits (binary digits read in base 4)
        if bit == "1":
            r = _reduce(r << 1)
    return r


def _jump_block(block: np.ndarray, first: int) -> np.ndarray:
    """Words x[first .. first+623] of the stream whose words x[0..623] are `block` (first >= 1)."""
    g = jump_poly(first - 1)
    taps = np.flatnonzero(np.unpackbits(np.frombuffer(g.to_bytes((_DEG + 7) // 8, "little"), np.uint8), bitorder="little"))
    seq = _sequence(block, _DEG + _N + 1)
    # x[(first-1) + m] = XOR_{i in taps} x[i + m] for m = 1..624 (exact for every bit: word m >= 1 lies inside the state)
    win = np.lib.stride_tricks.sliding_window_view(seq[1:], _N)
    out = np.zeros(_N, np.uint32)
    for lo in range(0, len(taps), 2048):        # bounded temporaries
        out ^= np.bitwise_xor.reduce(win[taps[lo:lo + 2048]], axis=0)
    return out


# torch.get_rng_state() of the CPU generator: {u64 seed; i32 left; i32 seeded; u64 next; u64 state[624]; normal cache ...}
_OFF_LEFT, _OFF_NEXT, _OFF_STATE = 8, 16, 24


def advance_state(state: torch.Tensor, n_outputs: int) -> torch.Tensor:
    """A copy of a CPU-generator state (torch.get_rng_state()) advanced by n_outputs 32-bit outputs."""
    if n_outputs < 0:
        raise ValueError("cannot advance backwards")
    raw = bytearray(state.numpy().tobytes())
    left = int(np.frombuffer(raw, np.int32, 1, _OFF_LEFT)[0])
    nxt = int(np.frombuffer(raw, np.uint64, 1, _OFF_NEXT)[0])
    block = np.frombuffer(raw, np.uint64, _N, _OFF_STATE).astype(np.uint32)
    if n_outputs == 0:
        return state.clone()
    if left == 1 and nxt != _N:
        # freshly seeded: the array holds the seed expansion and the first output regenerates it -- which is exactly
        # "624 words consumed" of a block that precedes the stream
        nxt = _N
    q = nxt + n_outputs                          # absolute index (in words from the start of `block`) of the next output
    b = (q - 1) // _N
    c = q - _N * b                               # consumed words of the target block, 1..624
    if b > 0:
        block = _jump_block(block, _N * b)
    np.frombuffer(raw, np.uint64, _N, _OFF_STATE)[:] = block.astype(np.uint64)
    np.frombuffer(raw, np.int32, 1, _OFF_LEFT)[0] = _N + 1 - c
    np.frombuffer(raw, np.uint64, 1, _OFF_NEXT)[0] = c
    return torch.frombuffer(raw, dtype=torch.uint8).clone()


def skip_uniforms(n: int, generator: torch.Generator = None) -> None:
    """Advance the CPU generator as if `torch.rand(n)` (float32) had been drawn and discarded."""
    if n <= 0:
        return
    gen = torch.default_generator if generator is None else generator
    gen.set_state(advance_state(gen.get_state(), n))
