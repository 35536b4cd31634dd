#!/usr/bin/env python3
"""Generator of tools/bin/issue_ubench.hip: hand-placed instruction streams that settle how much vector / LDS work issues
in the shadow of the matrix pipe on gfx950.

Why a generator: the question is about ISSUE ORDER, and hipcc reorders anything it is given as C++ (the round-1/2 probes,
tools/overlap_ubench.hip and tools/shadow_ubench.hip, went through `sched_group_barrier` and disagree with the hardware
guide from K = 2 fillers per gap on).  Every stream here is ONE `asm volatile` block per loop iteration, so the order in
the binary is the order written below.

Variants (all: 256 workgroups = one per CU, each wave loops `iters` times over a body of 8 MFMAs):
  same-wave    K fillers after every MFMA of the wave's own stream, hand-interleaved
  clumped      the same instructions, 8 MFMAs first, then the 8 K fillers (what a scheduler that sinks the fillers produces)
  partner      waves 0..3 run the bare MFMA stream, waves 4..7 (the second wave of every SIMD) run ONLY the fillers,
               with s_setprio (matrix wave, filler wave) swept
  alone        the filler stream of `partner` with the matrix waves idle (what the fillers cost by themselves)
Filler kinds: fma (v_fma_f32, independent), fmadep (one dependent chain per 4 fillers), exp (v_exp_f32), rcp, mix (the fused
kernels' tanhExp: 11 instructions per element, 3 of them transcendental, 16 elements in flight so that consecutive fillers are
independent), mixdep (the same, ONE element at a time: every filler waits for the previous one), cvt (v_cvt_pk_bf16_f32),
dsw16 / dsw32 / dsw64 (LDS stores), and eplg (the bf16 forward epilogue's per-element mix: tanhExp + 1/2 cvt + ds_write_b16).
Matrix instructions: f32 (v_mfma_f32_32x32x2_f32, 64 cycles), bf16 (v_mfma_f32_32x32x16_bf16, 32 cycles), f16 likewise.

  python tools/issue_ubench_gen.py && hipcc --offload-arch=gfx950 -O3 tools/bin/issue_ubench.hip -o tools/bin/issue_ubench
  tools/bin/issue_ubench > profiles/r04_issue_ubench.txt        (on the GPU box)
"""
import os

MFMA = {
    "f32": ("v_mfma_f32_32x32x2_f32", "a32", "b32", 64),
    "bf16": ("v_mfma_f32_32x32x16_bf16", "a16", "b16", 32),
    "f16": ("v_mfma_f32_32x32x16_f16", "a16", "b16", 32),
}
NACC = 8
NX = 16
NMIX = 8         # elements of the tanhExp mix in flight: consecutive fillers of one element are 8 fillers apart


class Fill:
    """emits filler j of a body; element registers: z (never written), u, t, w"""

    def __init__(self, kind):
        self.kind = kind
        self.j = 0

    def one(self):
        j = self.j
        self.j += 1
        k = self.kind
        e = j % NX
        if k == "fma":
            return f"v_fma_f32 %[u{e}], %[z{e}], %[k1], %[k2]"
        if k == "fmadep":           # chains of length 4: filler j depends on filler j-1 unless j % 4 == 0
            c = (j // 4) % NX
            return f"v_fma_f32 %[u{c}], %[u{c}], %[k1], %[k2]"
        if k == "exp":
            return f"v_exp_f32 %[u{e}], %[z{e}]"
        if k == "rcp":
            return f"v_rcp_f32 %[u{e}], %[z{e}]"
        if k == "cvt":
            return f"v_cvt_pk_bf16_f32 %[u{e}], %[z{e}], %[z{(e + 1) % NX}]"
        if k in ("dsw16", "dsw32", "dsw64"):
            off = (j % 32) * 528
            if k == "dsw16":
                return f"ds_write_b16 %[la], %[z{e}] offset:{off}"
            if k == "dsw32":
                return f"ds_write_b32 %[la], %[z{e}] offset:{off}"
            return f"ds_write_b64 %[la8], %[zz] offset:{off}"
        if k == "mixplain":          # the mix with its three transcendentals replaced by plain multiplies: same dependences, no quarter-rate unit
            seq = ["v_mul_f32 %[u{e}], %[z{e}], %[kl]", "v_mul_f32 %[u{e}], %[u{e}], %[k1]", "v_mul_f32 %[t{e}], %[u{e}], %[k2l]", "v_mul_f32 %[t{e}], %[t{e}], %[k1]",
                   "v_add_f32 %[t{e}], 1.0, %[t{e}]", "v_mul_f32 %[t{e}], %[t{e}], %[k1]", "v_fma_f32 %[t{e}], %[t{e}], -2.0, 1.0", "v_mul_f32 %[w{e}], %[z{e}], %[u{e}]",
                   "v_fma_f32 %[u{e}], %[t{e}], %[t{e}], -1.0", "v_fma_f32 %[w{e}], -%[w{e}], %[u{e}], %[t{e}]", "v_mul_f32 %[u{e}], %[z{e}], %[t{e}]"]
            return seq[(j // NMIX) % 11].format(e=j % NMIX)
        if k in ("mix", "mixdep", "eplg"):
            seq = [
                "v_mul_f32 %[u{e}], %[z{e}], %[kl]",
                "v_exp_f32 %[u{e}], %[u{e}]",
                "v_mul_f32 %[t{e}], %[u{e}], %[k2l]",
                "v_exp_f32 %[t{e}], %[t{e}]",
                "v_add_f32 %[t{e}], 1.0, %[t{e}]",
                "v_rcp_f32 %[t{e}], %[t{e}]",
                "v_fma_f32 %[t{e}], %[t{e}], -2.0, 1.0",
                "v_mul_f32 %[w{e}], %[z{e}], %[u{e}]",
                "v_fma_f32 %[u{e}], %[t{e}], %[t{e}], -1.0",
                "v_fma_f32 %[w{e}], -%[w{e}], %[u{e}], %[t{e}]",
                "v_mul_f32 %[u{e}], %[z{e}], %[t{e}]",
            ]
            if k == "eplg":         # + the conversions (one packed per two elements, y and y') and the 2-byte LDS store of y
                seq += ["v_cvt_pk_bf16_f32 %[u{e}], %[u{e}], %[w{e}]", "ds_write_b16 %[la], %[u{e}] offset:{off}"]
            n = len(seq)
            if k == "mixdep":
                el, st = (j // n) % NMIX, j % n
            else:
                el, st = j % NMIX, (j // NMIX) % n
            line = seq[st].format(e=el, off=(j % 32) * 528)
            if k == "mixdep" and line.startswith(("v_exp", "v_rcp")):
                line += "\\ns_nop 0"     # gfx940+: a VALU that reads a transcendental's result needs one wait state (hipcc inserts it; raw asm must)
            return line
        raise ValueError(k)


def nb_of(kind):
    """MFMAs per loop body: a multiple of NACC that makes NB * K fillers a whole number of the filler pattern's periods"""
    if kind in ("mix", "mixdep", "mixplain"):
        return NACC * 11
    if kind == "eplg":
        return NACC * 13
    return NACC * 4


def body(mf, kind, K, layout, nb):
    """layout: 'inter' (K fillers after every MFMA), 'clump' (8 MFMAs then 8K fillers), 'mfma' (MFMAs only), 'fill' (fillers only)"""
    ins, a, b, _ = MFMA[mf]
    f = Fill(kind) if kind else None
    lines = []
    for g in range(nb // NACC):
        if layout in ("inter", "clump", "mfma"):
            for i in range(NACC):
                lines.append(f"{ins} %[c{i}], %[{a}], %[{b}], %[c{i}]")
                if layout == "inter":
                    lines += [f.one() for _ in range(K)]
            if layout == "clump":
                lines += [f.one() for _ in range(NACC * K)]
        else:
            lines += [f.one() for _ in range(NACC * K)]
    if kind and kind.startswith(("dsw", "eplg")):
        lines.append("s_waitcnt lgkmcnt(0)")
    return lines


def asm_block(lines):
    txt = "\n".join(f'            "{l}\\n"' for l in lines)
    outs = [f'[c{i}] "+v"(c{i})' for i in range(NACC)]
    outs += [f'[u{e}] "+v"(u[{e}])' for e in range(NX)] + [f'[t{e}] "+v"(t[{e}])' for e in range(NX)] + [f'[w{e}] "+v"(w[{e}])' for e in range(NX)]
    ins = [f'[z{e}] "v"(z[{e}])' for e in range(NX)]
    ins += ['[a32] "v"(a32)', '[b32] "v"(b32)', '[a16] "v"(a16)', '[b16] "v"(b16)', '[k1] "v"(k1)', '[k2] "v"(k2)', '[kl] "v"(kl)', '[k2l] "v"(k2l)',
            '[la] "v"(la)', '[la8] "v"(la8)', '[zz] "v"(zz)']
    return "        asm volatile(\n" + txt + "\n            : " + ", ".join(outs) + "\n            : " + ", ".join(ins) + "\n            : \"memory\");\n"


PRE = r'''// GENERATED by tools/issue_ubench_gen.py -- do not edit
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define SETUP                                                                                         \
    __shared__ float lds[40 * 1024];                                                                  \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                       \
    f32x16 c0, c1, c2, c3, c4, c5, c6, c7;                                                            \
    for (int q = 0; q < 16; ++q) { c0[q] = 0.f; c1[q] = 0.f; c2[q] = 0.f; c3[q] = 0.f; c4[q] = 0.f; c5[q] = 0.f; c6[q] = 0.f; c7[q] = 0.f; } \
    bf16x8 a16, b16;                                                                                  \
    for (int i = 0; i < 8; ++i) { a16[i] = (__bf16)(0.01f * (float)(lane + i)); b16[i] = (__bf16)(0.02f * (float)(lane - i)); } \
    float a32 = 0.001f * lane + seed[0], b32 = 0.002f * lane;                                           \
    float z[16], u[16], t[16], w[16];                                                                 \
    for (int i = 0; i < 16; ++i) { z[i] = seed[i & 3] - 0.11f * (float)((lane + 5 * i) & 31); u[i] = 0.f; t[i] = 0.f; w[i] = 0.f; } \
    float k1 = 1.0001f, k2 = 0.5f, kl = 1.4426950f, k2l = 2.8853901f;                                  \
    unsigned la = (unsigned)(wave * 16 * 1024 + (lane & 31) * 2 + (lane >> 5) * 4 * 528), la8 = (unsigned)(wave * 16 * 1024 + lane * 8); \
    f32x2 zz = { z[0], z[1] };                                                                        \
    (void)lds;

#define FINISH                                                                                        \
    float s = 0.f;                                                                                    \
    for (int q = 0; q < 16; ++q) s += c0[q] + c1[q] + c2[q] + c3[q] + c4[q] + c5[q] + c6[q] + c7[q];  \
    for (int i = 0; i < 16; ++i) s += u[i] + t[i] + w[i];                                             \
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];                        \
    if (lane == 0) { cyc[(size_t)blockIdx.x * 8 + wave] = t1 - t0; }

'''


MAIN = r'''
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv)
{
    const char *filter = argc > 1 ? argv[1] : "";
    float *out, *seed;
    unsigned long long *cyc;
    CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    CK(hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long)));
    CK(hipMalloc(&seed, 16));
    const float hseed[4] = { 1.0f, 0.5f, -0.25f, 0.125f };
    CK(hipMemcpy(seed, hseed, 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# tools/issue_ubench_gen.py: 256 workgroups (one per CU), 4 waves (same-wave / clumped) or 8 waves (partner: waves 4..7 = the second wave of each SIMD)\n");
    printf("# cyc/MFMA  = median over the matrix waves of (s_memtime span / MFMAs issued); for 'filler-wave-alone' the span of the filler waves / MFMA-equivalents\n");
    printf("# extra     = cyc/MFMA above the bare matrix stream of the same body size; perfill = extra / K\n");
    printf("# fill_cyc  = partner mode: the FILLER waves' span per MFMA-equivalent (K fillers); wall = best hipEvent time of 3 launches\n");
    printf("%-22s %-5s %-7s %2s %-18s %9s %8s %8s %9s %9s  %s\n", "kernel", "mfma", "filler", "K", "mode", "cyc/MFMA", "extra", "perfill", "fill_cyc", "wall_ms", "note");
    double base[3][512] = {};
    for (size_t vi = 0; vi < sizeof(table) / sizeof(table[0]); ++vi) {
        V &v = table[vi];
        if (*filter && !strstr(v.name, filter) && !strstr(v.name, "base")) continue;
        const int iters = 524288 / v.cyc / v.nb;       // 524288 matrix cycles per wave
        float best = 1e30f;
        std::vector<unsigned long long> h(256 * 8);
        double med = 0, medf = 0;
        const bool alone = strstr(v.mode, "alone") != nullptr, partner = !strcmp(v.mode, "partner");
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(cyc, 0, 256 * 8 * sizeof(unsigned long long)));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(v.fn, dim3(256), dim3(v.block), 0, 0, iters, seed, cyc, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
            std::vector<double> sm, sf;
            for (int b = 0; b < 256; ++b)
                for (int wv = 0; wv < v.block / 64; ++wv) {
                    const bool filler_wave = (alone || partner) && wv >= 4;
                    (filler_wave ? sf : sm).push_back((double)h[b * 8 + wv]);
                }
            std::sort(sm.begin(), sm.end()); std::sort(sf.begin(), sf.end());
            med = sm[sm.size() / 2] / ((double)iters * v.nb);
            medf = sf.empty() ? 0.0 : sf[sf.size() / 2] / ((double)iters * v.nb);
        }
        const int mi = !strcmp(v.mf, "f32") ? 0 : !strcmp(v.mf, "bf16") ? 1 : 2;
        if (v.K == 0 && v.block == 256) base[mi][v.nb] = med;
        const double b0 = base[mi][v.nb];
        if (alone) { med = medf; medf = 0; }
        const double extra = (v.K && !alone) ? med - b0 : 0.0;
        printf("%-22s %-5s %-7s %2d %-18s %9.2f %8.2f %8.2f %9.2f %9.4f  %s\n", v.name, v.mf, v.kind, v.K, v.mode, med, extra, v.K ? extra / v.K : 0.0, medf, best, v.note);
        fflush(stdout);
    }
    return 0;
}
'''


def kernel(name, block, role_bodies, prio):
    """role_bodies: list of (condition, lines, prio) executed by the waves matching `condition`"""
    s = f"__global__ __launch_bounds__({block}) void {name}(int iters, const float *seed, unsigned long long *cyc, float *out)\n{{\n    SETUP\n"
    s += "    unsigned long long t0 = 0, t1 = 0;\n"
    first = True
    for cond, lines, pr in role_bodies:
        s += f"    {'if' if first else 'else if'} ({cond}) {{\n"
        first = False
        if pr:
            s += f'        asm volatile("s_setprio {pr}");\n'
        s += "        __syncthreads();\n        t0 = __builtin_readcyclecounter();\n"
        if lines:
            s += "        for (int it = 0; it < iters; ++it) {\n" + asm_block(lines) + "        }\n"
        s += "        t1 = __builtin_readcyclecounter();\n    }\n"
    s += "    FINISH\n}\n\n"
    return s


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    os.makedirs(os.path.join(here, "bin"), exist_ok=True)
    kernels, table = [], []

    def add(name, block, roles, mf, K, kind, mode, nb, note=""):
        kernels.append(kernel(name, block, roles, 0))
        table.append((name, block, mf, K, kind or "-", mode, nb, note))

    Ks = [1, 2, 3, 4, 5, 6, 8, 12]
    for mf in ("f32", "bf16", "f16"):
        for nb in sorted({nb_of(k) for k in ("fma", "mix", "eplg")}):
            add(f"k_{mf}_base_nb{nb}", 256, [("true", body(mf, None, 0, "mfma", nb), 0)], mf, 0, None, "mfma-only", nb)
        kinds = ["fma", "fmadep", "exp", "mix", "mixdep", "dsw16", "dsw32", "dsw64", "cvt", "eplg"] if mf != "f16" else ["fma", "mix"]
        for kind in kinds:
            nb = nb_of(kind)
            for K in Ks:
                add(f"k_{mf}_{kind}_i{K}", 256, [("true", body(mf, kind, K, "inter", nb), 0)], mf, K, kind, "same-wave", nb)
            for K in (2, 4, 8):
                if kind in ("fma", "mix", "eplg"):
                    add(f"k_{mf}_{kind}_c{K}", 256, [("true", body(mf, kind, K, "clump", nb), 0)], mf, K, kind, "clumped", nb)
        # two waves per SIMD: matrix wave + filler wave
        if mf == "f16":
            continue
        for kind in ("fma", "exp", "mix", "eplg"):
            nb = nb_of(kind)
            for K in (2, 4, 8, 12):
                add(f"k_{mf}_{kind}_alone{K}", 512, [("wave >= 4", body(mf, kind, K, "fill", nb), 0), ("true", [], 0)], mf, K, kind, "filler-wave-alone", nb)
                for pm, pv in ((0, 0), (1, 0), (0, 1), (3, 0), (0, 3)):
                    if (pm, pv) != (0, 0) and K not in (4, 8):
                        continue
                    add(f"k_{mf}_{kind}_p{K}_{pm}{pv}", 512, [("wave >= 4", body(mf, kind, K, "fill", nb), pv), ("true", body(mf, None, 0, "mfma", nb), pm)],
                        mf, K, kind, "partner", nb, f"prio(matrix,filler)=({pm},{pv})")
        # two FILLER waves per SIMD, no matrix work: does the vector pipe take instructions from both at the single-wave rate?
        if mf == "bf16":
            add("k_valu_fma_beside_exp", 512, [("wave >= 4", body(mf, "exp", 8, "fill", 32), 0), ("true", body(mf, "fma", 8, "fill", 32), 0)], mf, 8, "fma|exp",
                "partner", 32, "waves 0..3 v_fma only (cyc/MFMA column), waves 4..7 v_exp only (fill_cyc column); alone: 33.2 / 67.9")
            add("k_valu_fma_beside_exp4", 512, [("wave >= 4", body(mf, "exp", 4, "fill", 32), 0), ("true", body(mf, "fma", 8, "fill", 32), 0)], mf, 8, "fma|exp4",
                "partner", 32, "as above with 4 v_exp per 8 v_fma")
            for kind in ("fma", "exp", "mix", "mixplain", "eplg", "dsw16"):
                nb = nb_of(kind)
                add(f"k_valu2_{kind}", 512, [("true", body(mf, kind, 8, "fill", nb), 0)], mf, 8, kind, "two-filler-waves", nb)
                add(f"k_valu1_{kind}", 256, [("true", body(mf, kind, 8, "fill", nb), 0)], mf, 8, kind, "one-filler-wave", nb)
        add(f"k_{mf}_base2", 512, [("true", body(mf, None, 0, "mfma", 32), 0)], mf, 0, None, "two-matrix-waves", 32)

    src = PRE + "".join(kernels)
    src += "typedef void (*kfn)(int, const float *, unsigned long long *, float *);\n"
    src += "struct V { const char *name; kfn fn; int block; const char *mf; int K; const char *kind; const char *mode; int nb; const char *note; int cyc; };\n"
    src += "static V table[] = {\n"
    for (name, block, mf, K, kind, mode, nb, note) in table:
        src += f'    {{ "{name}", {name}, {block}, "{mf}", {K}, "{kind}", "{mode}", {nb}, "{note}", {MFMA[mf][3]} }},\n'
    src += "};\n"
    src += MAIN
    path = os.path.join(here, "bin", "issue_ubench.hip")
    with open(path, "w") as f:
        f.write(src)
    print(path, len(table), "kernels")


if __name__ == "__main__":
    main()
