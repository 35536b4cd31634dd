import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the GPU suite under `pytest -x`: the hot path's own parity tests run FIRST (stages bit-exact -> the fp32 field's per-key
# gates -> render_rays / render_image -> the BASELINE configurations C1 / C2 / C3), then the rest of the fp32 parity file, configs[4]
# (NDC + bf16) and the other operand policies, training, and only then everything that spawns processes -- variant libraries, bench.py
# runs, launchers -- so that a harness problem in a workflow test can never hide a kernel's parity result from the record.
_HEAD = [
    "test_raygen", "test_sample_coarse_bitexact", "test_sampling_bitexact_vs_oracle", "test_composite", "test_composite_edges",
    "test_resample_bitexact", "test_resample_edges", "test_resample_sort_sizes", "test_layer_ops", "test_neddf_bunny_field", "test_render_rays_end_to_end",
    "test_render_image_small", "test_c1_bunny_400x400_vs_oracle", "test_c2_single_pass_vs_oracle", "test_full_size_properties",
    "test_c3_full_frame_hierarchical_properties", "test_rays_on_random_cameras", "test_stages_on_random_inputs_vs_the_reference",
    "test_neddf_synth", "test_nerf_synth", "test_neddf_negative_bias_regime", "test_neddf_negative_bias_render_rays",
    "test_field_grid_views", "test_run_eval_end_to_end",
]
_FILES = ["test_gpu_parity.py", "test_gpu_c5.py", "test_gpu_train.py", "test_gpu_multi.py"]
# tests that start other processes, least entangled first; launcher-driven workflows (not SURVEY section 8 rows) at the very end
_SPAWNING = [
    "test_sample_points_from_rays_route_is_bit_identical", "test_library_communicator_single_rank", "test_ragged_gather_routes_on_one_rank", "test_c_client_of_the_abi", "test_forward_mode_kernels_in_subprocess",
    "test_reduced_cost_activation_against_the_branch_exact_build",
    "test_split_training_per_layer_route_in_subprocess", "test_wide_blocked_training_route_in_subprocess", "test_rccl_collectives_single_rank",
    "test_workspace_guard_bands", "test_sharded_render_is_independent_of_world_size", "test_sharded_render_eight_ranks_ragged", "test_bench_collective_path_on_one_rank",
    "test_bench_scaling_modes_agree_at_one_rank", "test_bench_preflight_fails_fast_and_readably", "test_smoke_under_asan",
    "test_bench_self_launch_two_ranks_shared_gpu", "test_run_eval_under_a_launcher_matches_single_process",
    "test_run_script_two_ranks_data_parallel",
]


def _order_key(item):
    name = item.originalname if hasattr(item, "originalname") and item.originalname else item.name.split("[")[0]
    fname = os.path.basename(str(item.fspath))
    if fname not in _FILES:
        return (0, 0, 0)                                 # CPU files keep their place (and their order: sort is stable)
    if name in _SPAWNING:
        return (3, _SPAWNING.index(name), 0)
    if fname == "test_gpu_parity.py" and name in _HEAD:
        return (1, 0, _HEAD.index(name))
    return (2, _FILES.index(fname), 0)


def pytest_collection_modifyitems(config, items):
    items.sort(key=_order_key)


@pytest.fixture(autouse=True)
def _bounds_probe_after_each_test():
    """Under NEDDF_GUARD=1 (the bounds probe that stands in for a GPU-side sanitizer) every test is followed by a check of every
    poisoned band of every context: `NEDDF_GUARD=1 python -m pytest tests -m gpu` turns the whole suite into an out-of-bounds sweep."""
    yield
    if os.environ.get("NEDDF_GUARD", "0").strip() in ("", "0"):
        return
    from neddf_amd._lib import Context, guard_mode
    if not guard_mode():          # the library's own rule: atoi(value) != 0
        return
    for ctx in list(Context._instances.values()):
        bands, bad = ctx.check_guards()
        assert bad == 0, "NEDDF_GUARD: %d byte(s) written into %d guard bands" % (bad, bands)


def free_port():
    """A TCP port nobody is bound to right now (asked of the kernel), for the rendezvous of a multi-process test: ports derived from the
    process id collide with whatever an earlier run left in TIME_WAIT -- a flake that stops a `pytest -x` run for nothing."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


# socket-level strings of the rendezvous only: nothing here may match a collective that hangs after the group is up (a bare "timed out"
# would -- a comm deadlock that clears on the second attempt must not be hidden)
_RENDEZVOUS_NOISE = ("Connection reset", "Connection refused", "connect()", "Address already in use", "Broken pipe",
                     "failed to connect", "EADDRINUSE", "store->get", "TCPStore", "rendezvous")


def run_ranks(make_cmd, world, env=None, timeout=600, attempts=2):
    """Start `world` worker processes (make_cmd(rank, port) -> argv), wait for all of them, return their outputs.  The rendezvous port comes
    from the kernel (free_port); a run that dies of the RENDEZVOUS itself -- a connection error before any assertion of the worker could
    speak -- is repeated once on a fresh port and the first attempt's output is kept in /tmp/neddf_rendezvous_flake.log: a socket-level
    hiccup of the test box is not a finding about the product, and under `pytest -x` it would hide every test behind it.  Assertion
    failures of a worker are never retried."""
    import subprocess
    outs = []
    for attempt in range(attempts):
        port = str(free_port())
        import tempfile
        import time
        logs = [tempfile.TemporaryFile() for _ in range(world)]
        procs = [subprocess.Popen(make_cmd(r, port), stdout=logs[r], stderr=subprocess.STDOUT,
                                  env=env[r] if isinstance(env, (list, tuple)) else env) for r in range(world)]      # one environment, or one per rank
        t0, died = time.time(), None
        while any(p.poll() is None for p in procs):          # a rank that died leaves its peers in a collective: they get 10 s, then go
            if died is None and any(p.poll() not in (None, 0) for p in procs):
                died = time.time()
            if time.time() - t0 > timeout or (died is not None and time.time() - died > 10.0):
                for p in procs:
                    if p.poll() is None:
                        p.kill()
            time.sleep(0.1)
        outs = []
        for f in logs:
            f.seek(0)
            outs.append(f.read().decode(errors="replace"))
            f.close()
        bad = [o for p, o in zip(procs, outs) if p.returncode != 0]
        if not bad:
            return outs
        text = "\n".join(bad)
        if attempt + 1 < attempts and "AssertionError" not in text and any(k in text for k in _RENDEZVOUS_NOISE):
            with open("/tmp/neddf_rendezvous_flake.log", "a") as f:
                f.write("---- world %d, attempt %d ----\n%s\n" % (world, attempt, text[-6000:]))
            import warnings
            # into pytest's own report (warnings summary), not only a file in /tmp: a retry is visible to whoever reads the run
            warnings.warn("run_ranks: world %d repeated once after a rendezvous failure: %s" % (world, text.strip().splitlines()[-1][:300]))
            continue
        raise AssertionError("a rank failed:\n" + text[-6000:])
    return outs


def golden(name):
    if name == "bunny_weights.npz":         # ships with the product (bench.py / smoke() measure on it): neddf_amd/fixtures
        from neddf_amd.fixtures import BUNNY_SMOKE_WEIGHTS
        return np.load(BUNNY_SMOKE_WEIGHTS)
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def bunny_weights():
    d = golden("bunny_weights.npz")
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def bunny_stages():
    d = golden("bunny_stages.npz")
    return {k: d[k] for k in d.files}


from neddf_amd.fixtures import BUNNY_SMOKE_CFG as BUNNY_CFG  # noqa: E402  (the shipped network's configuration)


def close(a, b, rtol, atol):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    return float(err.max()) <= 0.0, float(np.abs(a - b).max()), float((np.abs(a - b) / (np.abs(b) + atol)).max())


def assert_close(a, b, rtol, atol, what=""):
    ok, mabs, mrel = close(a, b, rtol, atol)
    assert ok, "%s: max abs %.3e, max rel %.3e (rtol %g atol %g)" % (what, mabs, mrel, rtol, atol)
