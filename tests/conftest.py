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


@pytest.fixture(autouse=True)
def _bounds_probe_after_each_test():
    """Under NEDDF_GUARD=1 (the bounds probe that stands in for a GPU-side sanitizer) every test is followed by a check of every
    poisoned band of every context: `NEDDF_GUARD=1 python -m pytest tests -m gpu` turns the whole suite into an out-of-bounds sweep."""
    yield
    if os.environ.get("NEDDF_GUARD", "0") != "1":
        return
    from neddf_amd._lib import Context
    for ctx in list(Context._instances.values()):
        bands, bad = ctx.check_guards()
        assert bad == 0, "NEDDF_GUARD: %d byte(s) written into %d guard bands" % (bad, bands)


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
