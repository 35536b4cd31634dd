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


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def bunny_weights():
    d = golden("bunny_weights.npz")
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def bunny_stages():
    d = golden("bunny_stages.npz")
    return {k: d[k] for k in d.files}


BUNNY_CFG = dict(embed_pos_rank=10, embed_dir_rank=4, ddf_layer_count=8, ddf_layer_width=256,
                 col_layer_count=4, col_layer_width=256, d_near=0.001, activation_type="tanhExp",
                 density_activation_type="LeakyReLU", lowpass_alpha_offset=10, skips=[4],
                 penalty_weight={"constraints_aux_grad": 0.05, "constraints_dDdt": 0.5,
                                 "constraints_color": 0.0001, "range_distance": 1.0,
                                 "range_aux_grad": 1.0, "range_color": 0.1})


def close(a, b, rtol, atol):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    return float(err.max()) <= 0.0, float(np.abs(a - b).max()), float((np.abs(a - b) / (np.abs(b) + atol)).max())


def assert_close(a, b, rtol, atol, what=""):
    ok, mabs, mrel = close(a, b, rtol, atol)
    assert ok, "%s: max abs %.3e, max rel %.3e (rtol %g atol %g)" % (what, mabs, mrel, rtol, atol)
