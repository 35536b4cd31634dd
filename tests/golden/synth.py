"""Deterministic synthetic weights shared by the golden generator and the tests: the generator itself lives with the product's
fixtures (neddf_amd/fixtures/synth.py -- bench.py uses it for its non-default hidden widths); this name keeps `import synth`
working for the generator script and the test modules."""
from neddf_amd.fixtures.synth import *  # noqa: F401,F403
from neddf_amd.fixtures.synth import neddf_state, nerf_state, neus_state, random_sampling  # noqa: F401
