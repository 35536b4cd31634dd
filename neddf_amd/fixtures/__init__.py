"""Fixtures that ship with the product: the reference's pretrained bunny_smoke network (the weights of
pretrained/bunny_smoke/models/model_02000.pth as arrays -- data, written by tests/golden/gen_goldens.py -- and the
network section of its frozen .hydra/config.yaml) and the deterministic synthetic-weight generator.  bench.py,
__graft_entry__.smoke() and the tools measure on these; the tests use the same files."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUNNY_SMOKE_WEIGHTS = os.path.join(HERE, "bunny_smoke_weights.npz")

# pretrained/bunny_smoke/.hydra/config.yaml: network (keywords of neddf.network.NeDDF, neddf.py:52-66)
BUNNY_SMOKE_CFG = dict(embed_pos_rank=10, embed_dir_rank=4, ddf_layer_count=8, ddf_layer_width=256,
                       col_layer_count=4, col_layer_width=256, d_near=0.001, activation_type="tanhExp",
                       density_activation_type="LeakyReLU", lowpass_alpha_offset=10, skips=[4],
                       penalty_weight={"constraints_aux_grad": 0.05, "constraints_dDdt": 0.5,
                                       "constraints_color": 0.0001, "range_distance": 1.0,
                                       "range_aux_grad": 1.0, "range_color": 0.1})
# ... render (keywords of neddf.render.NeRFRender, nerf_render.py:40-50)
BUNNY_SMOKE_RENDER = dict(sample_coarse=64, sample_fine=128, dist_near=2.0, dist_far=6.0, max_dist=6.0,
                          use_coarse_network=False, sampling_type="cone")


def bunny_smoke_weights():
    """{state-dict key: float32 array} of the shipped network (26 tensors, 646 661 parameters)."""
    d = np.load(BUNNY_SMOKE_WEIGHTS)
    return {k: d[k] for k in d.files}
