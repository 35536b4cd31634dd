"""Time the training step (render_rays with autograd -> losses -> backward -> Adam) at the reference's batch size on
synthetic targets.  `python tools/train_probe.py [rays] [steps]`"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import BUNNY_CFG, golden  # noqa: E402

import neddf_amd  # noqa: E402
from neddf_amd.loss import ColorLoss, FieldsConstraintLoss, MaskBCELoss  # noqa: E402


def main():
    rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    g = golden("bunny_stages.npz")
    wts = golden("bunny_weights.npz")
    cfg = dict(BUNNY_CFG, density_activation_type="ReLU", _target_="neddf.network.NeDDF")
    r = neddf_amd.NeRFRender(cfg, sample_coarse=64, sample_fine=128, dist_near=2.0, dist_far=6.0, max_dist=6.0,
                             use_coarse_network=False, sampling_type="cone")
    r.network_fine.load_state_dict({k: torch.from_numpy(wts[k]) for k in wts.files})
    r.to(dev)
    r.set_iter(1500)
    r.rng = "device"
    r.network_fine.weight_dtype = os.environ.get("NEDDF_PROBE_DTYPE", "fp32")       # "f16_split": split-fp16 GEMM operands
    cam = neddf_amd.Camera(neddf_amd.PinholeCalib(g["calib"].astype(np.float64)), None).to(dev)
    cam.R, cam.T = torch.from_numpy(g["R"]).to(dev), torch.from_numpy(g["T"]).to(dev)
    losses = [ColorLoss(1.0, 0.1), MaskBCELoss(0.05, 0.005), FieldsConstraintLoss(0.01, 0.01)]
    opt = torch.optim.Adam(r.get_parameters_list(), lr=5e-4)
    gen = torch.Generator(device="cpu").manual_seed(1)
    uv = (torch.rand(rays, 2, generator=gen) * 120 + 140).to(torch.int16).to(dev)
    target = {"color": torch.rand(rays, 3, generator=gen).to(dev), "mask": (torch.rand(rays, generator=gen) > 0.5).float().to(dev),
              "fields_penalty": torch.zeros(rays, device=dev)}

    def step():
        opt.zero_grad()
        out = r.render_rays(uv, cam)
        ld = {}
        for f in losses:
            ld.update(f(out, target))
        loss = torch.sum(torch.stack(list(ld.values())))
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pts = rays * (65 + 194)
    print("train step: %d rays, %d field points: %.2f ms/step, %.1f k rays/s, loss %.5f, peak mem %.2f GB" %
          (rays, pts, dt * 1e3, rays / dt / 1e3, float(loss), torch.cuda.max_memory_allocated() / 2**30))


if __name__ == "__main__":
    main()
