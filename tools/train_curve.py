#!/usr/bin/env python3
"""End-to-end training check on a synthetic scene: a shaded sphere seen from 8 poses, NeDDF from random initialisation,
1024-ray steps with the reference's loss set, under the fp32 and the split-fp16 operand policies (same seeds).
Prints the mean loss per block of steps and the PSNR of a held-out view.

    python tools/train_curve.py [steps] [size] [neddf|nerf|neus]
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import BUNNY_CFG  # noqa: E402

ANGLE_X = 0.6911112070083618


def pose(az, el=0.35, radius=4.0):
    p = radius * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    z = p / np.linalg.norm(p)
    x = np.cross([0.0, 0.0, 1.0], z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, p
    return m


def shade(m, size):
    """RGBA uint8 image of a unit-albedo sphere (radius 0.6) with normal-coded colour."""
    f = 0.5 * size / np.tan(0.5 * ANGLE_X)
    v, u = np.mgrid[0:size, 0:size]
    d = np.stack([(u + 0.5 - size / 2) / f, -(v + 0.5 - size / 2) / f, -np.ones_like(u, float)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = d @ m[:3, :3].T
    o = m[:3, 3]
    b = d @ o
    disc = b * b - (o @ o - 0.36)
    hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0))
    n = (o + t[..., None] * d) / 0.6
    rgb = np.where(hit[..., None], 0.5 + 0.5 * n, 0.0)
    return (np.concatenate([rgb, hit[..., None].astype(float)], -1) * 255).astype(np.uint8)


def make_dataset(root, size, n=8):
    from PIL import Image
    for split, offs in (("train", 0.0), ("test", 0.4)):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames = []
        for i in range(n if split == "train" else 1):
            m = pose(2 * np.pi * i / n + offs)
            Image.fromarray(shade(m, size), "RGBA").save(os.path.join(root, split, "r_%d.png" % i))
            frames.append({"file_path": "./%s/r_%d" % (split, i), "transform_matrix": m.tolist()})
        json.dump({"camera_angle_x": ANGLE_X, "frames": frames}, open(os.path.join(root, "transforms_%s.json" % split), "w"))


def run(dtype, root, steps, size, network="neddf"):
    from neddf_amd.config import instantiate
    cfg = {"dataset": {"_target_": "neddf.dataset.NeRFSyntheticDataset", "dataset_dir": root, "data_split": "train", "use_depth": False,
                       "use_mask": True},
           "render": {"_target_": "neddf.render.NeRFRender", "sample_coarse": 64, "sample_fine": 128, "dist_near": 2.0, "dist_far": 6.0,
                      "max_dist": 6.0, "use_coarse_network": False, "sampling_type": "cone"},
           "network": dict(BUNNY_CFG, _target_="neddf.network.NeDDF", density_activation_type="ReLU"),
           "trainer": {"_target_": "neddf.trainer.NeRFTrainer", "device": "cuda:0", "batch_size": 1024, "chunk": 1024, "epoch_max": 1,
                       "epoch_save_fields": 1, "epoch_test_rendering": 1, "epoch_save_model": 1},
           "loss": {"functions": [{"_target_": "neddf.loss.ColorLoss", "weight": 1.0, "weight_coarse": 0.1},
                                  {"_target_": "neddf.loss.MaskBCELoss", "weight": 0.05, "weight_coarse": 0.005},
                                  {"_target_": "neddf.loss.FieldsConstraintLoss", "weight": 0.01, "weight_coarse": 0.01}]}}
    if network != "neddf":          # the reference's NeRF / NeuS set-up: point samples, colour + mask loss (config/render/nerf_render.yaml)
        import yaml
        cfg["network"] = yaml.safe_load(open(os.path.join(ROOT, "config", "network", network + ".yaml")))
        cfg["render"].update(sampling_type="point", use_coarse_network=network == "nerf")
        cfg["loss"]["functions"] = cfg["loss"]["functions"][:2]
    torch.manual_seed(3)
    np.random.seed(3)
    tr = instantiate(cfg["trainer"], global_config=cfg, _recursive_=False)
    tr.neural_render.network_coarse.weight_dtype = dtype
    tr.neural_render.network_fine.weight_dtype = dtype
    tr.neural_render.set_iter(0)
    block, curve, acc = max(steps // 8, 1), [], []
    for it in range(steps):
        acc.append(tr.run_train_step(it % len(tr.dataset)))
        tr.neural_render.next_iter()
        if len(acc) == block:
            curve.append(float(np.mean(acc)))
            acc = []
    # held-out view
    import neddf_amd
    from neddf_amd.dataset import NeRFSyntheticDataset
    test = NeRFSyntheticDataset(root, "test", use_depth=False, use_mask=True)
    cam = neddf_amd.Camera(tr.camera_calib, test[0]["camera_params"]).to(tr.device)
    cam.update_transform()
    torch.manual_seed(0)
    img = tr.neural_render.render_image(size, size, cam, ["color"], 1, 1024)["color"].clamp(0, 1).cpu().numpy()
    gt = test[0]["rgb_images"] / 256.0
    psnr = -10 * np.log10(np.mean((img - gt) ** 2))
    return curve, psnr


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    network = sys.argv[3] if len(sys.argv) > 3 else "neddf"        # neddf | nerf | neus
    with tempfile.TemporaryDirectory() as root:
        make_dataset(root, size)
        for dtype in ("fp32", "f16_split"):
            curve, psnr = run(dtype, root, steps, size, network)
            print("%-5s %-9s loss per block of %d steps: %s   held-out view PSNR %.2f dB" % (network, dtype, max(steps // 8, 1), " ".join("%.4f" % c for c in curve), psnr))


if __name__ == "__main__":
    main()
