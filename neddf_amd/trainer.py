"""Evaluation half of the reference trainer -- host-side mirror of
neddf/trainer/{base_trainer,nerf_trainer}.py restricted to what
neddf/scripts/run_eval.py needs: construct from the frozen run config, load a
checkpoint, render every test view, write the PNGs, print PSNR/SSIM.
Training (losses, optimiser, backward) is out of scope and
raises.
"""
from pathlib import Path
from typing import Any, List

import numpy as np
import torch

from .camera import Camera, PinholeCalib
from .config import instantiate, to_plain
from .dataset import imwrite_bgr
from .metrics import peak_signal_noise_ratio, structural_similarity


def _get(cfg: Any, key: str) -> Any:
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


class BaseTrainer:
    """base_trainer.py:50-113 constructor keywords."""

    def __init__(self, global_config: Any, device: str = "cuda:0", batch_size: int = 1024, chunk: int = 1024,
                 epoch_max: int = 2000, epoch_save_fields: int = 2, epoch_test_rendering: int = 10,
                 epoch_save_model: int = 100, scheduler_lr: float = 0.99815, optimizer_lr: float = 0.0005,
                 optimizer_weight_decay: float = 0.0) -> None:
        self.config = global_config
        self.device = torch.device(device)
        self.batch_size, self.chunk = batch_size, chunk
        self.epoch_max, self.epoch_save_fields = epoch_max, epoch_save_fields
        self.epoch_test_rendering, self.epoch_save_model = epoch_test_rendering, epoch_save_model
        self.scheduler_lr, self.optimizer_lr, self.optimizer_weight_decay = scheduler_lr, optimizer_lr, optimizer_weight_decay
        self.dataset = instantiate(_get(self.config, "dataset"))
        self.camera_calib = PinholeCalib(self.dataset[0]["camera_calib_params"]).to(self.device)
        self.cameras: List[Camera] = [Camera(self.camera_calib, self.dataset[i]["camera_params"]).to(self.device)
                                      for i in range(len(self.dataset))]
        # loss functions are training-only (config.loss is read but never instantiated here)

    def load_pretrained_model(self, model_path: Path) -> None:
        """base_trainer.py:115-121"""
        self.neural_render.load_state_dict(torch.load(str(model_path), map_location="cpu"))

    def render_test(self, output_dir: Path, camera_id: int, downsampling: int = 1) -> None:
        """base_trainer.py:123-174: colour -> clamp(c*255) uint8, depth -> clamp((d-2)/4*50000/256) uint8,
        PNGs {id:03}_rgb / _rgb_gt / _depth, PSNR + SSIM vs the ground truth at full resolution."""
        rgb_gt = self.dataset[camera_id]["rgb_images"].astype(np.uint8)
        camera = self.cameras[camera_id]
        camera.update_transform()
        h, w = rgb_gt.shape[0], rgb_gt.shape[1]
        images = self.neural_render.render_image(w, h, camera, ["color", "depth"], downsampling, self.chunk)
        rgb_np = torch.clamp(images["color"] * 255, 0, 255).detach().cpu().numpy().astype(np.uint8)
        depth_np = torch.clamp((images["depth"] - 2.0) / 4.0 * 50000 / 256, 0, 255).detach().cpu().numpy().astype(np.uint8)
        output_dir = Path(output_dir)
        imwrite_bgr(output_dir / "{:03}_rgb.png".format(camera_id), rgb_np)
        imwrite_bgr(output_dir / "{:03}_rgb_gt.png".format(camera_id), rgb_gt)
        imwrite_bgr(output_dir / "{:03}_depth.png".format(camera_id), depth_np)
        if downsampling == 1:
            psnr = peak_signal_noise_ratio(rgb_np, rgb_gt)
            ssim = structural_similarity(rgb_np, rgb_gt, channel_axis=2)
            print("psnr: {}, ssim: {}".format(psnr, ssim))
            self.last_metrics = (psnr, ssim)

    def render_all(self, output_dir: Path) -> None:
        """base_trainer.py:176-188"""
        self.neural_render.set_iter(-1)
        for camera_id in range(len(self.dataset)):
            print("rendering from camera {}".format(camera_id))
            self.render_test(output_dir, camera_id, 1)

    def run_train(self) -> None:
        raise NotImplementedError("training is outside the accelerated path (DESIGN.md section 8)")


class NeRFTrainer(BaseTrainer):
    """nerf_trainer.py:23-45 (renderer construction; optimiser/scheduler/logger are training-only)."""

    def __init__(self, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.neural_render = instantiate(_get(self.config, "render"), network_config=to_plain(_get(self.config, "network")),
                                         _recursive_=False).to(self.device)
