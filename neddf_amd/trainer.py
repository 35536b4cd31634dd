"""Trainer -- host-side mirror of neddf/trainer/{base_trainer,nerf_trainer}.py: construct from the run config,
train (random pixel batches -> render_rays -> losses -> backward -> Adam, exponential LR decay per epoch, periodic
field slices / test renders / checkpoints) and evaluate (render every view, PNGs, PSNR/SSIM).  The per-step device
work -- both field evaluations, both volume integrals and their backward passes -- is HIP kernels (autograd.py);
the optimiser is torch.optim.Adam on the parameter tensors the kernels read in place.
"""
import math
from pathlib import Path
from typing import Any, Dict, List

import numpy as np
import torch

from .camera import Camera, PinholeCalib
from .config import instantiate, to_plain
from .dataset import imwrite_bgr
from .logger import ScalarLog
from .metrics import peak_signal_noise_ratio, structural_similarity
from .parallel import average_gradients, render_image_sharded


def _get(cfg: Any, key: str) -> Any:
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


def _has(cfg: Any, key: str) -> bool:
    return key in cfg if isinstance(cfg, dict) else hasattr(cfg, key)


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
        self.writes_outputs = True      # multi-rank runs: rank 0 only (scripts/run.py, scripts/run_eval.py set it)
        self.dataset = instantiate(_get(self.config, "dataset"))
        self.camera_calib = PinholeCalib(self.dataset[0]["camera_calib_params"]).to(self.device)
        self.cameras: List[Camera] = [Camera(self.camera_calib, self.dataset[i]["camera_params"]).to(self.device)
                                      for i in range(len(self.dataset))]
        loss_cfg = _get(self.config, "loss") if _has(self.config, "loss") else {"functions": []}
        self.loss_functions = [instantiate(f).to(self.device) for f in _get(loss_cfg, "functions")]     # base_trainer.py:110-113

    def load_pretrained_model(self, model_path: Path) -> None:
        """base_trainer.py:115-121"""
        self.neural_render.load_state_dict(torch.load(str(model_path), map_location="cpu"))

    def render_test(self, output_dir: Path, camera_id: int, downsampling: int = 1, sharded: Any = None) -> None:
        """base_trainer.py:123-174: colour -> clamp(c*255) uint8, depth -> clamp((d-2)/4*50000/256) uint8,
        PNGs {id:03}_rgb / _rgb_gt / _depth, PSNR + SSIM vs the ground truth at full resolution.
        sharded (not a reference keyword): None = ray-shard the frame over the ranks whenever a process group is up (a
        COLLECTIVE: every rank must call, with the same torch seed -- run_eval.py / render_all); False = render on this rank
        alone (the periodic test render of a data-parallel training run, which only rank 0 makes)."""
        rgb_gt = self.dataset[camera_id]["rgb_images"].astype(np.uint8)
        camera = self.cameras[camera_id]
        camera.update_transform()
        h, w = rgb_gt.shape[0], rgb_gt.shape[1]
        if sharded is None:
            sharded = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        if sharded:
            # under a launcher (scripts/run_eval.py, BASELINE.json configs[3]): every rank renders a slab of the frame's pixel
            # index, one all-gather of the pixels, rank 0 writes the files.  Same seed on every rank => the single-GPU image
            images = render_image_sharded(self.neural_render, w, h, camera, ["color", "depth"], downsampling, self.chunk)
            if not self.writes_outputs:
                return
        else:
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
            if self.writes_outputs:
                print("rendering from camera {}".format(camera_id))
            self.render_test(output_dir, camera_id, 1)
        if self.writes_outputs and len(self.dataset):
            # (one line after the reference's own printout, not inside it: the "psnr: .., ssim: .." lines keep the reference's format)
            print("note: PSNR is pinned on the reference's eval harness (tests/golden/eval_harness.npz); SSIM is restated from the published "
                  "algorithm with skimage's defaults and UNPINNED (skimage is not installed here)")

    def render_field_slices(self, output_field_dir: Path, epoch: int = 0) -> None:
        """base_trainer.py:190-204"""
        images = self.neural_render.render_field_slice()
        for key in images:
            imwrite_bgr(Path(output_field_dir) / "field_{}_{:04}.png".format(key, epoch), images[key])

    def construct_ground_truth(self, camera_id: int, us_int: torch.Tensor, vs_int: torch.Tensor,
                               loss_types: List[str]) -> Dict[str, torch.Tensor]:
        """base_trainer.py:206-246: targets of the selected pixels (rgb / 256, mask / 256, zero penalty)."""
        targets: Dict[str, torch.Tensor] = {}
        us, vs = us_int.cpu().numpy().astype(np.int64), vs_int.cpu().numpy().astype(np.int64)
        if "ColorLoss" in loss_types:
            rgb = self.dataset[camera_id]["rgb_images"]
            targets["color"] = torch.from_numpy(((1.0 / 256) * rgb[vs, us, :]).astype(np.float32)).to(self.device)
        if "MaskBCELoss" in loss_types or "MaskMSELoss" in loss_types:
            mask = self.dataset[camera_id]["mask_images"]
            targets["mask"] = torch.from_numpy(((1.0 / 256) * mask[vs, us]).astype(np.float32)).to(self.device)
        if "FieldsConstraintLoss" in loss_types:
            targets["fields_penalty"] = torch.zeros(us_int.shape, dtype=torch.float32)
        return targets

    def run_train(self) -> None:
        raise NotImplementedError()

    def run_train_step(self, camera_id: int) -> float:
        raise NotImplementedError()


class NeRFTrainer(BaseTrainer):
    """nerf_trainer.py:23-140"""

    def __init__(self, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.neural_render = instantiate(_get(self.config, "render"), network_config=to_plain(_get(self.config, "network")),
                                         _recursive_=False).to(self.device)
        if getattr(self.neural_render, "ray_space", "world") == "ndc":      # forward-facing data: NDC of the full image
            self.neural_render.ndc_width, self.neural_render.ndc_height = self.dataset.image_width, self.dataset.image_height
        self.optimizer = torch.optim.Adam(self.neural_render.get_parameters_list(), lr=self.optimizer_lr,
                                          weight_decay=self.optimizer_weight_decay)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=self.scheduler_lr)
        self.logger = None      # created by the first training step: evaluation runs leave no ./log behind

    def run_train(self) -> None:
        """nerf_trainer.py:47-79: epochs over a random permutation of the frames, outputs under the working directory
        (models/, render/)."""
        if self.writes_outputs:
            Path("models").mkdir(parents=True)
        render_dir = Path("render")
        frame_length = len(self.dataset)
        self.neural_render.set_iter(0)
        for epoch in range(0, self.epoch_max + 1):
            print("epoch: ", epoch)
            camera_ids = np.random.permutation(frame_length)
            for camera_id in camera_ids:
                self.run_train_step(int(camera_id))
                self.neural_render.next_iter()
            self.scheduler.step()
            if not self.writes_outputs:
                continue
            if epoch % self.epoch_save_fields == 0:
                output_field_dir = render_dir / "fields"
                output_field_dir.mkdir(parents=True, exist_ok=True)
                self.render_field_slices(output_field_dir, epoch)
            if epoch % self.epoch_test_rendering == 0:
                print("test rendering...")
                output_dir = render_dir / "{:04}".format(epoch)
                output_dir.mkdir(parents=True)
                # rank 0 alone: the other ranks are already in the next epoch's gradient all-reduce, and their generators
                # hold other seeds -- a sharded (collective) render here would mismatch collectives and deadlock
                self.render_test(output_dir, int(camera_ids[0]), downsampling=3, sharded=False)
            if epoch % self.epoch_save_model == 0:
                torch.save(self.neural_render.state_dict(), "models/model_{:0=5}.pth".format(epoch))

    def run_train_step(self, camera_id: int) -> float:
        """nerf_trainer.py:81-140; RNG draw order: u pixels, v pixels, then render_rays' own draws."""
        if self.logger is None:
            self.logger = ScalarLog() if self.writes_outputs else ScalarLog(sink="null")
        camera = self.cameras[camera_id]
        camera.update_transform()
        h, w = self.dataset[camera_id]["rgb_images"].shape[:2]
        with self.logger.step() as rec:
            self.optimizer.zero_grad()
            us_int = (torch.rand(self.batch_size) * (w - 1)).to(torch.int16).to(self.device)
            vs_int = (torch.rand(self.batch_size) * (h - 1)).to(torch.int16).to(self.device)
            names = [type(f).__name__ for f in self.loss_functions]
            targets = self.construct_ground_truth(camera_id, us_int, vs_int, names)
            with torch.enable_grad():
                rendered = self.neural_render.render_rays(torch.stack([us_int, vs_int], 1), camera)
                terms: Dict[str, torch.Tensor] = {}
                for f in self.loss_functions:
                    terms.update(f(rendered, targets))
                total = torch.stack(list(terms.values())).sum()
                total.backward()
            # data-parallel runs (scripts/run.py under torchrun: per-rank seed and device): one all-reduce of the gradients
            average_gradients(self.neural_render.get_parameters_list())
            self.optimizer.step()
            mse = float(torch.mean(torch.square(rendered["color"].detach() - targets["color"])).item())
            loss_value = float(total.item())
            rec.report(loss_value, 10 * math.log10(1.0 / mse), terms)
        return loss_value
