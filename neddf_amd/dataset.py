"""Blender-format dataset reader (transforms_{split}.json + RGBA PNGs) --
host-side mirror of neddf/dataset/{base_dataset,nerf_synthetic_dataset}.py.

I/O only; it feeds poses/intrinsics to the renderer
and ground-truth images to the PSNR/SSIM printout of the eval harness.  The
reference reads PNGs with cv2 (BGR channel order); cv2 is not available here, so
PIL is used and the channels are swapped to keep the reference's BGR convention
(the shipped checkpoints were trained against BGR targets).
"""
import json
from pathlib import Path
from typing import Dict

import numpy as np
from numpy import ndarray
from PIL import Image
from scipy.spatial.transform import Rotation
from torch.utils.data import Dataset


def imread_unchanged_bgr(path: Path) -> ndarray:
    """cv2.imread(path, IMREAD_UNCHANGED) equivalent: uint8 [h,w,4] as B,G,R,A (or [h,w,3] B,G,R)."""
    img = np.asarray(Image.open(path))
    if img.ndim == 2:
        return img
    if img.shape[2] == 4:
        return np.ascontiguousarray(img[:, :, [2, 1, 0, 3]])
    return np.ascontiguousarray(img[:, :, ::-1])


def imwrite_bgr(path: Path, img: ndarray) -> None:
    """cv2.imwrite equivalent for uint8 [h,w,3] (BGR) / [h,w,1] / [h,w] arrays."""
    if img.ndim == 3 and img.shape[2] == 1:
        img = img[:, :, 0]
    if img.ndim == 3:
        img = img[:, :, ::-1]
    Image.fromarray(np.ascontiguousarray(img)).save(path)


class BaseDataset(Dataset):
    def __init__(self, dataset_dir: str, data_split: str, use_depth: bool = False, use_mask: bool = False) -> None:
        self.dataset_dir = Path(dataset_dir)
        self.data_split = data_split
        self.camera_calib_params: ndarray = np.zeros(4)
        self.camera_params: ndarray = np.zeros((1, 6))
        self.rgb_images: ndarray = np.zeros(0)
        self.mask_images: ndarray = np.zeros(0)
        self.depth_images: ndarray = np.zeros(0)
        self.use_depth, self.use_mask = use_depth, use_mask
        self.load_data()

    def load_data(self) -> None:
        raise NotImplementedError()

    def __len__(self) -> int:
        return self.rgb_images.shape[0]

    @property
    def image_width(self) -> int:
        return self.rgb_images.shape[2]

    @property
    def image_height(self) -> int:
        return self.rgb_images.shape[1]


class NeRFSyntheticDataset(BaseDataset):
    """nerf_synthetic_dataset.py:25-84: focal from camera_angle_x (:49-50), pose as
    rotation vector + translation (:57-63), colour premultiplied by alpha/256 when use_mask (:67-75)."""

    def load_data(self) -> None:
        with open(self.dataset_dir / "transforms_{}.json".format(self.data_split)) as f:
            meta = json.load(f)
        rgb, mask, poses = [], [], []
        for frame in meta["frames"]:
            m = np.array(frame["transform_matrix"])
            p = np.zeros(6, np.float32)
            p[:3] = Rotation.from_matrix(m[:3, :3]).as_rotvec()
            p[3:] = m[:3, 3]
            poses.append(p)
            img = imread_unchanged_bgr(self.dataset_dir / (frame["file_path"] + ".png"))
            if self.use_mask:
                rgb.append((1.0 / 256) * img[:, :, 3, None].astype(np.float32) * img[:, :, :3].astype(np.float32))
                mask.append(img[:, :, 3])
            else:
                rgb.append(img[:, :, :3].astype(np.float32))
                mask.append(255 * np.ones_like(img[:, :, 0]))
        h, w = rgb[0].shape[:2]
        focal = 0.5 * w / np.tan(0.5 * float(meta["camera_angle_x"]))
        self.camera_calib_params = np.array([focal, focal, 0.5 * w, 0.5 * h])
        self.camera_params = np.stack(poses, 0)
        self.rgb_images = np.stack(rgb, 0)
        self.mask_images = np.stack(mask, 0)

    def __getitem__(self, item: int) -> Dict[str, ndarray]:
        return {"camera_calib_params": self.camera_calib_params, "camera_params": self.camera_params[item, :],
                "rgb_images": self.rgb_images[item], "mask_images": self.mask_images[item]}


class LLFFDataset(BaseDataset):
    """Forward-facing captures in the LLFF layout (`poses_bounds.npy` + `images[_<factor>]/`), the data BASELINE.json
    configs[4] names.  The reference has no such reader; this follows the published
    preprocessing of the original NeRF release for these scenes:

      * `poses_bounds.npy` rows = 3x5 camera-to-world matrix [R | t | (h, w, f)] + (near, far) depth bounds, rotation
        columns in LLFF order (down, right, back) -> (right, up, back) as every other pose in this package;
      * the scene is rescaled so that the nearest depth bound becomes 1 / bd_factor (bd_factor 0.75), which is what makes
        near plane 1.0 valid for NDC rays (NeRFRender.ray_space = "ndc");
      * poses are recentred on their average (mean position; mean viewing direction and up vector re-orthogonalised);
      * every `hold`-th view (default 8) is the test split, the rest the training split.

    Items have the keys of the other datasets: camera_calib_params [fx, fy, cx, cy], camera_params (rotation vector +
    translation), rgb_images (BGR float, 0..255), mask_images (all 255: real photographs have no alpha)."""

    def __init__(self, dataset_dir: str, data_split: str, use_depth: bool = False, use_mask: bool = False, factor: int = 1,
                 bd_factor: float = 0.75, hold: int = 8) -> None:
        self.factor, self.bd_factor, self.hold = factor, bd_factor, hold
        self.bounds: ndarray = np.zeros((0, 2))
        super().__init__(dataset_dir, data_split, use_depth, use_mask)

    @staticmethod
    def recenter(c2w: ndarray) -> ndarray:
        """[n,3,4] camera-to-world matrices expressed in the frame of their average pose."""
        center = c2w[:, :, 3].mean(0)
        back = c2w[:, :, 2].sum(0)
        back /= np.linalg.norm(back)
        up = c2w[:, :, 1].sum(0)
        right = np.cross(up, back)
        right /= np.linalg.norm(right)
        up = np.cross(back, right)
        avg = np.eye(4)
        avg[:3, :] = np.stack([right, up, back, center], 1)
        inv = np.linalg.inv(avg)
        full = np.tile(np.eye(4), (c2w.shape[0], 1, 1))
        full[:, :3, :] = c2w
        return (inv @ full)[:, :3, :]

    def load_data(self) -> None:
        arr = np.load(self.dataset_dir / "poses_bounds.npy")
        poses = arr[:, :15].reshape(-1, 3, 5).astype(np.float64)
        bounds = arr[:, 15:17].astype(np.float64)
        img_dir = self.dataset_dir / ("images" if self.factor == 1 else "images_{}".format(self.factor))
        files = sorted(f for f in img_dir.iterdir() if f.suffix.lower() in (".png", ".jpg", ".jpeg"))
        assert len(files) == poses.shape[0], "{} images for {} poses".format(len(files), poses.shape[0])
        c2w = np.concatenate([poses[:, :, 1:2], -poses[:, :, 0:1], poses[:, :, 2:4]], 2)       # (down, right, back) -> (right, up, back)
        scale = 1.0 / (bounds.min() * self.bd_factor)
        c2w[:, :, 3] *= scale
        bounds = bounds * scale
        c2w = self.recenter(c2w)
        ids = [i for i in range(len(files)) if (i % self.hold == 0) == (self.data_split == "test")]
        rgb = [imread_unchanged_bgr(files[i])[:, :, :3].astype(np.float32) for i in ids]
        h, w = rgb[0].shape[:2]
        focal = float(poses[0, 2, 4]) * w / float(poses[0, 1, 4])       # stored for the full-resolution width
        params = np.zeros((len(ids), 6), np.float32)
        for k, i in enumerate(ids):
            params[k, :3] = Rotation.from_matrix(c2w[i, :, :3]).as_rotvec()
            params[k, 3:] = c2w[i, :, 3]
        self.camera_calib_params = np.array([focal, focal, 0.5 * w, 0.5 * h])
        self.camera_params = params
        self.rgb_images = np.stack(rgb, 0)
        self.mask_images = 255 * np.ones(self.rgb_images.shape[:3], np.uint8)
        self.bounds = bounds[ids]

    def __getitem__(self, item: int) -> Dict[str, ndarray]:
        return {"camera_calib_params": self.camera_calib_params, "camera_params": self.camera_params[item, :],
                "rgb_images": self.rgb_images[item], "mask_images": self.mask_images[item]}
