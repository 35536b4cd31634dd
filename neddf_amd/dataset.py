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
