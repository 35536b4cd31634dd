"""Pinhole camera with an SE(3) pose -- host-side mirror of neddf/camera/.

Pose algebra (camera.py:66-118) stays on the host: it is a handful of 3x3
products per view.  Pixel -> ray (camera.py:155-187, pinhole_calib.py:51-74)
runs in the raygen HIP kernel.
"""
from typing import Optional

import numpy as np
import torch
from numpy import ndarray
from scipy.spatial.transform import Rotation
from torch import Tensor, nn

from ._lib import CameraDesc, Context
from .ray import Ray


class BaseCameraCalib(nn.Module):
    def __init__(self, calib_param: ndarray) -> None:
        super().__init__()
        self.params = nn.Parameter(torch.from_numpy(np.asarray(calib_param)).to(torch.float32))

    @property
    def device(self) -> torch.device:
        return self.params.device


class PinholeCalib(BaseCameraCalib):
    """Intrinsics [fx, fy, cx, cy] (pinhole_calib.py:8)."""

    def __init__(self, calib_param: ndarray) -> None:
        assert np.asarray(calib_param).shape == (4,)
        super().__init__(calib_param)

    fx = property(lambda self: self.params[0])
    fy = property(lambda self: self.params[1])
    cx = property(lambda self: self.params[2])
    cy = property(lambda self: self.params[3])

    def project_local(self, xyz: Tensor) -> Tensor:
        """camera-frame (right-up-back) points -> pixels (pinhole_calib.py:26-49)."""
        zi = torch.reciprocal(-xyz[:, 2])
        return torch.stack([self.fx * xyz[:, 0] * zi + self.cx, self.fy * (-xyz[:, 1]) * zi + self.cy], 1)

    def unproject_local(self, uv: Tensor) -> Tensor:
        """pixels -> unit camera-frame directions (pinhole_calib.py:51-74)."""
        x = (1.0 / self.fx) * (uv[:, 0] - self.cx)
        y = (1.0 / self.fy) * (uv[:, 1] - self.cy)
        v = torch.stack([x, -y, -torch.ones_like(x)], 1)
        return nn.functional.normalize(v, p=2, dim=1)


def _hat(v: Tensor) -> Tensor:
    z = torch.zeros((), dtype=v.dtype, device=v.device)
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]), torch.stack([-v[1], v[0], z])])


class Camera(nn.Module):
    """camera.py:13-64.  `params` (6, zero-initialised) perturbs the initial pose."""

    def __init__(self, camera_calib: BaseCameraCalib, initial_camera_param: Optional[ndarray] = None) -> None:
        super().__init__()
        if initial_camera_param is None:
            initial_camera_param = np.zeros(6, dtype=np.float32)
        self.camera_calib = camera_calib
        self.initial_params_np = np.asarray(initial_camera_param)
        self.params = nn.Parameter(torch.zeros(6, dtype=torch.float32))
        self.R = torch.eye(3, dtype=torch.float32)
        self.T = torch.zeros(3, dtype=torch.float32)
        self.update_transform()

    @property
    def device(self) -> torch.device:
        return self.params.device

    @property
    def R0(self) -> Tensor:
        m = Rotation.from_rotvec(self.initial_params_np[:3]).as_matrix().astype(np.float32)
        return torch.from_numpy(m).to(self.device)

    @property
    def T0(self) -> Tensor:
        return torch.from_numpy(self.initial_params_np[3:6].astype(np.float32)).to(self.device)

    def update_transform(self) -> None:
        """[R|T] = exp(params) composed with the initial pose (Rodrigues; camera.py:66-118)."""
        eye = torch.eye(3, dtype=torch.float32, device=self.device)
        rot, trans = self.params[0:3], self.params[3:6]
        theta = torch.norm(rot)
        if theta > 1e-10:
            ti = 1.0 / theta
            w = _hat(ti * rot)
            ww = torch.matmul(w, w)
            c, s = torch.cos(theta), torch.sin(theta)
            Ri = eye + s * w + (1.0 - c) * ww
            Vi = eye + (1 - c) * ti * ti * w + (theta - s) * ti * ti * ti * ww
        else:
            Ri = eye + _hat(rot)
            Vi = Ri
        self.R = torch.matmul(Ri, self.R0)
        self.T = (torch.matmul(Vi, trans[:, None]) + torch.matmul(Ri, self.T0[:, None]))[:, 0]

    def project(self, pos_world: Tensor) -> Tensor:
        pc = torch.matmul(self.R.T, (pos_world[:, :, None] - self.T[None, :, None]))[:, :, 0]
        return self.camera_calib.project_local(pc)

    def unproject(self, uv: Tensor) -> Tensor:
        return torch.matmul(self.R, self.camera_calib.unproject_local(uv).T).T + self.T[None, :]

    def get_center_of_pixels(self, pixel_id: Tensor, scale: float = 1.0) -> Tensor:
        return 0.5 + scale * pixel_id.to(torch.float32)

    def descriptor(self) -> CameraDesc:
        """POD copy of (R, T, intrinsics) for the C ABI."""
        d = CameraDesc()
        d.R[:] = self.R.detach().reshape(-1).tolist()
        d.T[:] = self.T.detach().tolist()
        d.calib[:] = self.camera_calib.params.detach().tolist()
        return d

    def create_rays(self, uv: Tensor) -> Ray:
        """Pixel indices [B,2] (int64/int32/int16/float) -> rays (camera.py:155-171)."""
        ctx = Context.get(self.device)
        uv = uv.to(self.device)
        rd, ro = ctx.raygen(uv, self.descriptor())
        return Ray(rd, ro, uv)
