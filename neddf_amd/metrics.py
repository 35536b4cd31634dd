"""PSNR / SSIM as printed by the reference's eval loop (base_trainer.py:171-174
calls skimage.metrics; skimage is not installed here, so the two metrics are
restated with their skimage defaults for uint8 images: data_range 255, 7x7
uniform window, K1=0.01, K2=0.03, sample covariance, border of 3 px cropped)."""
import numpy as np
from scipy.ndimage import uniform_filter


def peak_signal_noise_ratio(image_true: np.ndarray, image_test: np.ndarray, data_range: float = 255.0) -> float:
    err = np.mean((image_true.astype(np.float64) - image_test.astype(np.float64)) ** 2)
    return float(10 * np.log10(data_range ** 2 / err)) if err > 0 else float("inf")


def _ssim_plane(x: np.ndarray, y: np.ndarray, data_range: float, win: int = 7) -> float:
    x, y = x.astype(np.float64), y.astype(np.float64)
    npx = win * win
    cov_norm = npx / (npx - 1.0)
    ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
    uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())


def structural_similarity(im1: np.ndarray, im2: np.ndarray, channel_axis: int = 2, data_range: float = 255.0) -> float:
    assert im1.shape == im2.shape and channel_axis == 2
    return float(np.mean([_ssim_plane(im1[:, :, c], im2[:, :, c], data_range) for c in range(im1.shape[2])]))
