"""Metrics the evaluation driver reports.

`y_psnr_per_frame`: BT.601 luma PSNR exactly as `test.py:212-227` computes it on [0,1] RGB tensors.
`calculate_psnr` / `ssim` / `calculate_ssim`: `/root/reference/utils/util.py:140-196` (SSIM: 11x11 Gaussian,
sigma 1.5, valid region) without cv2 -- the 'valid' crop [5:-5] removes every border-dependent sample, so
a plain valid-mode correlation is the same quantity.
"""
import math

import numpy as np
import torch


def rgb_to_y(x):
    """x [...,3,H,W] in [0,1] -> Y in [0,1] following test.py:212-217 literally."""
    x = x * 255.0
    y = (x[..., 0, :, :] * 65.481 + x[..., 1, :, :] * 128.553 + x[..., 2, :, :] * 24.966) / 255.0 + 16.0
    return y / 255.0


def y_psnr_per_frame(real, fake):
    """real, fake [F,3,H,W] in [0,1] -> numpy [F] of 10*log10(1/mse) on the Y channel (test.py:223-238)."""
    yr, yf = rgb_to_y(real.float()), rgb_to_y(fake.float())
    mse = ((yr - yf) ** 2).reshape(yr.shape[0], -1).mean(dim=1)
    return (10 * torch.log10(1.0 ** 2 / mse)).cpu().numpy()


def calculate_psnr(img1, img2):
    img1, img2 = img1.astype(np.float64), img2.astype(np.float64)
    mse = np.mean((img1 - img2) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))


def _gauss_window(size=11, sigma=1.5):
    ax = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    k = np.exp(-(ax ** 2) / (2 * sigma ** 2))
    k /= k.sum()
    return np.outer(k, k)


def _valid_filter(img, window):
    from numpy.lib.stride_tricks import sliding_window_view
    if img.ndim == 3:       # cv2.filter2D filters every channel of an HxWxC image (util.py:188-190 passes HxWx3)
        return np.stack([_valid_filter(img[..., c], window) for c in range(img.shape[2])], -1)
    v = sliding_window_view(img, window.shape)
    return np.einsum("ijkl,kl->ij", v, window)


def ssim(img1, img2):
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    img1, img2 = img1.astype(np.float64), img2.astype(np.float64)
    w = _gauss_window()
    mu1, mu2 = _valid_filter(img1, w), _valid_filter(img2, w)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = _valid_filter(img1 ** 2, w) - mu1_sq
    s2 = _valid_filter(img2 ** 2, w) - mu2_sq
    s12 = _valid_filter(img1 * img2, w) - mu1_mu2
    return (((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))).mean()


def calculate_ssim(img1, img2):
    if img1.shape != img2.shape:
        raise ValueError("Input images must have the same dimensions.")
    if img1.ndim == 2:
        return ssim(img1, img2)
    if img1.ndim == 3:
        if img1.shape[2] == 3:
            return np.array([ssim(img1, img2) for _ in range(3)]).mean()   # util.py:188-190 (sic)
        if img1.shape[2] == 1:
            return ssim(np.squeeze(img1), np.squeeze(img2))
    raise ValueError("Wrong input image dimensions.")
