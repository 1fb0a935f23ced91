"""Synthetic benchmark / parity inputs: deterministic smooth + texture clean image,
Gaussian noise sigma/255 from numpy default_rng(seed), quantised to uint8 (the path's real input dtype,
demo.py:40-41).  Mirrors how the reference makes reproducible test noise
(RST/Denoising/test_gaussian_color_denoising.py:82-83)."""
import numpy as np


def synthetic_noisy_image(h, w, c=3, sigma=25.0, seed=0):
    """Synthetic input of SURVEY 8d: smooth + texture clean image, Gaussian noise sigma/255, uint8.
    Returns (clean_u8, noisy_u8)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    chans = []
    for ci in range(c):
        f1, f2 = 0.013 * (ci + 1), 0.031 * (ci + 2)
        base = 0.5 + 0.25 * np.sin(f1 * xx + 0.7 * ci) * np.cos(f2 * yy - 0.3 * ci)
        tex = 0.08 * np.sin(0.9 * xx + 1.3 * yy + ci) + 0.05 * ((xx.astype(np.int64) // 16 + yy.astype(np.int64) // 16) % 2)
        chans.append(base + tex)
    clean = np.clip(np.stack(chans, -1), 0, 1)
    clean_u8 = np.rint(clean * 255).astype(np.uint8)
    noisy = clean_u8.astype(np.float64) / 255.0 + rng.normal(0.0, sigma / 255.0, clean.shape)
    noisy_u8 = np.rint(np.clip(noisy, 0, 1) * 255).astype(np.uint8)
    return clean_u8, noisy_u8
