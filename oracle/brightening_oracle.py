"""CPU ORACLE for the brightening tools -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

AgenticIR/executor/brightening/__init__.py:19-53: cv2.imread -> cv2.cvtColor(BGR2HSV) -> a look-up on V (ConstantShift :37-41: min(V + 40, 255);
GammaCorrection :47-51: round(clip((V / 255) ** (1 / 1.5) * 255))) -> cv2.cvtColor(HSV2BGR) -> cv2.imwrite.

PARITY UNPINNED.  The arithmetic lives in OpenCV (opencv-python==4.8.0.76 in the reference's Dockerfile:64), which is neither in /root/reference
nor in this image, and the reference holds no fixture for these tools.  Restated here from OpenCV's published algorithm
(modules/imgproc/src/color_hsv.simd.hpp): 8-bit RGB2HSV_b with hrange 180 is integer arithmetic on two division tables with 12 fractional bits
(sdiv_table[v] = round(255 * 4096 / v), hdiv_table[d] = round(180 * 4096 / (6 d))); 8-bit HSV2RGB_b goes through float32 (h, s / 255, v / 255),
the six-sector formula, and saturate_cast<uchar>(x * 255) (round half to even).  OpenCV evaluates that formula in two algebraically equal
ways whose float32 roundings differ in rare pixels: the universal-intrinsics loop (HSV2RGB_simd: v - v s, v - (v s) h, (v - v s) + (v s) h),
which handles all but the last (width mod lanes) pixels of a row on every SIMD build, and the scalar tail (HSV2RGB_native: v (1 - s),
v (1 - s h), v (1 - s (1 - h))).  ``formulation='simd'`` is what the product implements; 'scalar' is kept to bound the difference (<= 1 LSB).
HistogramEqualization (:56-63) is cv2.createCLAHE(2.0, (8, 8)).apply(V): modules/imgproc/src/clahe.cpp restated in clahe_u8 below.
The anchor for parity is the call site above.
"""
from __future__ import annotations

import numpy as np

HSV_SHIFT = 12


def _tables():
    i = np.arange(1, 256, dtype=np.float64)
    sdiv = np.zeros(256, np.int64)
    hdiv = np.zeros(256, np.int64)
    sdiv[1:] = np.rint((255 << HSV_SHIFT) / (1.0 * i)).astype(np.int64)           # saturate_cast<int>(double) = cvRound: half to even
    hdiv[1:] = np.rint((180 << HSV_SHIFT) / (6.0 * i)).astype(np.int64)
    return sdiv, hdiv


def rgb_to_hsv_u8(img):
    """uint8 (h,w,3) RGB -> uint8 H (0..179), S, V; cv2.cvtColor(COLOR_RGB2HSV) = COLOR_BGR2HSV on the BGR copy."""
    sdiv, hdiv = _tables()
    r, g, b = (img[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = v - vmin
    vr = np.where(v == r, -1, 0)
    vg = np.where(v == g, -1, 0)
    s = (diff * sdiv[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))))
    h = (h * hdiv[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = h + np.where(h < 0, 180, 0)
    return np.clip(h, 0, 255).astype(np.uint8), s.astype(np.uint8), v.astype(np.uint8)


def hsv_to_rgb_u8(h, s, v, formulation='simd'):
    """uint8 H, S, V -> uint8 (h,w,3) RGB: HSV2RGB_b, float32 arithmetic, one rounding per operation."""
    f = np.float32
    hf = h.astype(f) * (f(6.0) / f(180))
    sf = s.astype(f) * f(1.0 / 255.0)
    vf = v.astype(f) * f(1.0 / 255.0)
    hf = np.fmod(hf, f(6.0))
    sector = np.floor(hf).astype(np.int64)
    hf = hf - sector.astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    hf = np.where(bad, f(0), hf)
    t0 = vf
    if formulation == 'simd':
        vs = vf * sf
        t1 = vf - vs
        t2 = vf - vs * hf
        t3 = t1 + vs * hf
    else:
        t1 = vf * (f(1) - sf)
        t2 = vf * (f(1) - sf * hf)
        t3 = vf * (f(1) - sf * (f(1) - hf))
    tab = np.stack([t0, t1, t2, t3], -1)
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])      # (b, g, r) table index per sector
    idx = sd[sector]
    b = np.take_along_axis(tab, idx[..., 0:1], -1)[..., 0]
    g = np.take_along_axis(tab, idx[..., 1:2], -1)[..., 0]
    r = np.take_along_axis(tab, idx[..., 2:3], -1)[..., 0]
    if formulation != 'simd':
        gray = sf == 0
        r, g, b = (np.where(gray, vf, c) for c in (r, g, b))
    out = np.stack([r, g, b], -1).astype(f) * f(255.0)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def v_lut(tool_name: str) -> np.ndarray:
    """The 256-entry look-up on V of each tool (brightening/__init__.py:37-51), evaluated as the reference writes it."""
    v = np.arange(256, dtype=np.uint8)
    if tool_name == 'constant_shift':
        return np.clip(np.uint16(v) + 40, 0, 255).round().astype(np.uint8)                 # :39-41
    if tool_name == 'gamma_correction':
        img = np.power(v / 255.0, 1.0 / 1.5) * 255                                          # cv2.pow on a float64 array :49-50
        return img.clip(0, 255).round().astype(np.uint8)
    raise ValueError(f"unknown brightening tool {tool_name!r}")


def _reflect101(a, bottom, right):
    return np.pad(a, ((0, bottom), (0, right)), mode='reflect')            # numpy 'reflect' = BORDER_REFLECT_101


def clahe_luts(v, clip_limit=2.0, tiles=8):
    """clahe.cpp CLAHE_CalcLut_Body: per-tile clipped histogram -> cumulative look-up.  Returns (luts [tiles*tiles, 256] uint8, tile_h, tile_w)."""
    h, w = v.shape
    if h % tiles == 0 and w % tiles == 0:
        ext = v
    else:
        ext = _reflect101(v, tiles - h % tiles, tiles - w % tiles)           # clahe.cpp pads by a whole tiles - (n % tiles), also when one side divides
    th, tw = ext.shape[0] // tiles, ext.shape[1] // tiles
    area = th * tw
    lut_scale = np.float32(255) / np.float32(area)
    clip = 0
    if clip_limit > 0:
        clip = max(int(clip_limit * area / 256), 1)
    luts = np.zeros((tiles * tiles, 256), np.uint8)
    for ty in range(tiles):
        for tx in range(tiles):
            hist = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch = clipped // 256
                residual = clipped - batch * 256
                hist += batch
                if residual != 0:
                    step = max(256 // residual, 1)
                    i = 0
                    while i < 256 and residual > 0:
                        hist[i] += 1
                        i += step
                        residual -= 1
            cum = np.cumsum(hist).astype(np.float32) * lut_scale                        # int sum -> float, one float multiply
            luts[ty * tiles + tx] = np.clip(np.rint(cum), 0, 255).astype(np.uint8)
    return luts, th, tw


def clahe_u8(v, clip_limit=2.0, tiles=8):
    """cv2.createCLAHE(clip_limit, (tiles, tiles)).apply(v) on a uint8 plane: clahe.cpp CLAHE_Interpolation_Body, float32."""
    f = np.float32
    h, w = v.shape
    luts, th, tw = clahe_luts(v, clip_limit, tiles)

    def axis(n, size):
        t = np.arange(n, dtype=f) * (f(1) / f(size)) - f(0.5)
        t1 = np.floor(t).astype(np.int64)
        a = (t - t1.astype(f)).astype(f)
        return np.maximum(t1, 0), np.minimum(t1 + 1, tiles - 1), a, (f(1) - a).astype(f)
    ty1, ty2, ya, ya1 = axis(h, th)
    tx1, tx2, xa, xa1 = axis(w, tw)
    src = v.astype(np.int64)
    l = luts.astype(f)

    def tap(tyv, txv):
        return l[tyv[:, None] * tiles + txv[None, :], src]
    top = tap(ty1, tx1) * xa1[None, :] + tap(ty1, tx2) * xa[None, :]
    bot = tap(ty2, tx1) * xa1[None, :] + tap(ty2, tx2) * xa[None, :]
    res = top * ya1[:, None] + bot * ya[:, None]
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def update_v(v, tool_name: str):
    if tool_name == 'histogram_equalization':
        return clahe_u8(v)
    return v_lut(tool_name)[v]


def brighten(img_rgb_u8, tool_name: str, formulation='simd'):
    """BrighteningTool._invoke between imread and imwrite (brightening/__init__.py:21-29), on RGB instead of BGR (H, S, V do not depend on the order)."""
    h, s, v = rgb_to_hsv_u8(img_rgb_u8)
    return hsv_to_rgb_u8(h, s, update_v(v, tool_name), formulation)
