"""CPU ORACLE for the pad-crop tiler -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates X-Restormer/xrestormer/models/xrestormer_model.py: pre_process :53-63, tile_process :65-133, post_process
:134-136, around any ``forward_fn`` (NCHW float tensor -> same, times ``scale``), and the same tiler as DiffBIR's RealESRGANer runs it
(utils/realesrgan/realesrganer.py:71-183: pre_pad, mod-pad by scale, tile_process, un-pad).
PINNED (round 2): tests/golden/make_golden_padcrop.py executes the reference's own methods (unbound, on a namespace carrying the attributes
they read, with the reference networks) and stores their float outputs; tests/test_padcrop.py checks both functions below against them.
The uint8 conversion around the tensor path (SingleImageDataset / tensor2img / cv2) lives in pip basicsr / OpenCV, which are not in
/root/reference: that boundary alone stays unpinned; it is restated as /255 in, clamp(0,1) -> round-half-even(x*255) out, like the
Restormer harness."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def padcrop_restore(lq, forward_fn, tile_size=None, tile_pad=32, padding_size=64, scale=1):
    """lq: (1,C,h,w) float tensor -> (1,C,h*scale,w*scale)."""
    _, _, h, w = lq.shape
    mod_pad_h = padding_size - h % padding_size if h % padding_size != 0 else 0          # :58-61
    mod_pad_w = padding_size - w % padding_size if w % padding_size != 0 else 0
    lq = F.pad(lq, (0, mod_pad_w, 0, mod_pad_h), 'reflect')                              # :63
    if not tile_size:
        out = forward_fn(lq)                                                             # test() :138-150
    else:
        b, c, height, width = lq.shape
        out = lq.new_zeros((b, c, height * scale, width * scale))                        # :72-77
        tiles_x, tiles_y = math.ceil(width / tile_size), math.ceil(height / tile_size)   # :78-79
        for y in range(tiles_y):
            for x in range(tiles_x):
                sx, sy = x * tile_size, y * tile_size                                    # :85-86
                ex, ey = min(sx + tile_size, width), min(sy + tile_size, height)         # :88-91
                sxp, exp_ = max(sx - tile_pad, 0), min(ex + tile_pad, width)             # :94-95
                syp, eyp = max(sy - tile_pad, 0), min(ey + tile_pad, height)             # :96-97
                tile = forward_fn(lq[:, :, syp:eyp, sxp:exp_])                           # :103-114
                oy, ox = (sy - syp) * scale, (sx - sxp) * scale                          # :125-128
                out[:, :, sy * scale:ey * scale, sx * scale:ex * scale] = tile[:, :, oy:oy + (ey - sy) * scale, ox:ox + (ex - sx) * scale]   # :117-133
    _, _, H, W = out.shape
    return out[:, :, 0:H - mod_pad_h * scale, 0:W - mod_pad_w * scale]                   # :134-136


def realesrganer_restore(img, forward_fn, scale, tile_size=0, tile_pad=10, pre_pad=0):
    """RealESRGANer.pre_process / process / tile_process / post_process (realesrganer.py:71-183).  img: (1,C,h,w) float tensor."""
    if pre_pad:
        img = F.pad(img, (0, pre_pad, 0, pre_pad), 'reflect')                            # :80-81
    mod_scale = 2 if scale == 2 else (4 if scale == 1 else None)                         # :83-86
    mph = mpw = 0
    if mod_scale is not None:
        _, _, h, w = img.shape
        mph = (mod_scale - h % mod_scale) if h % mod_scale else 0                        # :89-92
        mpw = (mod_scale - w % mod_scale) if w % mod_scale else 0
        img = F.pad(img, (0, mpw, 0, mph), 'reflect')                                    # :93
    # the tile loop (:99-165) is the loop of xrestormer_model.py:65-133 with no further padding
    out = padcrop_restore(img, forward_fn, tile_size if tile_size > 0 else None, tile_pad, 1, scale)
    _, _, H, W = out.shape
    out = out[:, :, 0:H - mph * scale, 0:W - mpw * scale]                                # :169-171
    if pre_pad:
        _, _, H, W = out.shape
        out = out[:, :, 0:H - pre_pad * scale, 0:W - pre_pad * scale]                    # :173-175
    return out


def padcrop_restore_image(img_u8, forward_fn, tile_size=None, tile_pad=32, padding_size=64, scale=1):
    """uint8 (h,w,c) -> uint8 (h*scale,w*scale,c)."""
    x = torch.from_numpy(np.ascontiguousarray(img_u8)).float().div(255.).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        y = padcrop_restore(x, forward_fn, tile_size, tile_pad, padding_size, scale)
    y = torch.clamp(y, 0, 1)[0].permute(1, 2, 0).numpy()
    return np.rint(y.astype(np.float32) * np.float32(255.0)).astype(np.uint8)
