"""CPU ORACLE for RRDBNet / Real-ESRGAN -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional torch-CPU restatement of AgenticIR/executor/super_resolution/tools/DiffBIR/utils/realesrgan/rrdbnet.py: pixel_unshuffle :53-71,
ResidualDenseBlock :74-105, RRDB :108-128, RRDBNet.forward :166-182; and of the RealESRGANer tensor path (realesrganer.py:71-183,
pre_process / tile_process / post_process, the origin of the pad-crop tiler in oracle/padcrop_oracle.py).  Pinned against the reference
module (tests/golden/make_golden_rrdbnet.py; plain torch, no stubs)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def pixel_unshuffle(x, scale):
    b, c, hh, hw = x.shape
    assert hh % scale == 0 and hw % scale == 0
    h, w = hh // scale, hw // scale
    return x.view(b, c, h, scale, w, scale).permute(0, 1, 3, 5, 2, 4).reshape(b, c * scale * scale, h, w)


def _conv(x, sd, p):
    return F.conv2d(x, sd[f'{p}.weight'], sd[f'{p}.bias'], padding=1)


def rdb(x, sd, p):
    lrelu = lambda t: F.leaky_relu(t, 0.2)
    feats = [x]
    for k in range(1, 5):
        feats.append(lrelu(_conv(torch.cat(feats, 1), sd, f'{p}.conv{k}')))
    return _conv(torch.cat(feats, 1), sd, f'{p}.conv5') * 0.2 + x


def rrdbnet_forward(x, sd, cfg):
    scale = cfg['scale']
    feat = pixel_unshuffle(x, 2) if scale == 2 else (pixel_unshuffle(x, 4) if scale == 1 else x)
    feat = _conv(feat, sd, 'conv_first')
    body = feat
    for i in range(cfg['num_block']):
        out = body
        for j in (1, 2, 3):
            out = rdb(out, sd, f'body.{i}.rdb{j}')
        body = out * 0.2 + body
    feat = feat + _conv(body, sd, 'conv_body')
    lrelu = lambda t: F.leaky_relu(t, 0.2)
    feat = lrelu(_conv(F.interpolate(feat, scale_factor=2, mode='nearest'), sd, 'conv_up1'))
    feat = lrelu(_conv(F.interpolate(feat, scale_factor=2, mode='nearest'), sd, 'conv_up2'))
    return _conv(lrelu(_conv(feat, sd, 'conv_hr')), sd, 'conv_last')
