"""CPU ORACLE for NAFNet -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional torch-CPU restatement of the reference architecture file
    AgenticIR/executor/dehazing/tools/X-Restormer/xrestormer/archs/NAFNet_arch.py
(LayerNorm2d :120-156, SimpleGate :158-161, NAFBlock :163-219, NAFNet :249-361).  The reference ships the
architecture only (no tool, harness or weights: SURVEY section 2 row 6); BASELINE config 2 is the width-32
SIDD layout enc [2,2,4,8] / middle 12 / dec [2,2,2,2].  Pinned against outputs of the reference module imported in
the dev container with a 1-symbol ``basicsr.utils.registry`` stub (tests/golden/make_golden_nafnet.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def layer_norm2d(x, w, b, eps=1e-6):
    """NAFNet_arch.py:120-134: per-pixel statistics over C, biased variance, eps 1e-6, affine."""
    mu = x.mean(1, keepdim=True)
    var = (x - mu).pow(2).mean(1, keepdim=True)
    y = (x - mu) / (var + eps).sqrt()
    return w.view(1, -1, 1, 1) * y + b.view(1, -1, 1, 1)


def naf_block(inp, sd, p):
    """NAFNet_arch.py:198-219."""
    c = inp.shape[1]
    x = layer_norm2d(inp, sd[f'{p}.norm1.weight'], sd[f'{p}.norm1.bias'])
    x = F.conv2d(x, sd[f'{p}.conv1.weight'], sd[f'{p}.conv1.bias'])
    x = F.conv2d(x, sd[f'{p}.conv2.weight'], sd[f'{p}.conv2.bias'], padding=1, groups=2 * c)
    x1, x2 = x.chunk(2, dim=1)
    x = x1 * x2                                                          # SimpleGate :158-161
    s = F.conv2d(F.adaptive_avg_pool2d(x, 1), sd[f'{p}.sca.1.weight'], sd[f'{p}.sca.1.bias'])   # :175-179
    x = x * s
    x = F.conv2d(x, sd[f'{p}.conv3.weight'], sd[f'{p}.conv3.bias'])
    y = inp + x * sd[f'{p}.beta']                                        # :211
    x = F.conv2d(layer_norm2d(y, sd[f'{p}.norm2.weight'], sd[f'{p}.norm2.bias']), sd[f'{p}.conv4.weight'], sd[f'{p}.conv4.bias'])
    x1, x2 = x.chunk(2, dim=1)
    x = F.conv2d(x1 * x2, sd[f'{p}.conv5.weight'], sd[f'{p}.conv5.bias'])
    return y + x * sd[f'{p}.gamma']                                      # :219


def nafnet_forward(inp, sd, cfg):
    """NAFNet_arch.py:301-326 (scale == 1).  cfg: width, enc_blk_nums, middle_blk_num, dec_blk_nums."""
    enc, dec = cfg['enc_blk_nums'], cfg['dec_blk_nums']
    B, C, H, W = inp.shape
    ps = 2 ** len(enc)                                                   # :299
    x0 = F.pad(inp, (0, (ps - W % ps) % ps, 0, (ps - H % ps) % ps))      # check_image_size :356-361 (zeros)
    x = F.conv2d(x0, sd['intro.weight'], sd['intro.bias'], padding=1)
    skips = []
    for i, n in enumerate(enc):
        for k in range(n):
            x = naf_block(x, sd, f'encoders.{i}.{k}')
        skips.append(x)
        x = F.conv2d(x, sd[f'downs.{i}.weight'], sd[f'downs.{i}.bias'], stride=2)      # :275-277
    for k in range(cfg['middle_blk_num']):
        x = naf_block(x, sd, f'middle_blks.{k}')
    for i, n in enumerate(dec):
        x = F.pixel_shuffle(F.conv2d(x, sd[f'ups.{i}.0.weight']), 2)                    # :286-291
        x = x + skips[-1 - i]                                                            # :320
        for k in range(n):
            x = naf_block(x, sd, f'decoders.{i}.{k}')
    x = F.conv2d(x, sd['ending.weight'], sd['ending.bias'], padding=1) + x0              # :323-324
    return x[:, :, :H, :W]
