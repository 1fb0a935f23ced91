"""NAFNet configuration and state-dict layout (reference: X-Restormer/xrestormer/archs/NAFNet_arch.py:249-299).

BASELINE config 2 = NAFNet-width32 in the SIDD layout enc [2,2,4,8] / middle 12 / dec [2,2,2,2] (29.16 M parameters).
As for Restormer no checkpoint exists offline: weights are generated from (seed, tensor name)."""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass, field, asdict

import numpy as np


@dataclass
class NAFNetConfig:
    img_channel: int = 3
    width: int = 32
    middle_blk_num: int = 12
    enc_blk_nums: list = field(default_factory=lambda: [2, 2, 4, 8])
    dec_blk_nums: list = field(default_factory=lambda: [2, 2, 2, 2])

    def to_dict(self):
        return asdict(self)


def reduced_nafnet_config():
    return NAFNetConfig(width=16, middle_blk_num=1, enc_blk_nums=[1, 1], dec_blk_nums=[1, 1])


def _block_spec(p, c):
    return [(f'{p}.beta', (1, c, 1, 1)), (f'{p}.gamma', (1, c, 1, 1)),
            (f'{p}.conv1.weight', (2 * c, c, 1, 1)), (f'{p}.conv1.bias', (2 * c,)),
            (f'{p}.conv2.weight', (2 * c, 1, 3, 3)), (f'{p}.conv2.bias', (2 * c,)),
            (f'{p}.conv3.weight', (c, c, 1, 1)), (f'{p}.conv3.bias', (c,)),
            (f'{p}.sca.1.weight', (c, c, 1, 1)), (f'{p}.sca.1.bias', (c,)),
            (f'{p}.conv4.weight', (2 * c, c, 1, 1)), (f'{p}.conv4.bias', (2 * c,)),
            (f'{p}.conv5.weight', (c, c, 1, 1)), (f'{p}.conv5.bias', (c,)),
            (f'{p}.norm1.weight', (c,)), (f'{p}.norm1.bias', (c,)),
            (f'{p}.norm2.weight', (c,)), (f'{p}.norm2.bias', (c,))]


def state_dict_spec(cfg: NAFNetConfig):
    """(name, shape) in the order of the reference module's state_dict()."""
    w, ic = cfg.width, cfg.img_channel
    spec = [('intro.weight', (w, ic, 3, 3)), ('intro.bias', (w,)), ('ending.weight', (ic, w, 3, 3)), ('ending.bias', (ic,))]
    chan = w
    enc, dec, ups, downs = [], [], [], []
    for i, n in enumerate(cfg.enc_blk_nums):
        for k in range(n):
            enc += _block_spec(f'encoders.{i}.{k}', chan)
        downs += [(f'downs.{i}.weight', (2 * chan, chan, 2, 2)), (f'downs.{i}.bias', (2 * chan,))]
        chan *= 2
    mid = []
    for k in range(cfg.middle_blk_num):
        mid += _block_spec(f'middle_blks.{k}', chan)
    for i, n in enumerate(cfg.dec_blk_nums):
        ups += [(f'ups.{i}.0.weight', (2 * chan, chan, 1, 1))]
        chan //= 2
        for k in range(n):
            dec += _block_spec(f'decoders.{i}.{k}', chan)
    return spec + enc + dec + mid + ups + downs        # ModuleList registration order (:256-260)


def seeded_tensor(seed, name, shape):
    rng = np.random.default_rng([int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())])
    if name.endswith('.beta') or name.endswith('.gamma'):
        return (0.5 + 0.2 * rng.standard_normal(shape)).astype(np.float32)       # trained values are O(0.1-1); zeros would kill the branches
    if '.norm' in name and name.endswith('.weight'):
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if name.endswith('.bias'):
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    std = 1.0 / np.sqrt(fan_in)
    if name == 'ending.weight':
        std *= 0.05
    if '.sca.' in name:
        std *= 2.0
    return (std * rng.standard_normal(shape)).astype(np.float32)


def seeded_state_dict(cfg: NAFNetConfig, seed: int = 0):
    return OrderedDict((n, seeded_tensor(seed, n, s)) for n, s in state_dict_spec(cfg))
