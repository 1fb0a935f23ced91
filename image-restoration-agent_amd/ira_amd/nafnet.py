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


# ---------------------------------------------------------------------------------------------------------------
# packer: reference state_dict -> device blob (layouts of csrc/ira_nafnet.hip; container format of ira_amd.pack)

def nafnet_cfg_struct(cfg: NAFNetConfig):
    from ._capi import IraNafnetConfig, NAFNET_MAX_LEVELS
    L = len(cfg.enc_blk_nums)
    if L != len(cfg.dec_blk_nums) or not 1 <= L <= NAFNET_MAX_LEVELS:
        raise ValueError(f"enc_blk_nums / dec_blk_nums must have the same length in 1..{NAFNET_MAX_LEVELS}")
    s = IraNafnetConfig()
    s.img_channel, s.width, s.n_levels, s.middle_blk_num = cfg.img_channel, cfg.width, L, cfg.middle_blk_num
    for i in range(L):
        s.enc_blk_nums[i] = cfg.enc_blk_nums[i]
        s.dec_blk_nums[i] = cfg.dec_blk_nums[i]
    return s


def pack_nafnet_block(sd, p: str, c: int, out: dict, name: str):
    """One NAFBlock (NAFNet_arch.py:163-219).  Folds: LayerNorm2d gain/bias into conv1 / conv4
    (conv(g*n + b) = (W diag g) n + (W b + bias)); beta into conv3 and gamma into conv5 (weights and biases)."""
    from .pack import frag_major, split16, up32, _np, _pad_rows, _dw_pack
    f64 = lambda k: _np(sd[f'{p}.{k}']).astype(np.float64)
    for norm, conv, dst in (('norm1', 'conv1', 'c1'), ('norm2', 'conv4', 'c4')):
        w = f64(f'{conv}.weight')[:, :, 0, 0]
        out[f'{name}.{dst}_w'] = frag_major(split16(_pad_rows(w * f64(f'{norm}.weight')[None, :], up32(2 * c))))
        out[f'{name}.{dst}_b'] = _pad_rows((w @ f64(f'{norm}.bias') + f64(f'{conv}.bias')).astype(np.float32), up32(2 * c))
    out[f'{name}.dw_w'] = _dw_pack(_np(sd[f'{p}.conv2.weight']))
    out[f'{name}.dw_b'] = _np(sd[f'{p}.conv2.bias']).astype(np.float32)
    out[f'{name}.sca_w'] = np.ascontiguousarray(_np(sd[f'{p}.sca.1.weight'])[:, :, 0, 0])
    out[f'{name}.sca_b'] = _np(sd[f'{p}.sca.1.bias']).astype(np.float32)
    beta, gamma = f64('beta').reshape(-1), f64('gamma').reshape(-1)
    out[f'{name}.c3_w'] = frag_major(split16(_pad_rows(beta[:, None] * f64('conv3.weight')[:, :, 0, 0], up32(c))))
    out[f'{name}.c3_b'] = _pad_rows((beta * f64('conv3.bias')).astype(np.float32), up32(c))
    out[f'{name}.c5_w'] = frag_major(split16(_pad_rows(gamma[:, None] * f64('conv5.weight')[:, :, 0, 0], up32(c))))
    out[f'{name}.c5_b'] = _pad_rows((gamma * f64('conv5.bias')).astype(np.float32), up32(c))


def pack_nafnet_arrays(sd, cfg: NAFNetConfig) -> dict:
    from .pack import frag_major, split16, up32, _np, _pad_rows, _conv3_pack
    out: dict = {}
    out['intro_w'] = np.ascontiguousarray(_np(sd['intro.weight']))
    out['intro_b'] = _np(sd['intro.bias']).astype(np.float32)
    out['ending_w'] = _conv3_pack(_np(sd['ending.weight']))
    out['ending_b'] = _pad_rows(_np(sd['ending.bias']).astype(np.float32), up32(cfg.img_channel))
    ch = cfg.width
    for i, n in enumerate(cfg.enc_blk_nums):
        for k in range(n):
            pack_nafnet_block(sd, f'encoders.{i}.{k}', ch, out, f'enc.{i}.{k}')
        w = _np(sd[f'downs.{i}.weight']).astype(np.float64)                  # [2c][c][2][2] -> [2c][(dy,dx,ci)]
        w = w.transpose(0, 2, 3, 1).reshape(2 * ch, 4 * ch)
        out[f'down.{i}_w'] = frag_major(split16(_pad_rows(w, up32(2 * ch))))
        out[f'down.{i}_b'] = _pad_rows(_np(sd[f'downs.{i}.bias']).astype(np.float32), up32(2 * ch))
        ch *= 2
    for k in range(cfg.middle_blk_num):
        pack_nafnet_block(sd, f'middle_blks.{k}', ch, out, f'mid.{k}')
    for i, n in enumerate(cfg.dec_blk_nums):
        w = _np(sd[f'ups.{i}.0.weight']).astype(np.float64)[:, :, 0, 0]     # [2ch][ch]; PixelShuffle: row = cc*4 + q
        nq = (2 * ch) // 4
        w = w.reshape(nq, 4, ch).transpose(1, 0, 2).reshape(2 * ch, ch)     # row' = q*nq + cc
        out[f'up.{i}_w'] = frag_major(split16(_pad_rows(w, up32(2 * ch))))
        ch //= 2
        for k in range(n):
            pack_nafnet_block(sd, f'decoders.{i}.{k}', ch, out, f'dec.{i}.{k}')
    return out


def pack_nafnet_blob(sd, cfg: NAFNetConfig) -> bytes:
    from .pack import blob_from_arrays
    return blob_from_arrays(pack_nafnet_arrays(sd, cfg))
