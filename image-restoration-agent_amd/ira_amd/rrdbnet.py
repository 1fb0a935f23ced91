"""RRDBNet (the Real-ESRGAN generator) configuration, state-dict layout, seeded weights and packer (reference
AgenticIR/executor/super_resolution/tools/DiffBIR/utils/realesrgan/rrdbnet.py:74-182; built by set_realesrgan, realesrganer.py:304-339,
as the background upsampler of DiffBIR's face pipeline, inference_face.py:137-142 -- no registered tool dispatches it)."""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass, asdict

import numpy as np


@dataclass
class RRDBNetConfig:
    num_in_ch: int = 3
    num_out_ch: int = 3
    scale: int = 4
    num_feat: int = 64
    num_block: int = 23
    num_grow_ch: int = 32

    def to_dict(self):
        return asdict(self)


MODEL_FILES = {2: 'RealESRGAN_x2plus.pth', 4: 'RealESRGAN_x4plus.pth'}          # realesrganer.py:316-319


def realesrgan_config(scale: int = 2) -> RRDBNetConfig:
    """set_realesrgan (realesrganer.py:304-328): 3, 3, 64, 23, 32 at scale 2 or 4."""
    if scale not in (2, 4):
        raise ValueError("RealESRGAN official models only support x2 and x4 upsampling")
    return RRDBNetConfig(scale=scale)


def reduced_rrdbnet_config(**kw) -> RRDBNetConfig:
    d = dict(num_feat=32, num_block=2, num_grow_ch=32, scale=4)
    d.update(kw)
    return RRDBNetConfig(**d)


def unshuffle_factor(scale: int) -> int:
    return {4: 1, 2: 2, 1: 4}[scale]                                              # rrdbnet.py:149-152,167-172


def state_dict_spec(cfg: RRDBNetConfig):
    """(name, shape) in the order of the reference module's state_dict()."""
    nf, gc = cfg.num_feat, cfg.num_grow_ch
    ic = cfg.num_in_ch * unshuffle_factor(cfg.scale) ** 2
    spec = [('conv_first.weight', (nf, ic, 3, 3)), ('conv_first.bias', (nf,))]
    for i in range(cfg.num_block):
        for j in (1, 2, 3):
            for k in range(1, 6):
                n = nf if k == 5 else gc
                spec += [(f'body.{i}.rdb{j}.conv{k}.weight', (n, nf + (k - 1) * gc, 3, 3)), (f'body.{i}.rdb{j}.conv{k}.bias', (n,))]
    for name, n in (('conv_body', nf), ('conv_up1', nf), ('conv_up2', nf), ('conv_hr', nf), ('conv_last', cfg.num_out_ch)):
        spec += [(f'{name}.weight', (n, nf, 3, 3)), (f'{name}.bias', (n,))]
    return spec


def seeded_tensor(seed, name, shape):
    rng = np.random.default_rng([int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())])
    if len(shape) == 1:
        return (0.02 * rng.standard_normal(shape)).astype(np.float32)
    std = np.sqrt(2.0 / int(np.prod(shape[1:])))
    if '.rdb' in name:
        std *= 0.3                            # dense blocks: keep the 23-block trunk near unit scale (the reference initialises them at 0.1)
    if name.startswith('conv_last'):
        std *= 0.3
    return (std * rng.standard_normal(shape)).astype(np.float32)


def seeded_state_dict(cfg: RRDBNetConfig, seed: int = 0):
    return OrderedDict((n, seeded_tensor(seed, n, s)) for n, s in state_dict_spec(cfg))


def rrdbnet_cfg_struct(cfg: RRDBNetConfig):
    from ._capi import IraRrdbnetConfig
    return IraRrdbnetConfig(cfg.num_in_ch, cfg.num_out_ch, cfg.scale, cfg.num_feat, cfg.num_block, cfg.num_grow_ch)


def pack_rrdbnet_arrays(sd, cfg: RRDBNetConfig) -> dict:
    """Blob entries of csrc/ira_rt_rrdbnet.hip.  conv5 of every dense block carries the 0.2 of ``x5 * 0.2 + x`` (rrdbnet.py:105)."""
    from .pack import _np, _pad_rows, up32
    from .swinir import _conv3_pack_k
    out: dict = {}

    def conv(dst, src, scale=1.0):
        w = _np(sd[f'{src}.weight']).astype(np.float64) * scale
        out[f'{dst}_w'] = _conv3_pack_k(w, w.shape[1])
        out[f'{dst}_b'] = _pad_rows((_np(sd[f'{src}.bias']).astype(np.float64) * scale).astype(np.float32), up32(w.shape[0]))
    ic = cfg.num_in_ch * unshuffle_factor(cfg.scale) ** 2
    if ic % 16 == 0:
        conv('first', 'conv_first')
    else:
        out['first_w'] = np.ascontiguousarray(_np(sd['conv_first.weight']))
        out['first_b'] = _np(sd['conv_first.bias']).astype(np.float32)
    for i in range(cfg.num_block):
        for j in (1, 2, 3):
            for k in range(1, 6):
                conv(f'b{i}.r{j}.c{k}', f'body.{i}.rdb{j}.conv{k}', 0.2 if k == 5 else 1.0)
    for dst, src in (('body', 'conv_body'), ('up1', 'conv_up1'), ('up2', 'conv_up2'), ('hr', 'conv_hr'), ('last', 'conv_last')):
        conv(dst, src)
    return out


def pack_rrdbnet_blob(sd, cfg: RRDBNetConfig) -> bytes:
    from .pack import blob_from_arrays
    return blob_from_arrays(pack_rrdbnet_arrays(sd, cfg))
