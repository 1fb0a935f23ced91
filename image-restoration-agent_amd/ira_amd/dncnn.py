"""DnCNN (BASELINE config 1: "DnCNN sigma=25 Gaussian denoise, single 256x256 grayscale tile").

The reference repository contains NO DnCNN code: this is a build-defined plumbing model, the
classic DnCNN-S layout as an ``nn.Sequential`` named ``model`` would serialise it:
    model.0            Conv2d(in_nc, nc, 3, padding=1, bias=True)         -> ReLU (model.1)
    model.{2+3k}       Conv2d(nc, nc, 3, padding=1, bias=True)            k = 0 .. nb-3
    model.{3+3k}       BatchNorm2d(nc)  (eval: running statistics)        -> ReLU (model.{4+3k})
    model.{2+3(nb-2)}  Conv2d(nc, in_nc, 3, padding=1, bias=True)
    output = input - model(input)                                          (the network predicts the noise)
Parity versus the reference is therefore UNPINNED by construction (nothing to compare with); the oracle
(oracle/dncnn_oracle.py) is checked against torch.nn modules built from this layout."""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass, asdict

import numpy as np


@dataclass
class DnCNNConfig:
    in_nc: int = 1
    nc: int = 64
    nb: int = 17

    def to_dict(self):
        return asdict(self)


def state_dict_spec(cfg: DnCNNConfig):
    spec = [('model.0.weight', (cfg.nc, cfg.in_nc, 3, 3)), ('model.0.bias', (cfg.nc,))]
    for k in range(cfg.nb - 2):
        c, b = 2 + 3 * k, 3 + 3 * k
        spec += [(f'model.{c}.weight', (cfg.nc, cfg.nc, 3, 3)), (f'model.{c}.bias', (cfg.nc,)),
                 (f'model.{b}.weight', (cfg.nc,)), (f'model.{b}.bias', (cfg.nc,)),
                 (f'model.{b}.running_mean', (cfg.nc,)), (f'model.{b}.running_var', (cfg.nc,))]
    last = 2 + 3 * (cfg.nb - 2)
    spec += [(f'model.{last}.weight', (cfg.in_nc, cfg.nc, 3, 3)), (f'model.{last}.bias', (cfg.in_nc,))]
    return spec


def seeded_tensor(seed, name, shape):
    rng = np.random.default_rng([int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())])
    if name.endswith('running_var'):
        return (0.5 + rng.random(shape)).astype(np.float32)
    if name.endswith('running_mean'):
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if len(shape) == 1 and name.endswith('.weight'):                 # BN gain
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if name.endswith('.bias'):
        return (0.05 * rng.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    std = np.sqrt(2.0 / fan_in)                                      # He: keeps the ReLU chain at unit scale
    if shape[0] <= 3:
        std *= 0.05                                                  # last layer: a noise-sized residual
    return (std * rng.standard_normal(shape)).astype(np.float32)


def seeded_state_dict(cfg: DnCNNConfig, seed: int = 0):
    return OrderedDict((n, seeded_tensor(seed, n, s)) for n, s in state_dict_spec(cfg))


def dncnn_cfg_struct(cfg: DnCNNConfig):
    from ._capi import IraDncnnConfig
    s = IraDncnnConfig()
    s.in_nc, s.nc, s.nb = cfg.in_nc, cfg.nc, cfg.nb
    return s


def pack_dncnn_arrays(sd, cfg: DnCNNConfig, bn_eps: float = 1e-5) -> dict:
    """BatchNorm (eval) folded into the preceding conv: w' = w * g / sqrt(var + eps), b' = (b - mean) * g / sqrt(var + eps) + beta.
    The last conv is negated so that the kernel's "conv + skip" epilogue computes input - noise."""
    from .pack import _conv3_pack, _np, _pad_rows, up32
    out: dict = {}
    out['first_w'] = np.ascontiguousarray(_np(sd['model.0.weight']))
    out['first_b'] = _np(sd['model.0.bias']).astype(np.float32)
    for k in range(cfg.nb - 2):
        c, b = 2 + 3 * k, 3 + 3 * k
        w = _np(sd[f'model.{c}.weight']).astype(np.float64)
        scale = _np(sd[f'model.{b}.weight']).astype(np.float64) / np.sqrt(_np(sd[f'model.{b}.running_var']).astype(np.float64) + bn_eps)
        bias = (_np(sd[f'model.{c}.bias']).astype(np.float64) - _np(sd[f'model.{b}.running_mean'])) * scale + _np(sd[f'model.{b}.bias'])
        out[f'mid.{k}_w'] = _conv3_pack(w * scale[:, None, None, None])
        out[f'mid.{k}_b'] = _pad_rows(bias.astype(np.float32), up32(cfg.nc))
    last = 2 + 3 * (cfg.nb - 2)
    out['last_w'] = _conv3_pack(-_np(sd[f'model.{last}.weight']).astype(np.float64))
    out['last_b'] = _pad_rows((-_np(sd[f'model.{last}.bias'])).astype(np.float32), up32(cfg.in_nc))
    return out


def pack_dncnn_blob(sd, cfg: DnCNNConfig) -> bytes:
    from .pack import blob_from_arrays
    return blob_from_arrays(pack_dncnn_arrays(sd, cfg))
