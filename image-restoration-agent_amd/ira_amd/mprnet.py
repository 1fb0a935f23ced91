"""MPRNet configuration, state-dict layout, seeded weights (reference
AgenticIR/executor/denoising/tools/MPRNet/{Deblurring,Denoising,Deraining}/MPRNet.py, identical files except the constructor
defaults at :239; harness demo.py; tool AgenticIR/executor/multitask_tools.py:176-204)."""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass, asdict

import numpy as np


@dataclass
class MPRNetConfig:
    in_c: int = 3
    out_c: int = 3
    n_feat: int = 96
    scale_unetfeats: int = 48
    scale_orsnetfeats: int = 32
    num_cab: int = 8
    kernel_size: int = 3
    reduction: int = 4
    bias: bool = False

    def to_dict(self):
        return asdict(self)


TASK_DEFAULTS = {'Deblurring': dict(n_feat=96, scale_unetfeats=48, scale_orsnetfeats=32),      # MPRNet.py:239 of each task directory
                 'Denoising': dict(n_feat=80, scale_unetfeats=48, scale_orsnetfeats=32),
                 'Deraining': dict(n_feat=40, scale_unetfeats=20, scale_orsnetfeats=16)}
TASK_WEIGHT_FILE = {t: f'{t}/pretrained_models/model_{t.lower()}.pth' for t in TASK_DEFAULTS}        # demo.py:60
SUBTASK_TO_TASK = {'denoising': 'Denoising', 'motion_deblurring': 'Deblurring', 'deraining': 'Deraining'}   # multitask_tools.py:191-195


def task_config(task: str) -> MPRNetConfig:
    if task not in TASK_DEFAULTS:
        raise ValueError(f"Not supported task {task}")
    return MPRNetConfig(**TASK_DEFAULTS[task])


def reduced_mprnet_config(**kw) -> MPRNetConfig:
    d = dict(n_feat=16, scale_unetfeats=8, scale_orsnetfeats=8, num_cab=2)
    d.update(kw)
    return MPRNetConfig(**d)


def _cab(p, c, cfg):
    r = cfg.reduction
    return [(f'{p}.CA.conv_du.0.weight', (c // r, c, 1, 1)), (f'{p}.CA.conv_du.2.weight', (c, c // r, 1, 1)),
            (f'{p}.body.0.weight', (c, c, 3, 3)), (f'{p}.body.1.weight', (1,)), (f'{p}.body.2.weight', (c, c, 3, 3))]


def state_dict_spec(cfg: MPRNetConfig):
    """(name, shape) in the order of the reference module's state_dict() (bias=False as in every task).  The single nn.PReLU()
    instance is shared by every CAB (MPRNet.py:242), so its one parameter appears under every '...body.1.weight' name."""
    if cfg.bias:
        raise ValueError("bias=True is not used by any reference task and is not built")
    n, su, so = cfg.n_feat, cfg.scale_unetfeats, cfg.scale_orsnetfeats
    c1, c2, c3, co = n, n + su, n + 2 * su, n + so
    spec = []
    for i in (1, 2, 3):
        spec += [(f'shallow_feat{i}.0.weight', (n, cfg.in_c, 3, 3))] + _cab(f'shallow_feat{i}.1', n, cfg)

    def encoder(p, csff):
        out = []
        for lvl, c in ((1, c1), (2, c2), (3, c3)):
            for k in range(2):
                out += _cab(f'{p}.encoder_level{lvl}.{k}', c, cfg)
        out += [(f'{p}.down12.down.1.weight', (c2, c1, 1, 1)), (f'{p}.down23.down.1.weight', (c3, c2, 1, 1))]
        if csff:
            for kind in ('enc', 'dec'):
                for lvl, c in ((1, c1), (2, c2), (3, c3)):
                    out.append((f'{p}.csff_{kind}{lvl}.weight', (c, c, 1, 1)))
        return out

    def decoder(p):
        out = []
        for lvl, c in ((1, c1), (2, c2), (3, c3)):
            for k in range(2):
                out += _cab(f'{p}.decoder_level{lvl}.{k}', c, cfg)
        out += _cab(f'{p}.skip_attn1', c1, cfg) + _cab(f'{p}.skip_attn2', c2, cfg)
        out += [(f'{p}.up21.up.1.weight', (c1, c2, 1, 1)), (f'{p}.up32.up.1.weight', (c2, c3, 1, 1))]
        return out
    spec += encoder('stage1_encoder', False) + decoder('stage1_decoder') + encoder('stage2_encoder', True) + decoder('stage2_decoder')
    for i in (1, 2, 3):
        for k in range(cfg.num_cab):
            spec += _cab(f'stage3_orsnet.orb{i}.body.{k}', co, cfg)
        spec.append((f'stage3_orsnet.orb{i}.body.{cfg.num_cab}.weight', (co, co, 3, 3)))
    spec += [('stage3_orsnet.up_enc1.up.1.weight', (c1, c2, 1, 1)), ('stage3_orsnet.up_dec1.up.1.weight', (c1, c2, 1, 1)),
             ('stage3_orsnet.up_enc2.0.up.1.weight', (c2, c3, 1, 1)), ('stage3_orsnet.up_enc2.1.up.1.weight', (c1, c2, 1, 1)),
             ('stage3_orsnet.up_dec2.0.up.1.weight', (c2, c3, 1, 1)), ('stage3_orsnet.up_dec2.1.up.1.weight', (c1, c2, 1, 1))]
    for kind in ('enc', 'dec'):
        for i in (1, 2, 3):
            spec.append((f'stage3_orsnet.conv_{kind}{i}.weight', (co, c1, 1, 1)))
    for s in ('sam12', 'sam23'):
        spec += [(f'{s}.conv1.weight', (n, n, 1, 1)), (f'{s}.conv2.weight', (3, n, 1, 1)), (f'{s}.conv3.weight', (n, 3, 1, 1))]
    spec += [('concat12.weight', (n, 2 * n, 3, 3)), ('concat23.weight', (co, 2 * n, 3, 3)), ('tail.weight', (cfg.out_c, co, 3, 3))]
    return spec


def seeded_tensor(seed, name, shape):
    rng = np.random.default_rng([int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())])
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    std = 1.0 / np.sqrt(fan_in)
    if '.CA.conv_du' in name:
        std *= 2.0
    if name == 'tail.weight':
        std *= 0.02
    if name.endswith('conv2.weight'):
        std *= 0.1
    if '.body.2.weight' in name or 'csff' in name or 'conv_enc' in name or 'conv_dec' in name:
        std *= 0.5                        # residual branches: keep the 60-CAB chains at unit scale
    return (std * rng.standard_normal(shape)).astype(np.float32)


def seeded_state_dict(cfg: MPRNetConfig, seed: int = 0):
    """The shared PReLU parameter gets ONE value (0.25 + small seed-dependent offset) under all its names."""
    prelu = np.float32(0.25 + 0.05 * np.sin(seed + 1.0))
    out = OrderedDict()
    for n, s in state_dict_spec(cfg):
        out[n] = np.full(s, prelu, np.float32) if n.endswith('.body.1.weight') else seeded_tensor(seed, n, s)
    return out


def up16(n: int) -> int:
    return (n + 15) // 16 * 16


def prelu_value(sd) -> float:
    """The one shared PReLU parameter; a state dict whose '...body.1.weight' entries disagree was not produced by the reference module."""
    vals = {float(np.asarray(v).reshape(-1)[0]) for k, v in sd.items() if k.endswith('.body.1.weight')}
    if len(vals) != 1:
        raise ValueError(f"expected one shared PReLU value, found {sorted(vals)[:4]}")
    return vals.pop()


def mprnet_cfg_struct(cfg: MPRNetConfig, prelu: float):
    from ._capi import IraMprnetConfig
    return IraMprnetConfig(cfg.in_c, cfg.out_c, cfg.n_feat, cfg.scale_unetfeats, cfg.scale_orsnetfeats, cfg.num_cab, cfg.reduction, prelu)


def pack_mprnet_arrays(sd, cfg: MPRNetConfig) -> dict:
    """Blob entries of csrc/ira_rt_mprnet.hip.  Channels are zero-padded to multiples of 16; DownSample = 2x2 mean + 1x1 becomes one
    [(dy,dx,ci)] contraction with W/4 per tap; the ORSNet up -> 1x1 chains (MPRNet.py:229-231) are multiplied out in float64."""
    from .pack import frag_major, split16, up32, _np
    from .swinir import _conv3_pack_k
    f64 = lambda k: _np(sd[k]).astype(np.float64)
    m2 = lambda k: f64(k)[:, :, 0, 0]
    out: dict = {}

    def pw(w):                                             # [cout][cin] -> hi/lo fragment-major, rows up32(up16(cout)), K up16(cin)
        co, ci = w.shape
        a = np.zeros((up32(up16(co)), up16(ci)), np.float64)
        a[:co, :ci] = w
        return frag_major(split16(a))

    def cab(p, c):
        cp = up16(c)
        out[f'{p}.c0'] = _conv3_pack_k(f64(f'{p}.body.0.weight'), cp)
        out[f'{p}.c2'] = _conv3_pack_k(f64(f'{p}.body.2.weight'), cp)
        out[f'{p}.du0'] = np.ascontiguousarray(_np(sd[f'{p}.CA.conv_du.0.weight'])[:, :, 0, 0])
        out[f'{p}.du2'] = np.ascontiguousarray(_np(sd[f'{p}.CA.conv_du.2.weight'])[:, :, 0, 0])

    n, su, so = cfg.n_feat, cfg.scale_unetfeats, cfg.scale_orsnetfeats
    ch, co = (n, n + su, n + 2 * su), n + so
    np_ = up16(n)
    for i in (1, 2, 3):
        w = np.zeros((np_, cfg.in_c, 3, 3), np.float32)
        w[:n] = _np(sd[f'shallow_feat{i}.0.weight'])
        out[f'shallow_feat{i}.0'] = w
        cab(f'shallow_feat{i}.1', n)
    for s in (1, 2):
        e, d = f'stage{s}_encoder', f'stage{s}_decoder'
        for lvl in (1, 2, 3):
            for k in range(2):
                cab(f'{e}.encoder_level{lvl}.{k}', ch[lvl - 1])
                cab(f'{d}.decoder_level{lvl}.{k}', ch[lvl - 1])
        for lvl in (1, 2):
            w = m2(f'{e}.down{lvl}{lvl + 1}.down.1.weight')                        # [c_next][c]
            cn, c = w.shape
            a = np.zeros((up32(up16(cn)), 4, up16(c)), np.float64)
            a[:cn, :, :c] = (w / 4.0)[:, None, :]
            out[f'{e}.down{lvl}{lvl + 1}'] = frag_major(split16(a.reshape(a.shape[0], -1)))
            cab(f'{d}.skip_attn{lvl}', ch[lvl - 1])
            out[f'{d}.up{lvl + 1}{lvl}'] = pw(m2(f'{d}.up{lvl + 1}{lvl}.up.1.weight'))
        if s == 2:
            for kind in ('enc', 'dec'):
                for lvl in (1, 2, 3):
                    out[f'{e}.csff_{kind}{lvl}'] = pw(m2(f'{e}.csff_{kind}{lvl}.weight'))
    o = 'stage3_orsnet'
    for i in (1, 2, 3):
        for k in range(cfg.num_cab):
            cab(f'{o}.orb{i}.body.{k}', co)
        out[f'{o}.orb{i}.body.{cfg.num_cab}'] = _conv3_pack_k(f64(f'{o}.orb{i}.body.{cfg.num_cab}.weight'), up16(co))
    for kind in ('enc', 'dec'):
        out[f'{o}.{kind}1'] = pw(m2(f'{o}.conv_{kind}1.weight'))
        out[f'{o}.{kind}2'] = pw(m2(f'{o}.conv_{kind}2.weight') @ m2(f'{o}.up_{kind}1.up.1.weight'))
        out[f'{o}.{kind}3'] = pw(m2(f'{o}.conv_{kind}3.weight') @ m2(f'{o}.up_{kind}2.1.up.1.weight') @ m2(f'{o}.up_{kind}2.0.up.1.weight'))
    for s in ('sam12', 'sam23'):
        out[f'{s}.conv1'] = pw(m2(f'{s}.conv1.weight'))
        a = np.zeros((32, np_), np.float64)
        a[:3, :n] = m2(f'{s}.conv2.weight')
        out[f'{s}.conv2'] = frag_major(split16(a))
        out[f'{s}.conv3'] = np.ascontiguousarray(_np(sd[f'{s}.conv3.weight'])[:, :, 0, 0])
    for name, cout in (('concat12', n), ('concat23', co)):
        w = f64(f'{name}.weight')                                                  # [cout][2n][3][3]: [shallow | sam] (:305, :330)
        a = np.zeros((cout, 2 * np_, 3, 3), np.float64)
        a[:, :n] = w[:, :n]
        a[:, np_:np_ + n] = w[:, n:]
        out[name] = _conv3_pack_k(a, 2 * np_)
    out['tail'] = _conv3_pack_k(f64('tail.weight'), up16(co))
    return out


def pack_mprnet_blob(sd, cfg: MPRNetConfig) -> bytes:
    from .pack import blob_from_arrays
    return blob_from_arrays(pack_mprnet_arrays(sd, cfg))
