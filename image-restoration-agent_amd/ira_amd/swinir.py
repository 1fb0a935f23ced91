"""SwinIR configuration, state-dict layout, seeded weights (reference: AgenticIR/executor/super_resolution/tools/SwinIR/
models/network_swinir.py:618-840; task constructors inference.py:136-200; the agent's three variants
AgenticIR/executor/multitask_tools.py:81-133).  SURVEY 8 rows a9-a10."""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass, field, asdict

import numpy as np


@dataclass
class SwinIRConfig:
    upscale: int = 1
    in_chans: int = 3
    img_size: int = 128
    window_size: int = 8
    img_range: float = 1.0
    depths: list = field(default_factory=lambda: [6] * 6)
    embed_dim: int = 180
    num_heads: list = field(default_factory=lambda: [6] * 6)
    mlp_ratio: float = 2.0
    upsampler: str = ''                    # '', 'pixelshuffle', 'nearest+conv'
    resi_connection: str = '1conv'         # '1conv', '3conv'

    def to_dict(self):
        return asdict(self)


def task_config(task: str, scale: int = 4, large_model: bool = True, training_patch_size: int = 64) -> SwinIRConfig:
    """inference.py:136-200 (define_model)."""
    if task == 'classical_sr':
        return SwinIRConfig(upscale=scale, img_size=training_patch_size, upsampler='pixelshuffle')
    if task == 'real_sr':
        if not large_model:
            return SwinIRConfig(upscale=scale, img_size=64, upsampler='nearest+conv')
        return SwinIRConfig(upscale=scale, img_size=64, depths=[6] * 9, embed_dim=240, num_heads=[8] * 9, upsampler='nearest+conv',
                            resi_connection='3conv')
    if task == 'gray_dn':
        return SwinIRConfig(in_chans=1)
    if task == 'color_dn':
        return SwinIRConfig()
    if task == 'jpeg_car':
        return SwinIRConfig(in_chans=1, img_size=126, window_size=7, img_range=255.)
    if task == 'color_jpeg_car':
        return SwinIRConfig(img_size=126, window_size=7, img_range=255.)
    raise ValueError(f"unsupported SwinIR task {task!r} (lightweight_sr's pixelshuffledirect upsampler is not built)")


def reduced_swinir_config(**kw) -> SwinIRConfig:
    d = dict(img_size=32, depths=[2, 2], embed_dim=60, num_heads=[2, 2], mlp_ratio=2.0)
    d.update(kw)
    return SwinIRConfig(**d)


PARAM_KEY = {'real_sr': 'params_ema'}      # inference.py:161; every other task: 'params'
RGB_MEAN = (0.4488, 0.4371, 0.4040)        # network_swinir.py:659-660


def _resi_spec(p, dim, kind):
    if kind == '1conv':
        return [(f'{p}.weight', (dim, dim, 3, 3)), (f'{p}.bias', (dim,))]
    q = dim // 4
    return [(f'{p}.0.weight', (q, dim, 3, 3)), (f'{p}.0.bias', (q,)), (f'{p}.2.weight', (q, q, 1, 1)), (f'{p}.2.bias', (q,)),
            (f'{p}.4.weight', (dim, q, 3, 3)), (f'{p}.4.bias', (dim,))]


def state_dict_spec(cfg: SwinIRConfig):
    """(name, shape) of the PARAMETERS, in the order of the reference module's state_dict() (its registered buffers
    relative_position_index / attn_mask are derived data and are ignored by the packer)."""
    C, ws = cfg.embed_dim, cfg.window_size
    hid = int(C * cfg.mlp_ratio)
    spec = [('conv_first.weight', (C, cfg.in_chans, 3, 3)), ('conv_first.bias', (C,)),
            ('patch_embed.norm.weight', (C,)), ('patch_embed.norm.bias', (C,))]
    for i, (depth, heads) in enumerate(zip(cfg.depths, cfg.num_heads)):
        for k in range(depth):
            p = f'layers.{i}.residual_group.blocks.{k}'
            spec += [(f'{p}.norm1.weight', (C,)), (f'{p}.norm1.bias', (C,)),
                     (f'{p}.attn.relative_position_bias_table', ((2 * ws - 1) ** 2, heads)),
                     (f'{p}.attn.qkv.weight', (3 * C, C)), (f'{p}.attn.qkv.bias', (3 * C,)),
                     (f'{p}.attn.proj.weight', (C, C)), (f'{p}.attn.proj.bias', (C,)),
                     (f'{p}.norm2.weight', (C,)), (f'{p}.norm2.bias', (C,)),
                     (f'{p}.mlp.fc1.weight', (hid, C)), (f'{p}.mlp.fc1.bias', (hid,)),
                     (f'{p}.mlp.fc2.weight', (C, hid)), (f'{p}.mlp.fc2.bias', (C,))]
        spec += _resi_spec(f'layers.{i}.conv', C, cfg.resi_connection)
    spec += [('norm.weight', (C,)), ('norm.bias', (C,))]
    spec += _resi_spec('conv_after_body', C, cfg.resi_connection)
    nf, oc = 64, cfg.in_chans
    if cfg.upsampler == 'pixelshuffle':
        spec += [('conv_before_upsample.0.weight', (nf, C, 3, 3)), ('conv_before_upsample.0.bias', (nf,))]
        s, j = cfg.upscale, 0
        while s > 1:
            spec += [(f'upsample.{j}.weight', (4 * nf, nf, 3, 3)), (f'upsample.{j}.bias', (4 * nf,))]
            s //= 2
            j += 2
        spec += [('conv_last.weight', (oc, nf, 3, 3)), ('conv_last.bias', (oc,))]
    elif cfg.upsampler == 'nearest+conv':
        spec += [('conv_before_upsample.0.weight', (nf, C, 3, 3)), ('conv_before_upsample.0.bias', (nf,)),
                 ('conv_up1.weight', (nf, nf, 3, 3)), ('conv_up1.bias', (nf,))]
        if cfg.upscale == 4:
            spec += [('conv_up2.weight', (nf, nf, 3, 3)), ('conv_up2.bias', (nf,))]
        spec += [('conv_hr.weight', (nf, nf, 3, 3)), ('conv_hr.bias', (nf,)), ('conv_last.weight', (oc, nf, 3, 3)), ('conv_last.bias', (oc,))]
    elif cfg.upsampler == '':
        spec += [('conv_last.weight', (oc, C, 3, 3)), ('conv_last.bias', (oc,))]
    else:
        raise ValueError(f"upsampler {cfg.upsampler!r} is not built (supported: '', 'pixelshuffle', 'nearest+conv')")
    return spec


def seeded_tensor(seed, name, shape):
    rng = np.random.default_rng([int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())])
    if name.endswith('relative_position_bias_table'):
        return (0.5 * rng.standard_normal(shape)).astype(np.float32)
    if 'norm' in name and name.endswith('.weight'):
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if name.endswith('.bias'):
        return (0.05 * rng.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    std = 1.0 / np.sqrt(fan_in)
    if name.endswith('qkv.weight'):
        std *= 1.5                                   # logits of O(1): the softmax is neither flat nor one-hot
    if name.startswith('conv_last'):
        std *= 0.1
    return (std * rng.standard_normal(shape)).astype(np.float32)


def seeded_state_dict(cfg: SwinIRConfig, seed: int = 0):
    return OrderedDict((n, seeded_tensor(seed, n, s)) for n, s in state_dict_spec(cfg))


# ---------------------------------------------------------------------------------------------------------------
# packer: reference state_dict -> device blob (layouts of csrc/ira_swinir.hip / ira_runtime.hip::sw_forward_enqueue)
UPSAMPLER_ID = {'': 0, 'pixelshuffle': 1, 'nearest+conv': 2}


def swinir_cfg_struct(cfg: SwinIRConfig):
    from ._capi import IraSwinirConfig, SWINIR_MAX_LAYERS
    L = len(cfg.depths)
    if L != len(cfg.num_heads) or not 1 <= L <= SWINIR_MAX_LAYERS:
        raise ValueError(f"depths / num_heads must have the same length in 1..{SWINIR_MAX_LAYERS}")
    if cfg.upsampler not in UPSAMPLER_ID:
        raise ValueError(f"upsampler {cfg.upsampler!r} is not built (supported: '', 'pixelshuffle', 'nearest+conv')")
    s = IraSwinirConfig()
    s.upscale, s.in_chans, s.window_size, s.img_range = cfg.upscale, cfg.in_chans, cfg.window_size, float(cfg.img_range)
    s.n_layers, s.embed_dim, s.mlp_hidden = L, cfg.embed_dim, int(cfg.embed_dim * cfg.mlp_ratio)      # network_swinir.py:219
    for i in range(L):
        s.depths[i], s.num_heads[i] = cfg.depths[i], cfg.num_heads[i]
    s.upsampler, s.resi_3conv = UPSAMPLER_ID[cfg.upsampler], 1 if cfg.resi_connection == '3conv' else 0
    return s


def relative_position_index(ws: int) -> np.ndarray:
    """network_swinir.py:92-103."""
    coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing='ij')).reshape(2, -1)
    rel = (coords[:, :, None] - coords[:, None, :]).transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _pad2(a, rows, cols):
    out = np.zeros((rows, cols), np.float64)
    out[:a.shape[0], :a.shape[1]] = a
    return out


def _pad1(a, n):
    out = np.zeros(n, np.float32)
    out[:a.shape[0]] = a
    return out


def _conv3_pack_k(w, kp):
    """[N][K][3][3] -> [2][9][up32(N)/32][kp/16][64][8] with the input channels zero-padded to kp."""
    from .pack import frag_major, split16, up32
    n, k = w.shape[:2]
    out = np.zeros((9, up32(n), kp), dtype=np.float64)
    out[:, :n, :k] = w.reshape(n, k, 9).transpose(2, 0, 1)
    return frag_major(split16(out))


def pack_swinir_arrays(sd, cfg: SwinIRConfig) -> dict:
    """LayerNorm gains/biases folded into qkv / fc1; q rows pre-scaled by head_dim**-0.5 (:86,121); rows of qkv laid out
    [q|k|v][head][32] and the columns of proj [head][32] (head_dim 30 zero-padded to 32); embed_dim zero-padded to a
    multiple of 32 everywhere; relative position bias gathered to [heads][64][64]."""
    from .pack import frag_major, split16, up32, _np
    C, ws = cfg.embed_dim, cfg.window_size
    Cp, hid = up32(C), int(C * cfg.mlp_ratio)
    hidp, N = up32(hid), ws * ws
    f64 = lambda k: _np(sd[k]).astype(np.float64)
    out: dict = {}
    w = np.zeros((Cp, cfg.in_chans, 3, 3), np.float32)
    w[:C] = _np(sd['conv_first.weight'])
    out['first_w'] = w
    out['first_b'] = _pad1(_np(sd['conv_first.bias']), Cp)
    out['pe_norm_g'], out['pe_norm_b'] = _pad1(_np(sd['patch_embed.norm.weight']), Cp), _pad1(_np(sd['patch_embed.norm.bias']), Cp)
    out['norm_g'], out['norm_b'] = _pad1(_np(sd['norm.weight']), Cp), _pad1(_np(sd['norm.bias']), Cp)
    rpi = relative_position_index(ws)

    def resi(src, dst):
        if cfg.resi_connection == '1conv':
            out[f'{dst}_w'] = _conv3_pack_k(f64(f'{src}.weight'), Cp)
            out[f'{dst}_b'] = _pad1(_np(sd[f'{src}.bias']), Cp)
            return
        q = C // 4
        qp = up32(q)
        out[f'{dst}.0_w'] = _conv3_pack_k(f64(f'{src}.0.weight'), Cp)
        out[f'{dst}.0_b'] = _pad1(_np(sd[f'{src}.0.bias']), qp)
        out[f'{dst}.2_w'] = frag_major(split16(_pad2(f64(f'{src}.2.weight')[:, :, 0, 0], qp, qp)))
        out[f'{dst}.2_b'] = _pad1(_np(sd[f'{src}.2.bias']), qp)
        out[f'{dst}.4_w'] = _conv3_pack_k(f64(f'{src}.4.weight'), qp)
        out[f'{dst}.4_b'] = _pad1(_np(sd[f'{src}.4.bias']), Cp)

    for i, (depth, heads) in enumerate(zip(cfg.depths, cfg.num_heads)):
        d = C // heads
        if d > 32 or C % heads:
            raise ValueError(f"head_dim {C}/{heads} unsupported (must divide and be <= 32)")
        Kh = heads * 32
        for k in range(depth):
            p, o = f'layers.{i}.residual_group.blocks.{k}', f'l{i}.b{k}'
            g1, b1, g2, b2 = f64(f'{p}.norm1.weight'), f64(f'{p}.norm1.bias'), f64(f'{p}.norm2.weight'), f64(f'{p}.norm2.bias')
            wq, bq = f64(f'{p}.attn.qkv.weight'), f64(f'{p}.attn.qkv.bias')
            wfold, bfold = wq * g1[None, :], wq @ b1 + bq
            W = np.zeros((3 * Kh, Cp), np.float64)
            B = np.zeros(3 * Kh, np.float64)
            for part in range(3):
                sc = d ** -0.5 if part == 0 else 1.0
                for hd in range(heads):
                    src = slice(part * C + hd * d, part * C + (hd + 1) * d)
                    dst = slice((part * heads + hd) * 32, (part * heads + hd) * 32 + d)
                    W[dst, :C] = wfold[src] * sc
                    B[dst] = bfold[src] * sc
            out[f'{o}.qkv_w'] = frag_major(split16(W))
            out[f'{o}.qkv_b'] = B.astype(np.float32)
            tab = _np(sd[f'{p}.attn.relative_position_bias_table'])             # [(2ws-1)^2][heads]
            bt = np.zeros((heads, 64, 64), np.float32)
            bt[:, :N, :N] = tab[rpi.reshape(-1)].reshape(N, N, heads).transpose(2, 0, 1)
            out[f'{o}.bias_tab'] = bt
            wp = f64(f'{p}.attn.proj.weight')
            Wp_ = np.zeros((Cp, Kh), np.float64)
            for hd in range(heads):
                Wp_[:C, hd * 32:hd * 32 + d] = wp[:, hd * d:(hd + 1) * d]
            out[f'{o}.proj_w'] = frag_major(split16(Wp_))
            out[f'{o}.proj_b'] = _pad1(_np(sd[f'{p}.attn.proj.bias']), Cp)
            w1 = f64(f'{p}.mlp.fc1.weight')
            out[f'{o}.fc1_w'] = frag_major(split16(_pad2(w1 * g2[None, :], hidp, Cp)))
            out[f'{o}.fc1_b'] = _pad1((w1 @ b2 + f64(f'{p}.mlp.fc1.bias')).astype(np.float32), hidp)
            out[f'{o}.fc2_w'] = frag_major(split16(_pad2(f64(f'{p}.mlp.fc2.weight'), Cp, hidp)))
            out[f'{o}.fc2_b'] = _pad1(_np(sd[f'{p}.mlp.fc2.bias']), Cp)
        resi(f'layers.{i}.conv', f'l{i}.conv')
    resi('conv_after_body', 'cab')
    oc = cfg.in_chans
    if cfg.upsampler == '':
        out['last_w'] = _conv3_pack_k(f64('conv_last.weight'), Cp)
        out['last_b'] = _pad1(_np(sd['conv_last.bias']), up32(oc))
        return out
    out['cbu_w'] = _conv3_pack_k(f64('conv_before_upsample.0.weight'), Cp)
    out['cbu_b'] = _pad1(_np(sd['conv_before_upsample.0.bias']), 64)
    n_up = {1: 0, 2: 1, 4: 2}[cfg.upscale]
    for j in range(n_up):
        src = f'upsample.{2 * j}' if cfg.upsampler == 'pixelshuffle' else f'conv_up{j + 1}'
        wj = f64(f'{src}.weight')
        out[f'up{j}_w'] = _conv3_pack_k(wj, 64)
        out[f'up{j}_b'] = _pad1(_np(sd[f'{src}.bias']), up32(wj.shape[0]))
    if cfg.upsampler == 'nearest+conv':
        out['hr_w'] = _conv3_pack_k(f64('conv_hr.weight'), 64)
        out['hr_b'] = _pad1(_np(sd['conv_hr.bias']), 64)
    out['last_w'] = _conv3_pack_k(f64('conv_last.weight'), 64)
    out['last_b'] = _pad1(_np(sd['conv_last.bias']), up32(oc))
    # Ablation hook for tools/ablate_lo_swinir.py (measurement only, never set on a product path): IRA_ABLATE_LO='fc1_w,fc2_w' zeroes the
    # lo plane of the weights whose packed name ends in one of the suffixes, i.e. those layers then run with fp16 (hi plane) weights
    import os
    abl = [a for a in os.environ.get('IRA_ABLATE_LO', '').split(',') if a]
    for name, arr in out.items():
        if any(name.endswith(a) for a in abl) and arr.ndim >= 2 and arr.shape[0] == 2:
            arr[1] = 0
    return out


def pack_swinir_blob(sd, cfg: SwinIRConfig) -> bytes:
    from .pack import blob_from_arrays
    return blob_from_arrays(pack_swinir_arrays(sd, cfg))
