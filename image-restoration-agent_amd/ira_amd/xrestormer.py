"""X-Restormer configuration, state-dict layout, seeded weights and packer (reference
AgenticIR/executor/dehazing/tools/X-Restormer/xrestormer/archs/xrestormer_arch.py:190-422, test presets
AgentApp/model_service/dehazing/X-Restormer/X-Restormer/options/test/00{2,3,4,5}_xrestormer_*.yml).

X-Restormer = the Restormer U-Net whose blocks are [channel attention (MDTA) + GDFN, overlapping cross-attention (OCAB) +
GDFN] (:239-262).  scale > 1 (the SR preset): the input is upsampled bilinearly first (:378-379) and the network runs at the output size."""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass, field, asdict

import numpy as np


@dataclass
class XRestormerConfig:
    inp_channels: int = 3
    out_channels: int = 3
    dim: int = 48
    num_blocks: list = field(default_factory=lambda: [2, 4, 4, 4])
    num_refinement_blocks: int = 4
    channel_heads: list = field(default_factory=lambda: [1, 2, 4, 8])
    spatial_heads: list = field(default_factory=lambda: [1, 2, 4, 8])
    overlap_ratio: list = field(default_factory=lambda: [0.5, 0.5, 0.5, 0.5])
    window_size: int = 8
    spatial_dim_head: int = 16
    bias: bool = False
    ffn_expansion_factor: float = 2.66
    LayerNorm_type: str = 'WithBias'
    dual_pixel_task: bool = False
    scale: int = 1

    def to_dict(self):
        return asdict(self)

    def hidden(self, dim):
        return int(dim * self.ffn_expansion_factor)       # xrestormer_arch.py:137

    def overlap_win(self, level):
        return int(self.window_size * self.overlap_ratio[level]) + self.window_size      # :196


def reduced_xrestormer_config(**kw):
    d = dict(dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, channel_heads=[1, 2, 4, 8], spatial_heads=[1, 2, 2, 4])
    d.update(kw)
    return XRestormerConfig(**d)


def _stages(cfg):
    """(stage name, number of blocks, dim, level index for heads / overlap) in module registration order (:349-374)."""
    d, nb = cfg.dim, cfg.num_blocks
    return [('encoder_level1', nb[0], d, 0), ('encoder_level2', nb[1], 2 * d, 1), ('encoder_level3', nb[2], 4 * d, 2), ('latent', nb[3], 8 * d, 3),
            ('decoder_level3', nb[2], 4 * d, 2), ('decoder_level2', nb[1], 2 * d, 1), ('decoder_level1', nb[0], 2 * d, 0),
            ('refinement', cfg.num_refinement_blocks, 2 * d, 0)]


def _block_spec(p, dim, cfg, lvl):
    hid, inner, r = cfg.hidden(dim), cfg.spatial_dim_head * cfg.spatial_heads[lvl], cfg.overlap_win(lvl)
    b = cfg.bias
    wb = cfg.LayerNorm_type != 'BiasFree'
    out = []

    def conv(name, o, i, k, groups=1):
        out.append((f'{p}.{name}.weight', (o, i // groups, k, k)))
        if b:
            out.append((f'{p}.{name}.bias', (o,)))

    def ffn(name):
        conv(f'{name}.project_in', 2 * hid, dim, 1)
        conv(f'{name}.dwconv', 2 * hid, 2 * hid, 3, groups=2 * hid)
        conv(f'{name}.project_out', dim, hid, 1)
    conv('spatial_attn.qkv', 3 * inner, dim, 1)
    conv('spatial_attn.project_out', dim, inner, 1)
    out += [(f'{p}.spatial_attn.rel_pos_emb.rel_height', (2 * r - 1, cfg.spatial_dim_head)),
            (f'{p}.spatial_attn.rel_pos_emb.rel_width', (2 * r - 1, cfg.spatial_dim_head))]
    out.append((f'{p}.channel_attn.temperature', (cfg.channel_heads[lvl], 1, 1)))
    conv('channel_attn.qkv', 3 * dim, dim, 1)
    conv('channel_attn.qkv_dwconv', 3 * dim, 3 * dim, 3, groups=3 * dim)
    conv('channel_attn.project_out', dim, dim, 1)
    for n in ('norm1', 'norm2', 'norm3', 'norm4'):
        out.append((f'{p}.{n}.body.weight', (dim,)))
        if wb:
            out.append((f'{p}.{n}.body.bias', (dim,)))
    ffn('channel_ffn')
    ffn('spatial_ffn')
    return out


def state_dict_spec(cfg: XRestormerConfig):
    """(name, shape) in the order of the reference module's state_dict() (:341-374)."""
    d = cfg.dim
    st = {s[0]: s for s in _stages(cfg)}
    spec = [('patch_embed.proj.weight', (d, cfg.inp_channels, 3, 3))]

    def stage(name):
        _, n, dim, lvl = st[name]
        out = []
        for i in range(n):
            out += _block_spec(f'{name}.{i}', dim, cfg, lvl)
        return out
    spec += stage('encoder_level1') + [('down1_2.body.0.weight', (d // 2, d, 3, 3))]
    spec += stage('encoder_level2') + [('down2_3.body.0.weight', (d, 2 * d, 3, 3))]
    spec += stage('encoder_level3') + [('down3_4.body.0.weight', (2 * d, 4 * d, 3, 3))]
    spec += stage('latent') + [('up4_3.body.0.weight', (16 * d, 8 * d, 3, 3)), ('reduce_chan_level3.weight', (4 * d, 8 * d, 1, 1))]
    if cfg.bias:
        spec.append(('reduce_chan_level3.bias', (4 * d,)))
    spec += stage('decoder_level3') + [('up3_2.body.0.weight', (8 * d, 4 * d, 3, 3)), ('reduce_chan_level2.weight', (2 * d, 4 * d, 1, 1))]
    if cfg.bias:
        spec.append(('reduce_chan_level2.bias', (2 * d,)))
    spec += stage('decoder_level2') + [('up2_1.body.0.weight', (4 * d, 2 * d, 3, 3))]
    spec += stage('decoder_level1') + stage('refinement') + [('output.weight', (cfg.out_channels, 2 * d, 3, 3))]
    if cfg.bias:
        spec.append(('output.bias', (cfg.out_channels,)))
    return spec


def seeded_tensor(seed, name, shape):
    rng = np.random.default_rng([int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())])
    if name.endswith('temperature'):
        return rng.uniform(1.0, 6.0, shape).astype(np.float32)
    if 'rel_height' in name or 'rel_width' in name:
        return (rng.standard_normal(shape) * shape[1] ** -0.5).astype(np.float32)          # RelPosEmb init (:58-62)
    if '.norm' in name and name.endswith('body.weight'):
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if name.endswith('.bias'):
        return (0.05 * rng.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    std = 1.0 / np.sqrt(fan_in)
    if 'spatial_attn.qkv' in name:
        std *= 2.0                                       # logits of O(1)
    if name == 'output.weight':
        std *= 0.05
    return (std * rng.standard_normal(shape)).astype(np.float32)


def seeded_state_dict(cfg: XRestormerConfig, seed: int = 0):
    return OrderedDict((n, seeded_tensor(seed, n, s)) for n, s in state_dict_spec(cfg))


# ---------------------------------------------------------------------------------------------------------------
# packer: reference state_dict -> device blob.  The Restormer part (U-Net glue, channel attention, first GDFN) goes through
# ira_amd.pack with the key names translated; the OCAB and the second GDFN are added per block.

def restormer_equivalent(cfg: XRestormerConfig):
    from .config import RestormerConfig
    return RestormerConfig(inp_channels=cfg.inp_channels, out_channels=cfg.out_channels, dim=cfg.dim, num_blocks=list(cfg.num_blocks),
                           num_refinement_blocks=cfg.num_refinement_blocks, heads=list(cfg.channel_heads),
                           ffn_expansion_factor=cfg.ffn_expansion_factor, bias=cfg.bias, LayerNorm_type=cfg.LayerNorm_type,
                           dual_pixel_task=cfg.dual_pixel_task)


def xrestormer_cfg_struct(cfg: XRestormerConfig):
    from ._capi import IraXrestormerConfig
    from .backend import _cfg_struct
    if cfg.scale not in (1, 2, 4):
        raise ValueError("scale must be 1, 2 or 4")
    s = IraXrestormerConfig()
    s.base = _cfg_struct(restormer_equivalent(cfg))
    for l in range(4):
        s.spatial_heads[l] = cfg.spatial_heads[l]
        s.overlap_win[l] = cfg.overlap_win(l)
    s.window_size, s.spatial_dim_head, s.scale = cfg.window_size, cfg.spatial_dim_head, cfg.scale
    return s


def pack_xrestormer_arrays(sd, cfg: XRestormerConfig) -> dict:
    from .pack import pack_arrays, frag_major, split16, up32, _np, _pad_rows, _hidden_remap, _dw_pack, _dw_unit_pack, _dw_unit_bias
    rcfg = restormer_equivalent(cfg)
    tr = {}
    for k, v in sd.items():
        if '.spatial_attn.' in k or '.spatial_ffn.' in k or '.norm3.' in k or '.norm4.' in k:
            continue
        tr[k.replace('.channel_attn.', '.attn.').replace('.channel_ffn.', '.ffn.')] = v
    out = pack_arrays(tr, rcfg)
    wb = cfg.LayerNorm_type != 'BiasFree'
    dh = cfg.spatial_dim_head
    f64 = lambda k: _np(sd[k]).astype(np.float64)
    for name, n, dim, lvl in _stages(cfg):
        heads = cfg.spatial_heads[lvl]
        inner, hid, r = dh * heads, cfg.hidden(dim), cfg.overlap_win(lvl)
        hp = up32(hid)
        for i in range(n):
            p, o = f'{name}.{i}', f'{name}.{i}'
            g3, g4 = f64(f'{p}.norm3.body.weight'), f64(f'{p}.norm4.body.weight')
            wq = f64(f'{p}.spatial_attn.qkv.weight')[:, :, 0, 0]                      # [3*inner][dim]: q | k | v
            scale = np.ones(3 * inner)
            scale[:inner] = dh ** -0.5                                                 # xrestormer_arch.py:199,225
            out[f'{o}.oc_qkv_w'] = frag_major(split16(_pad_rows(wq * g3[None, :] * scale[:, None], up32(3 * inner))))
            bq = np.zeros(3 * inner)
            has_b = False
            if wb:
                bq += wq @ f64(f'{p}.norm3.body.bias'); has_b = True
            if cfg.bias:
                bq += f64(f'{p}.spatial_attn.qkv.bias'); has_b = True
            if has_b:
                out[f'{o}.oc_qkv_b'] = _pad_rows((bq * scale).astype(np.float32), up32(3 * inner))
            rel = np.zeros((2, 32, dh), np.float32)
            rel[0, :2 * r - 1] = _np(sd[f'{p}.spatial_attn.rel_pos_emb.rel_width'])
            rel[1, :2 * r - 1] = _np(sd[f'{p}.spatial_attn.rel_pos_emb.rel_height'])
            out[f'{o}.oc_rel'] = rel
            wo = np.zeros((up32(dim), inner), np.float64)
            wo[:dim] = f64(f'{p}.spatial_attn.project_out.weight')[:, :, 0, 0]
            out[f'{o}.oc_proj_w'] = frag_major(split16(wo))
            wi = f64(f'{p}.spatial_ffn.project_in.weight')[:, :, 0, 0]
            out[f'{o}.pin2_w'] = frag_major(split16(_hidden_remap(wi * g4[None, :], hid, hp)))
            bi = np.zeros(2 * hid)
            has_b = False
            if wb:
                bi += wi @ f64(f'{p}.norm4.body.bias'); has_b = True
            if cfg.bias:
                bi += f64(f'{p}.spatial_ffn.project_in.bias'); has_b = True
            if has_b:
                out[f'{o}.pin2_b'] = _hidden_remap(bi.astype(np.float32), hid, hp)
            out[f'{o}.dw3_w'] = _dw_pack(_hidden_remap(_np(sd[f'{p}.spatial_ffn.dwconv.weight']), hid, hp))
            out[f'{o}.dw3_u'] = _dw_unit_pack(out[f'{o}.dw3_w'], hp)
            wout = np.zeros((up32(dim), hp), np.float64)
            wout[:dim, :hid] = f64(f'{p}.spatial_ffn.project_out.weight')[:, :, 0, 0]
            out[f'{o}.pout2_w'] = frag_major(split16(wout))
            if cfg.bias:
                out[f'{o}.oc_proj_b'] = _pad_rows(_np(sd[f'{p}.spatial_attn.project_out.bias']).astype(np.float32), up32(dim))
                out[f'{o}.dw3_b'] = _hidden_remap(_np(sd[f'{p}.spatial_ffn.dwconv.bias']).astype(np.float32), hid, hp)
                out[f'{o}.dw3_ub'] = _dw_unit_bias(out[f'{o}.dw3_b'], hp)
                out[f'{o}.pout2_b'] = _pad_rows(_np(sd[f'{p}.spatial_ffn.project_out.bias']).astype(np.float32), up32(dim))
    return out


def pack_xrestormer_blob(sd, cfg: XRestormerConfig) -> bytes:
    from .pack import blob_from_arrays
    return blob_from_arrays(pack_xrestormer_arrays(sd, cfg))
