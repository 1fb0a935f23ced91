"""X-Restormer configuration, state-dict layout, seeded weights and packer (reference
AgenticIR/executor/dehazing/tools/X-Restormer/xrestormer/archs/xrestormer_arch.py:190-422, test presets
AgentApp/model_service/dehazing/X-Restormer/X-Restormer/options/test/00{2,3,4,5}_xrestormer_*.yml).

X-Restormer = the Restormer U-Net whose blocks are [channel attention (MDTA) + GDFN, overlapping cross-attention (OCAB) +
GDFN] (:239-262).  scale > 1 (the SR preset: bilinear x4 of the input first, :378-379) is not built."""
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
