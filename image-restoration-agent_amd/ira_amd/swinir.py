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
