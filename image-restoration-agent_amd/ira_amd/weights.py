"""State-dict layout of the reference Restormer and a seeded generator for it.

No pretrained weight is available offline (every checkpoint is fetched at image-build
time), so parity and throughput runs use deterministic build-generated weights.  The generator is
keyed by tensor *name*, so the reference module in the dev container, the CPU oracle and the HIP
backend on the GPU box all see bit-identical fp32 tensors from (seed, name, shape) alone.

Key names follow restormer_arch.py:193-243 (checked against the reference's ``state_dict()`` by
tests/test_weights.py through the committed fixture tests/golden/restormer_state_dict_spec.json).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

from .config import RestormerConfig


def _block_spec(prefix: str, dim: int, hidden: int, heads: int, with_bias_ln: bool, bias: bool):
    out = []
    out.append((f'{prefix}.norm1.body.weight', (dim,)))
    if with_bias_ln:
        out.append((f'{prefix}.norm1.body.bias', (dim,)))
    out.append((f'{prefix}.attn.temperature', (heads, 1, 1)))
    out.append((f'{prefix}.attn.qkv.weight', (dim * 3, dim, 1, 1)))
    if bias:
        out.append((f'{prefix}.attn.qkv.bias', (dim * 3,)))
    out.append((f'{prefix}.attn.qkv_dwconv.weight', (dim * 3, 1, 3, 3)))
    if bias:
        out.append((f'{prefix}.attn.qkv_dwconv.bias', (dim * 3,)))
    out.append((f'{prefix}.attn.project_out.weight', (dim, dim, 1, 1)))
    if bias:
        out.append((f'{prefix}.attn.project_out.bias', (dim,)))
    out.append((f'{prefix}.norm2.body.weight', (dim,)))
    if with_bias_ln:
        out.append((f'{prefix}.norm2.body.bias', (dim,)))
    out.append((f'{prefix}.ffn.project_in.weight', (hidden * 2, dim, 1, 1)))
    if bias:
        out.append((f'{prefix}.ffn.project_in.bias', (hidden * 2,)))
    out.append((f'{prefix}.ffn.dwconv.weight', (hidden * 2, 1, 3, 3)))
    if bias:
        out.append((f'{prefix}.ffn.dwconv.bias', (hidden * 2,)))
    out.append((f'{prefix}.ffn.project_out.weight', (dim, hidden, 1, 1)))
    if bias:
        out.append((f'{prefix}.ffn.project_out.bias', (dim,)))
    return out


def state_dict_spec(cfg: RestormerConfig) -> list[tuple[str, tuple[int, ...]]]:
    """(name, shape) pairs in the order of the reference module's ``state_dict()``."""
    d = cfg.dim
    wb = cfg.LayerNorm_type != 'BiasFree'       # restormer_arch.py:63-66
    b = cfg.bias
    nb, heads = cfg.num_blocks, cfg.heads
    spec: list = []

    def stage(name, n, dim, h):
        for i in range(n):
            spec.extend(_block_spec(f'{name}.{i}', dim, cfg.hidden(dim), h, wb, b))

    spec.append(('patch_embed.proj.weight', (d, cfg.inp_channels, 3, 3)))
    stage('encoder_level1', nb[0], d, heads[0])
    spec.append(('down1_2.body.0.weight', (d // 2, d, 3, 3)))
    stage('encoder_level2', nb[1], d * 2, heads[1])
    spec.append(('down2_3.body.0.weight', (d, d * 2, 3, 3)))
    stage('encoder_level3', nb[2], d * 4, heads[2])
    spec.append(('down3_4.body.0.weight', (d * 2, d * 4, 3, 3)))
    stage('latent', nb[3], d * 8, heads[3])
    spec.append(('up4_3.body.0.weight', (d * 16, d * 8, 3, 3)))
    spec.append(('reduce_chan_level3.weight', (d * 4, d * 8, 1, 1)))
    if b:
        spec.append(('reduce_chan_level3.bias', (d * 4,)))
    stage('decoder_level3', nb[2], d * 4, heads[2])
    spec.append(('up3_2.body.0.weight', (d * 8, d * 4, 3, 3)))
    spec.append(('reduce_chan_level2.weight', (d * 2, d * 4, 1, 1)))
    if b:
        spec.append(('reduce_chan_level2.bias', (d * 2,)))
    stage('decoder_level2', nb[1], d * 2, heads[1])
    spec.append(('up2_1.body.0.weight', (d * 4, d * 2, 3, 3)))
    stage('decoder_level1', nb[0], d * 2, heads[0])
    stage('refinement', cfg.num_refinement_blocks, d * 2, heads[0])
    if cfg.dual_pixel_task:
        spec.append(('skip_conv.weight', (d * 2, d, 1, 1)))
        if b:
            spec.append(('skip_conv.bias', (d * 2,)))
    spec.append(('output.weight', (cfg.out_channels, d * 2, 3, 3)))
    if b:
        spec.append(('output.bias', (cfg.out_channels,)))
    return spec


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())])


def seeded_tensor(seed: int, name: str, shape: tuple[int, ...]) -> np.ndarray:
    """Deterministic fp32 tensor for one state-dict entry.

    Scales keep every branch O(1) so the residual stream neither dies nor saturates:
    convs ~ N(0, 1/fan_in); LN gains ~ 1 +- 0.1; LN / conv biases ~ 0.1 N; temperature in [1, 6]
    (a trained Restormer has non-uniform attention: give the softmax something to do); the last
    conv is scaled down so the predicted residual is of the size a denoiser produces (~0.1 rms).
    """
    rng = _rng(seed, name)
    if name.endswith('temperature'):
        return rng.uniform(1.0, 6.0, size=shape).astype(np.float32)
    if '.norm' in name and name.endswith('.weight'):
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if name.endswith('.bias'):
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    std = 1.0 / np.sqrt(fan_in)
    if name == 'output.weight':
        std *= 0.05
    return (std * rng.standard_normal(shape)).astype(np.float32)


def seeded_state_dict(cfg: RestormerConfig, seed: int = 0, residual_scale: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """``residual_scale`` multiplies every branch's last 1x1 (attention and GDFN ``project_out``): the size of what each block adds to the
    residual stream relative to the stream itself (the parity sweep of tests/golden/make_golden_sweep.py runs x0.25 / x1 / x4)."""
    sd = OrderedDict((n, seeded_tensor(seed, n, s)) for n, s in state_dict_spec(cfg))
    if residual_scale != 1.0:
        for n in sd:
            if n.endswith('.project_out.weight'):
                sd[n] = (sd[n] * np.float32(residual_scale)).astype(np.float32)
    return sd


def to_torch(sd_np):
    import torch
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd_np.items())
