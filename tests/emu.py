"""Numerics-contract emulator of the HIP pipeline (TEST INFRASTRUCTURE).

Same algebra and the same rounding points as image-restoration-agent_amd/csrc ("Numerics
contract"): fp16 storage of activations, fp16 MFMA operands, fp32 accumulation / statistics /
softmax / GELU, LayerNorm gain folded into the following 1x1 weights, attention applied as a
per-tile matrix  M = W_o . blockdiag(softmax)  on V.  Used to (a) choose the residual-stream dtype
against the 1e-3 dB bar on CPU and (b) give the per-kernel GPU tests a tighter comparator than the
fp32 oracle.  Runs on torch CPU; fp32 tensors hold the fp16-rounded values.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def r16(t):
    return t.half().float()


class Contract:
    """Which storage points round to fp16.  ``off`` = set of tags left un-rounded (ablations)."""

    def __init__(self, res_fp16=True, act_fp16=True, off=(), only=None, on=()):
        self.res_fp16 = res_fp16
        self.act_fp16 = act_fp16
        self.off = set(off)
        self.only = None if only is None else set(only)
        self.on = set(on)

    def _on(self, tag):
        if self.only is not None:
            return tag in self.only
        if tag.startswith('w_') and 'w' in self.off and tag not in self.on:
            return False                     # the per-layer weight tags follow the global 'w' switch unless switched on by name
        return tag not in self.off

    def res(self, t, tag='res'):
        return r16(t) if (self.res_fp16 and self._on(tag)) else t

    def act(self, t, tag='act'):
        return r16(t) if (self.act_fp16 and self._on(tag)) else t

    def w(self, t, tag='w'):
        return r16(t) if self._on(tag) else t


DEEP_HI_PLANE = ['w_pin_deep', 'w_qkv_deep', 'w_pout_deep', 'w_M_deep']


def GPU_CONTRACT(dim=None):
    """What csrc/ implements: fp32 residual stream, fp16 branch activations, weights as fp16 hi + lo
    planes (~exact), dense 3x3 inputs split hi/lo in registers (~exact); with the default weight_planes = 1 the 1x1 layers of the levels
    with 192 / 384 channels (levels 3-4 when dim = 48) run on plain fp16 weights."""
    return Contract(res_fp16=False, act_fp16=True, off=['w', 'M', 'c3w', 'c3in', 'dw1w', 'dw2w'], on=DEEP_HI_PLANE if dim == 48 else ())


def ln_core(x, ln_type):
    mu = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, keepdim=True, unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    return x * rstd if ln_type == 'BiasFree' else (x - mu) * rstd


def fold_ln(w, lnw, lnb, ct, tag='w'):
    """conv1x1(LN(x)) = conv1x1_{W*diag(lnw)}(core(x)) + W@lnb."""
    w2 = w[:, :, 0, 0]
    wf = ct.w(w2 * lnw.view(1, -1), tag)
    bias = (w2 @ lnb) if lnb is not None else None
    return wf, bias


def conv1x1(x, w2, bias=None):
    y = torch.einsum('bchw,oc->bohw', x, w2)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return y


def attention_mix(q, k, temperature, wo, heads):
    """Per-tile matrix M (C x C) with  project_out(attn @ v) == conv1x1(v, M)."""
    b, c, h, w = q.shape
    ch = c // heads
    qf = q.reshape(b, heads, ch, h * w)
    kf = k.reshape(b, heads, ch, h * w)
    gram = qf @ kf.transpose(-2, -1)
    nq = qf.pow(2).sum(-1).sqrt().clamp_min(1e-12)
    nk = kf.pow(2).sum(-1).sqrt().clamp_min(1e-12)
    logits = gram / (nq.unsqueeze(-1) * nk.unsqueeze(-2)) * temperature.view(1, heads, 1, 1)
    attn = logits.softmax(-1)                                   # (b, heads, ch, ch)
    wo2 = wo[:, :, 0, 0].view(1, c, heads, ch)                  # out, head, ch_in
    m = torch.einsum('bohi,bhij->bohj', wo2.expand(b, -1, -1, -1), attn).reshape(b, c, c)
    return m


def block(x, sd, p, heads, ln_type, ct: Contract):
    c = x.shape[1]
    lnb1 = sd.get(f'{p}.norm1.body.bias')
    # per-layer weight tags carry a '_deep' suffix on levels 3-4 (encoder_level3, latent, decoder_level3): the lo-plane ablation per level
    dp = '_deep' if ('level3' in p or p.startswith('latent')) else ''
    wq, bq = fold_ln(sd[f'{p}.attn.qkv.weight'], sd[f'{p}.norm1.body.weight'], lnb1, ct, 'w_qkv' + dp)
    xn = ct.act(ln_core(x, ln_type), 'xn1')
    qkv_pre = ct.act(conv1x1(xn, wq, bq), 'qkv_pre')
    qkv = F.conv2d(qkv_pre, ct.w(sd[f'{p}.attn.qkv_dwconv.weight'], 'dw1w'), padding=1, groups=3 * c)
    q, k, v = qkv.chunk(3, dim=1)
    q, k, v = ct.act(q, 'qk'), ct.act(k, 'qk'), ct.act(v, 'v')
    m = ct.w(attention_mix(q, k, sd[f'{p}.attn.temperature'], sd[f'{p}.attn.project_out.weight'], heads), 'M' if not dp else 'w_M_deep')
    x = ct.res(x + torch.einsum('bchw,boc->bohw', v, m), 'res_attn')

    lnb2 = sd.get(f'{p}.norm2.body.bias')
    wi, bi = fold_ln(sd[f'{p}.ffn.project_in.weight'], sd[f'{p}.norm2.body.weight'], lnb2, ct, 'w_pin' + dp)
    xn = ct.act(ln_core(x, ln_type), 'xn2')
    hp = ct.act(conv1x1(xn, wi, bi), 'hp')
    hd = F.conv2d(hp, ct.w(sd[f'{p}.ffn.dwconv.weight'], 'dw2w'), padding=1, groups=hp.shape[1])
    h1, h2 = hd.chunk(2, dim=1)
    g = ct.act(F.gelu(h1) * h2, 'g')
    x = ct.res(x + conv1x1(g, ct.w(sd[f'{p}.ffn.project_out.weight'][:, :, 0, 0], 'w_pout' + dp)), 'res_ffn')
    return x


def conv3x3(x, w, ct, deep=False):
    # deep: the convolutions between levels 2-3-4 (down2_3, down3_4, up4_3, up3_2) carry their own weight tag for the per-level ablation
    return F.conv2d(ct.act(x, 'c3in'), ct.w(w, 'w_c3_deep' if deep else 'c3w'), padding=1)


def forward(inp, sd, cfg, ct: Contract, taps=None):
    nb, heads, ln = cfg['num_blocks'], cfg['heads'], cfg['LayerNorm_type']

    def stage(x, name, n, h):
        for i in range(n):
            x = block(x, sd, f'{name}.{i}', h, ln, ct)
            if taps is not None:
                taps[f'{name}.{i}'] = x
        return x

    inp16 = ct.act(inp)
    x1 = ct.res(F.conv2d(inp16, sd['patch_embed.proj.weight'], padding=1))
    if taps is not None:
        taps['patch_embed'] = x1
    e1 = stage(x1, 'encoder_level1', nb[0], heads[0])
    e2 = stage(ct.res(F.pixel_unshuffle(conv3x3(e1, sd['down1_2.body.0.weight'], ct), 2)), 'encoder_level2', nb[1], heads[1])
    e3 = stage(ct.res(F.pixel_unshuffle(conv3x3(e2, sd['down2_3.body.0.weight'], ct, True), 2)), 'encoder_level3', nb[2], heads[2])
    lat = stage(ct.res(F.pixel_unshuffle(conv3x3(e3, sd['down3_4.body.0.weight'], ct, True), 2)), 'latent', nb[3], heads[3])

    u3 = ct.act(F.pixel_shuffle(conv3x3(lat, sd['up4_3.body.0.weight'], ct, True), 2))
    d3 = ct.res(conv1x1(torch.cat([u3, ct.act(e3)], 1), ct.w(sd['reduce_chan_level3.weight'][:, :, 0, 0])))
    d3 = stage(d3, 'decoder_level3', nb[2], heads[2])
    u2 = ct.act(F.pixel_shuffle(conv3x3(d3, sd['up3_2.body.0.weight'], ct, True), 2))
    d2 = ct.res(conv1x1(torch.cat([u2, ct.act(e2)], 1), ct.w(sd['reduce_chan_level2.weight'][:, :, 0, 0])))
    d2 = stage(d2, 'decoder_level2', nb[1], heads[1])
    u1 = ct.res(F.pixel_shuffle(conv3x3(d2, sd['up2_1.body.0.weight'], ct), 2))
    d1 = torch.cat([u1, e1], 1)
    d1 = stage(d1, 'decoder_level1', nb[0], heads[0])
    d1 = stage(d1, 'refinement', cfg['num_refinement_blocks'], heads[0])
    return conv3x3(d1, sd['output.weight'], ct) + inp
