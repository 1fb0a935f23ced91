"""CPU ORACLE for X-Restormer -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional torch-CPU restatement of AgenticIR/executor/dehazing/tools/X-Restormer/xrestormer/archs/xrestormer_arch.py
(RelPosEmb :49-82, OCAB :190-236, TransformerBlock :239-262, XRestormer :324-422); LayerNorm, GDFN, MDTA and the U-Net
plumbing are byte-for-byte Restormer's and are taken from oracle/restormer_oracle.py.  Pinned against outputs of the reference
module imported with ``basicsr.utils.registry`` / ``torchstat`` stubs (tests/golden/make_golden_xrestormer.py).  Its harness
(the pad-crop tiler of xrestormer_model.py:53-136) is oracle/padcrop_oracle.py."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import restormer_oracle as R


def rel_logits(q, rel_w, rel_h, ws, r):
    """RelPosEmb.forward (:64-82) written as a direct gather.  q: (B, ws*ws, d) (already scaled); rel_*: (2r-1, d).
    The skew trick of rel_to_abs (:24-36) maps relative index j - i + r - 1 to key position j for query position i, so
    logits[q=(yi,xi)][k=(yj,xj)] = q . rel_w[xj - xi + r - 1] + q . rel_h[yj - yi + r - 1]."""
    B = q.shape[0]
    lw = q @ rel_w.t()                                   # (B, 64, 2r-1)
    lh = q @ rel_h.t()
    qi = torch.arange(ws)
    kj = torch.arange(r)
    idx = kj[None, :] - qi[:, None] + r - 1              # (ws, r)
    yi, xi = torch.meshgrid(qi, qi, indexing='ij')       # query coordinates
    yi, xi = yi.reshape(-1), xi.reshape(-1)
    w_abs = torch.gather(lw, 2, idx[xi][None].expand(B, -1, -1))      # (B, 64, r): by xj
    h_abs = torch.gather(lh, 2, idx[yi][None].expand(B, -1, -1))      # (B, 64, r): by yj
    return (h_abs[:, :, :, None] + w_abs[:, :, None, :]).reshape(B, ws * ws, r * r)


def ocab(x, sd, p, ws, r, heads, dh):
    """OCAB.forward (:209-236).  x: (B, C, H, W), H and W multiples of ws."""
    B, C, H, W = x.shape
    inner = heads * dh
    qkv = F.conv2d(x, sd[f'{p}.qkv.weight'], sd.get(f'{p}.qkv.bias'))
    qs, ks, vs = qkv.chunk(3, dim=1)
    nh, nw = H // ws, W // ws
    qs = qs.view(B, inner, nh, ws, nw, ws).permute(0, 2, 4, 3, 5, 1).reshape(B * nh * nw, ws * ws, inner)
    unf = lambda t: F.unfold(t, kernel_size=(r, r), stride=ws, padding=(r - ws) // 2)       # zeros outside the image (:200)
    kv = []
    for t in (ks, vs):
        u = unf(t).view(B, inner, r * r, nh * nw).permute(0, 3, 2, 1).reshape(B * nh * nw, r * r, inner)
        kv.append(u)
    split = lambda t: t.view(t.shape[0], t.shape[1], heads, dh).permute(0, 2, 1, 3).reshape(t.shape[0] * heads, t.shape[1], dh)
    q, k, v = split(qs) * dh ** -0.5, split(kv[0]), split(kv[1])
    attn = q @ k.transpose(-2, -1) + rel_logits(q, sd[f'{p}.rel_pos_emb.rel_width'], sd[f'{p}.rel_pos_emb.rel_height'], ws, r)
    out = attn.softmax(-1) @ v                                                             # (B*nwin*heads, 64, dh)
    out = out.view(B, nh, nw, heads, ws, ws, dh).permute(0, 3, 6, 1, 4, 2, 5).reshape(B, inner, H, W)
    return F.conv2d(out, sd[f'{p}.project_out.weight'], sd.get(f'{p}.project_out.bias'))


def _ln(x, sd, p, ln_type):
    return R.layer_norm(x, sd[f'{p}.body.weight'], sd.get(f'{p}.body.bias'), ln_type)


def _ffn(x, sd, p):
    x = F.conv2d(x, sd[f'{p}.project_in.weight'], sd.get(f'{p}.project_in.bias'))
    x = F.conv2d(x, sd[f'{p}.dwconv.weight'], sd.get(f'{p}.dwconv.bias'), padding=1, groups=x.shape[1])
    x1, x2 = x.chunk(2, dim=1)
    return F.conv2d(F.gelu(x1) * x2, sd[f'{p}.project_out.weight'], sd.get(f'{p}.project_out.bias'))


def _channel_attn(x, sd, p, heads):
    """ChannelAttention.forward (:165-186) = Restormer's MDTA."""
    b, c, h, w = x.shape
    qkv = F.conv2d(x, sd[f'{p}.qkv.weight'], sd.get(f'{p}.qkv.bias'))
    qkv = F.conv2d(qkv, sd[f'{p}.qkv_dwconv.weight'], sd.get(f'{p}.qkv_dwconv.bias'), padding=1, groups=3 * c)
    q, k, v = (t.reshape(b, heads, c // heads, h * w) for t in qkv.chunk(3, dim=1))
    q, k = F.normalize(q, dim=-1), F.normalize(k, dim=-1)
    attn = ((q @ k.transpose(-2, -1)) * sd[f'{p}.temperature']).softmax(-1)
    return F.conv2d((attn @ v).reshape(b, c, h, w), sd[f'{p}.project_out.weight'], sd.get(f'{p}.project_out.bias'))


def block(x, sd, p, cfg, lvl):
    """TransformerBlock.forward (:256-262)."""
    ln = cfg['LayerNorm_type']
    ws, r = cfg['window_size'], int(cfg['window_size'] * cfg['overlap_ratio'][lvl]) + cfg['window_size']
    x = x + _channel_attn(_ln(x, sd, f'{p}.norm1', ln), sd, f'{p}.channel_attn', cfg['channel_heads'][lvl])
    x = x + _ffn(_ln(x, sd, f'{p}.norm2', ln), sd, f'{p}.channel_ffn')
    x = x + ocab(_ln(x, sd, f'{p}.norm3', ln), sd, f'{p}.spatial_attn', ws, r, cfg['spatial_heads'][lvl], cfg['spatial_dim_head'])
    x = x + _ffn(_ln(x, sd, f'{p}.norm4', ln), sd, f'{p}.spatial_ffn')
    return x


def xrestormer_forward(inp, sd, cfg):
    """XRestormer.forward (:376-410).  H*scale, W*scale multiples of 8 * window_size (the harness pads: padding_size 64, or 16 at scale 4)."""
    nb = cfg['num_blocks']
    if cfg.get('scale', 1) > 1:
        inp = F.interpolate(inp, scale_factor=cfg['scale'], mode='bilinear', align_corners=False)      # :378-379

    def stage(x, name, n, lvl):
        for i in range(n):
            x = block(x, sd, f'{name}.{i}', cfg, lvl)
        return x
    conv3 = lambda x, k: F.conv2d(x, sd[k], None, padding=1)
    e1 = stage(conv3(inp, 'patch_embed.proj.weight'), 'encoder_level1', nb[0], 0)
    e2 = stage(F.pixel_unshuffle(conv3(e1, 'down1_2.body.0.weight'), 2), 'encoder_level2', nb[1], 1)
    e3 = stage(F.pixel_unshuffle(conv3(e2, 'down2_3.body.0.weight'), 2), 'encoder_level3', nb[2], 2)
    lat = stage(F.pixel_unshuffle(conv3(e3, 'down3_4.body.0.weight'), 2), 'latent', nb[3], 3)
    d3 = torch.cat([F.pixel_shuffle(conv3(lat, 'up4_3.body.0.weight'), 2), e3], 1)
    d3 = stage(F.conv2d(d3, sd['reduce_chan_level3.weight'], sd.get('reduce_chan_level3.bias')), 'decoder_level3', nb[2], 2)
    d2 = torch.cat([F.pixel_shuffle(conv3(d3, 'up3_2.body.0.weight'), 2), e2], 1)
    d2 = stage(F.conv2d(d2, sd['reduce_chan_level2.weight'], sd.get('reduce_chan_level2.bias')), 'decoder_level2', nb[1], 1)
    d1 = torch.cat([F.pixel_shuffle(conv3(d2, 'up2_1.body.0.weight'), 2), e1], 1)
    d1 = stage(d1, 'decoder_level1', nb[0], 0)
    d1 = stage(d1, 'refinement', cfg['num_refinement_blocks'], 0)
    return F.conv2d(d1, sd['output.weight'], sd.get('output.bias'), padding=1) + inp
