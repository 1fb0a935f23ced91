"""CPU ORACLE for SwinIR -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional torch-CPU restatement of AgenticIR/executor/super_resolution/tools/SwinIR/models/network_swinir.py (network) and
.../SwinIR/inference.py:67-90,294-322 (harness: always-pad by flip-concat, overlap-average tiler with scale factor).
Pinned against outputs of the reference module imported in the dev container with a 3-symbol ``timm.models.layers`` stub
(tests/golden/make_golden_swinir.py).  The harness *script* needs cv2 and cannot be imported: its arithmetic is restated
here around the network (uint8 in, uint8 out, RGB order kept: the BGR<->RGB flips of inference.py:71,88 cancel)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

RGB_MEAN = (0.4488, 0.4371, 0.4040)


def window_partition(x, ws):
    """network_swinir.py:33-46.  (B,H,W,C) -> (nW*B, ws, ws, C)."""
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(windows, ws, H, W):
    """:49-62."""
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def relative_position_index(ws):
    """:92-103."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing='ij'))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def calculate_mask(H, W, ws, shift):
    """:216-237."""
    img_mask = torch.zeros((1, H, W, 1))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img_mask[:, h, w, :] = cnt
            cnt += 1
    mw = window_partition(img_mask, ws).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))


def window_attention(x, sd, p, ws, heads, mask):
    """WindowAttention.forward :114-145.  x: (nW*B, N, C)."""
    B_, N, C = x.shape
    qkv = F.linear(x, sd[f'{p}.qkv.weight'], sd[f'{p}.qkv.bias']).reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (C // heads) ** -0.5
    attn = q @ k.transpose(-2, -1)
    bias = sd[f'{p}.relative_position_bias_table'][relative_position_index(ws).view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, N, N)
    attn = attn.softmax(-1)
    x = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(x, sd[f'{p}.proj.weight'], sd[f'{p}.proj.bias'])


def swin_block(x, x_size, sd, p, ws, shift, heads):
    """SwinTransformerBlock.forward :239-279 (tokens (B, HW, C))."""
    H, W = x_size
    B, L, C = x.shape
    shortcut = x
    x = F.layer_norm(x, (C,), sd[f'{p}.norm1.weight'], sd[f'{p}.norm1.bias'], 1e-5).view(B, H, W, C)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(x, ws).view(-1, ws * ws, C)
    aw = window_attention(xw, sd, f'{p}.attn', ws, heads, calculate_mask(H, W, ws, shift) if shift > 0 else None)
    x = window_reverse(aw.view(-1, ws, ws, C), ws, H, W)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = shortcut + x.view(B, H * W, C)
    y = F.layer_norm(x, (C,), sd[f'{p}.norm2.weight'], sd[f'{p}.norm2.bias'], 1e-5)
    y = F.linear(F.gelu(F.linear(y, sd[f'{p}.mlp.fc1.weight'], sd[f'{p}.mlp.fc1.bias'])), sd[f'{p}.mlp.fc2.weight'], sd[f'{p}.mlp.fc2.bias'])   # Mlp :24-30
    return x + y


def resi_conv(x, sd, p, kind):
    """'1conv' / '3conv' (:462-469, :727-735)."""
    if kind == '1conv':
        return F.conv2d(x, sd[f'{p}.weight'], sd[f'{p}.bias'], padding=1)
    x = F.leaky_relu(F.conv2d(x, sd[f'{p}.0.weight'], sd[f'{p}.0.bias'], padding=1), 0.2)
    x = F.leaky_relu(F.conv2d(x, sd[f'{p}.2.weight'], sd[f'{p}.2.bias']), 0.2)
    return F.conv2d(x, sd[f'{p}.4.weight'], sd[f'{p}.4.bias'], padding=1)


def forward_features(x, sd, cfg):
    """:790-803 with RSTB.forward :481-482 and BasicLayer :397-404 (ape False, patch_norm True)."""
    B, C, H, W = x.shape
    ws = cfg['window_size']
    t = x.flatten(2).transpose(1, 2)                                                        # PatchEmbed :524-528
    t = F.layer_norm(t, (C,), sd['patch_embed.norm.weight'], sd['patch_embed.norm.bias'], 1e-5)
    for i, (depth, heads) in enumerate(zip(cfg['depths'], cfg['num_heads'])):
        t0 = t
        for k in range(depth):
            t = swin_block(t, (H, W), sd, f'layers.{i}.residual_group.blocks.{k}', ws, 0 if k % 2 == 0 else ws // 2, heads)   # :385-386
        img = t.transpose(1, 2).view(B, C, H, W)                                            # PatchUnEmbed :562-565
        t = resi_conv(img, sd, f'layers.{i}.conv', cfg['resi_connection']).flatten(2).transpose(1, 2) + t0
    t = F.layer_norm(t, (C,), sd['norm.weight'], sd['norm.bias'], 1e-5)
    return t.transpose(1, 2).view(B, C, H, W)


def swinir_forward(x, sd, cfg):
    """SwinIR.forward :805-840."""
    H, W = x.shape[2:]
    ws, up, r = cfg['window_size'], cfg['upscale'], cfg['img_range']
    x = F.pad(x, (0, (ws - W % ws) % ws, 0, (ws - H % ws) % ws), 'reflect')                 # check_image_size :783-788
    mean = torch.tensor(RGB_MEAN).view(1, 3, 1, 1).to(x) if cfg['in_chans'] == 3 else torch.zeros(1, 1, 1, 1).to(x)
    x = (x - mean) * r
    kind = cfg['resi_connection']
    xf = F.conv2d(x, sd['conv_first.weight'], sd['conv_first.bias'], padding=1)
    body = resi_conv(forward_features(xf, sd, cfg), sd, 'conv_after_body', kind) + xf
    if cfg['upsampler'] == 'pixelshuffle':
        y = F.leaky_relu(F.conv2d(body, sd['conv_before_upsample.0.weight'], sd['conv_before_upsample.0.bias'], padding=1), 0.01)
        s, j = up, 0
        while s > 1:
            y = F.pixel_shuffle(F.conv2d(y, sd[f'upsample.{j}.weight'], sd[f'upsample.{j}.bias'], padding=1), 2)
            s //= 2
            j += 2
        y = F.conv2d(y, sd['conv_last.weight'], sd['conv_last.bias'], padding=1)
    elif cfg['upsampler'] == 'nearest+conv':
        y = F.leaky_relu(F.conv2d(body, sd['conv_before_upsample.0.weight'], sd['conv_before_upsample.0.bias'], padding=1), 0.01)
        y = F.leaky_relu(F.conv2d(F.interpolate(y, scale_factor=2, mode='nearest'), sd['conv_up1.weight'], sd['conv_up1.bias'], padding=1), 0.2)
        if up == 4:
            y = F.leaky_relu(F.conv2d(F.interpolate(y, scale_factor=2, mode='nearest'), sd['conv_up2.weight'], sd['conv_up2.bias'], padding=1), 0.2)
        y = F.conv2d(F.leaky_relu(F.conv2d(y, sd['conv_hr.weight'], sd['conv_hr.bias'], padding=1), 0.2), sd['conv_last.weight'], sd['conv_last.bias'], padding=1)
    else:
        y = x + F.conv2d(body, sd['conv_last.weight'], sd['conv_last.bias'], padding=1)
    y = y / r + mean
    return y[:, :, :H * up, :W * up]


# ------------------------------------------------------------------------------------------------ harness
def harness_pad(img, ws):
    """inference.py:76-80: ALWAYS pads (h//ws + 1)*ws - h rows by flip-concat, even when h is a multiple of ws."""
    _, _, h, w = img.shape
    hp, wp = (h // ws + 1) * ws - h, (w // ws + 1) * ws - w
    img = torch.cat([img, torch.flip(img, [2])], 2)[:, :, :h + hp, :]
    return torch.cat([img, torch.flip(img, [3])], 3)[:, :, :, :w + wp]


def tile_index_list(n, tile, stride):
    return list(range(0, n - tile, stride)) + [n - tile]


def swinir_test(img, forward_fn, scale, ws, tile=None, tile_overlap=32):
    """inference.py:294-322."""
    if tile is None:
        return forward_fn(img)
    b, c, h, w = img.shape
    tile = min(tile, h, w)
    assert tile % ws == 0, "tile size should be a multiple of window_size"
    stride = tile - tile_overlap
    E = torch.zeros(b, c, h * scale, w * scale).type_as(img)
    Wt = torch.zeros_like(E)
    for hi in tile_index_list(h, tile, stride):
        for wi in tile_index_list(w, tile, stride):
            out = forward_fn(img[..., hi:hi + tile, wi:wi + tile])
            E[..., hi * scale:(hi + tile) * scale, wi * scale:(wi + tile) * scale].add_(out)
            Wt[..., hi * scale:(hi + tile) * scale, wi * scale:(wi + tile) * scale].add_(torch.ones_like(out))
    return E.div_(Wt)


def swinir_restore_image(img_u8, sd, cfg, tile=None, tile_overlap=32, forward_fn=None):
    """uint8 (h,w,c) -> uint8 (h*s, w*s, c): inference.py:69-90."""
    fwd = forward_fn or (lambda t: swinir_forward(t, sd, cfg))
    x = torch.from_numpy(np.ascontiguousarray(img_u8)).float().div(255.).permute(2, 0, 1).unsqueeze(0)   # get_image_pair: /255
    _, _, h, w = x.shape
    with torch.no_grad():
        y = swinir_test(harness_pad(x, cfg['window_size']), fwd, cfg['upscale'], cfg['window_size'], tile, tile_overlap)
    y = y[..., :h * cfg['upscale'], :w * cfg['upscale']]
    y = y[0].float().clamp_(0, 1).permute(1, 2, 0).numpy()
    return (y * 255.0).round().astype(np.uint8)
