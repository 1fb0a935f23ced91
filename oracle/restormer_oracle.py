"""CPU ORACLE for the Restormer hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path (image-restoration-agent_amd/) never imports it and has no CPU fallback.

What it is: a functional restatement, on torch CPU tensors (fp32 by default, fp64 for a
high-precision yardstick), of the reference network
    AgenticIR/executor/defocus_deblurring/tools/Restormer/basicsr/models/archs/restormer_arch.py
and of the reference harness
    AgenticIR/executor/defocus_deblurring/tools/Restormer/demo.py:109-168
    (== AgentApp/model_service/denoising/Restormer/model_serving.py:158-209).
Every function cites the reference lines it follows.  It takes the weights as a plain
name -> tensor mapping with the reference's state-dict names, so the same seeded weights feed the
reference module (dev container only), this oracle, and the HIP backend.

Pinning: the reference has no golden vectors or unit tests for this path, so the
oracle is pinned against outputs of the reference module itself, imported in the dev container by
tests/golden/make_golden.py; the resulting fixtures are committed under tests/golden/ and checked by
tests/test_oracle.py (the reference never travels to the GPU box).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- network

def layer_norm(x, weight, bias, ln_type):
    """restormer_arch.py:25-70. x: (B,C,H,W); statistics over C per pixel, biased variance, eps 1e-5.

    BiasFree (:37-39): x / sqrt(var + 1e-5) * w  -- the mean is NOT subtracted in the numerator.
    WithBias (:54-57): (x - mu) / sqrt(var + 1e-5) * w + b.
    """
    mu = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, keepdim=True, unbiased=False)
    w = weight.view(1, -1, 1, 1)
    if ln_type == 'BiasFree':
        return x / torch.sqrt(var + 1e-5) * w
    return (x - mu) / torch.sqrt(var + 1e-5) * w + bias.view(1, -1, 1, 1)


def _b(sd, name):
    return sd.get(name, None)


def attention(x, sd, p, heads):
    """MDTA, restormer_arch.py:111-132."""
    b, c, h, w = x.shape
    qkv = F.conv2d(x, sd[f'{p}.qkv.weight'], _b(sd, f'{p}.qkv.bias'))                      # :114
    qkv = F.conv2d(qkv, sd[f'{p}.qkv_dwconv.weight'], _b(sd, f'{p}.qkv_dwconv.bias'),
                   padding=1, groups=3 * c)                                                  # :114
    q, k, v = qkv.chunk(3, dim=1)                                                            # :115
    q = q.reshape(b, heads, c // heads, h * w)                                               # :117-119
    k = k.reshape(b, heads, c // heads, h * w)
    v = v.reshape(b, heads, c // heads, h * w)
    q = F.normalize(q, dim=-1)                                                               # :121-122 (eps 1e-12)
    k = F.normalize(k, dim=-1)
    attn = (q @ k.transpose(-2, -1)) * sd[f'{p}.temperature'].view(1, heads, 1, 1)           # :124
    attn = attn.softmax(dim=-1)                                                              # :125
    out = (attn @ v).reshape(b, c, h, w)                                                     # :127-129
    return F.conv2d(out, sd[f'{p}.project_out.weight'], _b(sd, f'{p}.project_out.bias'))    # :131


def feed_forward(x, sd, p):
    """GDFN, restormer_arch.py:88-93 (exact-erf GELU, :91)."""
    hidden2 = sd[f'{p}.project_in.weight'].shape[0]
    x = F.conv2d(x, sd[f'{p}.project_in.weight'], _b(sd, f'{p}.project_in.bias'))
    x = F.conv2d(x, sd[f'{p}.dwconv.weight'], _b(sd, f'{p}.dwconv.bias'), padding=1, groups=hidden2)
    x1, x2 = x.chunk(2, dim=1)
    x = F.gelu(x1) * x2
    return F.conv2d(x, sd[f'{p}.project_out.weight'], _b(sd, f'{p}.project_out.bias'))


def transformer_block(x, sd, p, heads, ln_type):
    """restormer_arch.py:146-150."""
    x = x + attention(layer_norm(x, sd[f'{p}.norm1.body.weight'], _b(sd, f'{p}.norm1.body.bias'), ln_type),
                      sd, f'{p}.attn', heads)
    x = x + feed_forward(layer_norm(x, sd[f'{p}.norm2.body.weight'], _b(sd, f'{p}.norm2.body.bias'), ln_type),
                         sd, f'{p}.ffn')
    return x


def _stage(x, sd, name, n, heads, ln_type, taps=None):
    for i in range(n):
        x = transformer_block(x, sd, f'{name}.{i}', heads, ln_type)
        if taps is not None:
            taps[f'{name}.{i}'] = x
    return x


def downsample(x, w):
    """restormer_arch.py:171-179: conv3x3 (no bias) + PixelUnshuffle(2)."""
    return F.pixel_unshuffle(F.conv2d(x, w, padding=1), 2)


def upsample(x, w):
    """restormer_arch.py:181-189: conv3x3 (no bias) + PixelShuffle(2)."""
    return F.pixel_shuffle(F.conv2d(x, w, padding=1), 2)


def restormer_forward(inp_img, sd, cfg, taps=None):
    """restormer_arch.py:245-284.  cfg: dict with the constructor arguments (:193-205)."""
    nb, heads, ln = cfg['num_blocks'], cfg['heads'], cfg['LayerNorm_type']
    x1 = F.conv2d(inp_img, sd['patch_embed.proj.weight'], padding=1)                        # :247
    if taps is not None:
        taps['patch_embed'] = x1
    e1 = _stage(x1, sd, 'encoder_level1', nb[0], heads[0], ln, taps)                         # :248
    e2 = _stage(downsample(e1, sd['down1_2.body.0.weight']), sd, 'encoder_level2', nb[1], heads[1], ln, taps)
    e3 = _stage(downsample(e2, sd['down2_3.body.0.weight']), sd, 'encoder_level3', nb[2], heads[2], ln, taps)
    lat = _stage(downsample(e3, sd['down3_4.body.0.weight']), sd, 'latent', nb[3], heads[3], ln, taps)

    d3 = torch.cat([upsample(lat, sd['up4_3.body.0.weight']), e3], 1)                        # :259-260
    d3 = F.conv2d(d3, sd['reduce_chan_level3.weight'], _b(sd, 'reduce_chan_level3.bias'))    # :261
    d3 = _stage(d3, sd, 'decoder_level3', nb[2], heads[2], ln, taps)
    d2 = torch.cat([upsample(d3, sd['up3_2.body.0.weight']), e2], 1)                         # :264-265
    d2 = F.conv2d(d2, sd['reduce_chan_level2.weight'], _b(sd, 'reduce_chan_level2.bias'))    # :266
    d2 = _stage(d2, sd, 'decoder_level2', nb[1], heads[1], ln, taps)
    d1 = torch.cat([upsample(d2, sd['up2_1.body.0.weight']), e1], 1)                         # :269-270
    d1 = _stage(d1, sd, 'decoder_level1', nb[0], heads[0], ln, taps)                         # :271
    d1 = _stage(d1, sd, 'refinement', cfg['num_refinement_blocks'], heads[0], ln, taps)      # :273
    if cfg.get('dual_pixel_task', False):                                                    # :276-278
        d1 = d1 + F.conv2d(x1, sd['skip_conv.weight'], _b(sd, 'skip_conv.bias'))
        return F.conv2d(d1, sd['output.weight'], _b(sd, 'output.bias'), padding=1)
    return F.conv2d(d1, sd['output.weight'], _b(sd, 'output.bias'), padding=1) + inp_img     # :281


# ----------------------------------------------------------------------------- harness

def tile_index_list(size, tile, stride):
    """demo.py:140-141: list(range(0, size-tile, stride)) + [size-tile]."""
    return list(range(0, size - tile, stride)) + [size - tile]


def pad_amounts(height, width, multiple=8):
    """demo.py:123-126: pad up to a multiple of 8 only when not already a multiple."""
    H = ((height + multiple) // multiple) * multiple
    W = ((width + multiple) // multiple) * multiple
    padh = H - height if height % multiple != 0 else 0
    padw = W - width if width % multiple != 0 else 0
    return padh, padw


def img_as_ubyte(x):
    """skimage.img_as_ubyte on a float image in [0,1] (demo.py:161): rint(x*255) in the input float
    type (round-half-to-even), clip, cast."""
    x = np.asarray(x)
    y = np.rint(x * x.dtype.type(255))
    return np.clip(y, 0, 255).astype(np.uint8)


def restore_image(img_u8, sd, cfg, tile=None, tile_overlap=32, dtype=torch.float32, return_float=False,
                  forward_fn=None):
    """demo.py:120-161 on an (h, w, c) uint8 array -> (h, w, c) uint8 array.

    forward_fn: optional replacement for the network call ``model(x)`` (tests/golden/make_golden.py
    passes the imported reference module here, so the harness fixtures are produced BY the reference
    network under this restatement of the harness script, which itself cannot be imported: cv2,
    skimage, natsort are missing offline).
    tile=None: whole image (:129-131).  Otherwise the overlap-average tiler (:132-153): E += model(patch),
    W += 1 in row-major tile order, restored = E / W.
    """
    img_u8 = np.asarray(img_u8)
    assert img_u8.dtype == np.uint8 and img_u8.ndim == 3
    input_ = torch.from_numpy(img_u8.copy()).float().div(255.).permute(2, 0, 1).unsqueeze(0)   # :120
    input_ = input_.to(dtype)
    if forward_fn is None:
        sd = {k: v.to(dtype) for k, v in sd.items()}
        forward_fn = lambda t: restormer_forward(t, sd, cfg)   # noqa: E731
    height, width = input_.shape[2], input_.shape[3]
    padh, padw = pad_amounts(height, width)                                                  # :123-126
    input_ = F.pad(input_, (0, padw, 0, padh), 'reflect')                                    # :127
    with torch.no_grad():
        if tile is None:
            restored = forward_fn(input_)                                                    # :131
        else:
            b, c, h, w = input_.shape
            tile = min(tile, h, w)                                                           # :135
            assert tile % 8 == 0, "tile size should be multiple of 8"                        # :136
            stride = tile - tile_overlap                                                     # :139
            h_idx_list = tile_index_list(h, tile, stride)
            w_idx_list = tile_index_list(w, tile, stride)
            E = torch.zeros(b, c, h, w, dtype=dtype)
            Wt = torch.zeros_like(E)
            for h_idx in h_idx_list:                                                         # :145-152
                for w_idx in w_idx_list:
                    in_patch = input_[..., h_idx:h_idx + tile, w_idx:w_idx + tile]
                    out_patch = forward_fn(in_patch)
                    E[..., h_idx:h_idx + tile, w_idx:w_idx + tile].add_(out_patch)
                    Wt[..., h_idx:h_idx + tile, w_idx:w_idx + tile].add_(torch.ones_like(out_patch))
            restored = E.div_(Wt)                                                            # :153
    restored = torch.clamp(restored, 0, 1)                                                   # :155
    restored = restored[:, :, :height, :width]                                               # :158
    restored = restored.permute(0, 2, 3, 1).cpu().numpy()[0]                                 # :160
    if return_float:
        return restored
    return img_as_ubyte(restored.astype(np.float32))                                         # :161


# ----------------------------------------------------------------------------- metric

def psnr_u8(a_u8, b_u8):
    """RST/Denoising/utils.py:10-25: float64 MSE on [0,255] images, 20*log10(255/sqrt(mse))."""
    a = np.asarray(a_u8, dtype=np.float64)
    b = np.asarray(b_u8, dtype=np.float64)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    return 20.0 * math.log10(255.0 / math.sqrt(mse))


def psnr_f(a, b, peak=1.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    return 10.0 * math.log10(peak * peak / mse)


def synthetic_noisy_image(h, w, c=3, sigma=25.0, seed=0):
    """Synthetic input of SURVEY 8d (shared with bench.py): see ira_amd/synth.py."""
    from ira_amd.synth import synthetic_noisy_image as f
    return f(h, w, c, sigma, seed)
