"""SwinIR on a real MI355X, through the C ABI: the window-attention / LayerNorm kernels and the new
options of the LN+1x1 kernel against plain torch statements (network_swinir.py cited per test), the whole network against
the goldens produced by the reference module (tests/golden/make_golden_swinir.py) and the harness against the oracle."""
import ctypes as C
import glob
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ira_amd.swinir import SwinIRConfig, reduced_swinir_config, seeded_state_dict, relative_position_index
from ira_amd.synth import synthetic_noisy_image
from oracle import restormer_oracle as O
from oracle import swinir_oracle as S
from test_gpu_ops import close, dev, ptr, r16, rowperm_pad, split16, up32

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(glob.glob(os.path.join(GOLD, 'swinir_*.npz')))

_cache = {}


def backend(cfg, seed):
    from ira_amd.backend import SwinIRBackend
    key = (json.dumps(cfg.to_dict(), sort_keys=True), seed)
    if key not in _cache:
        if len(_cache) >= 2:
            _, old = _cache.popitem()
            old.close()
        _cache[key] = SwinIRBackend(cfg, seeded_state_dict(cfg, seed), device=0)
    return _cache[key]


@pytest.fixture(scope='module')
def ctx():
    return backend(reduced_swinir_config(), 1)


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize('ws,shift,heads,d,T,H,W', [(8, 0, 2, 30, 2, 16, 24), (8, 4, 6, 30, 1, 24, 16), (7, 3, 2, 30, 1, 14, 21),
                                                    (7, 0, 3, 16, 2, 7, 14), (8, 4, 8, 30, 1, 8, 8), (8, 4, 1, 32, 1, 32, 8)])
def test_window_attention(ctx, ws, shift, heads, d, T, H, W):
    """roll + window_partition + (q k^T + bias + mask) softmax v + window_reverse + roll back (:114-145,216-237,249-271)."""
    g = torch.Generator().manual_seed(ws * 100 + shift + heads)
    C_ = heads * d
    N = ws * ws
    q, k, v = (r16(torch.randn(T, H, W, heads, d, generator=g)) for _ in range(3))
    q = r16(q * 0.4)
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5
    rpi = torch.from_numpy(relative_position_index(ws))
    bias = table[rpi.view(-1)].view(N, N, heads).permute(2, 0, 1).contiguous()
    # reference: the oracle's own pieces
    def win(t):
        t = t.reshape(T, H, W, C_)
        if shift:
            t = torch.roll(t, (-shift, -shift), (1, 2))
        return S.window_partition(t, ws).view(-1, N, heads, d).permute(0, 2, 1, 3)
    attn = win(q).double() @ win(k).double().transpose(-2, -1) + bias.double().unsqueeze(0)
    if shift:
        m = S.calculate_mask(H, W, ws, shift).double()
        nW = m.shape[0]
        attn = (attn.view(T, nW, heads, N, N) + m.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    o = (attn.softmax(-1) @ win(v).double()).transpose(1, 2).reshape(-1, ws, ws, C_)
    o = S.window_reverse(o.float(), ws, H, W)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    # device layout: [q|k|v][head][32]
    qkv = torch.zeros(T, H, W, 3, heads, 32)
    qkv[..., 0, :, :d], qkv[..., 1, :, :d], qkv[..., 2, :, :d] = q, k, v
    bt = torch.zeros(heads, 64, 64)
    bt[:, :N, :N] = bias
    out = torch.full((T, H, W, heads, 32), float('nan'), dtype=torch.float16).cuda()
    qd, bd = dev(qkv.half()), dev(bt)
    rc = ctx._lib.ira_op_swin_attn(ctx.handle, ptr(qd), ptr(bd), ptr(out), T, H, W, heads, ws, shift)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    got = out.cpu().float()
    assert torch.isfinite(got).all()
    close(got[..., :d].reshape(T, H, W, C_), o, 3e-3, atol_scale=2e-3)
    if d < 32:
        assert (got[..., d:] == 0).all()


@pytest.mark.parametrize('C_,Cp,npx', [(180, 192, 77), (240, 256, 50), (60, 64, 9), (64, 64, 33)])
def test_layernorm_affine(ctx, C_, Cp, npx):
    """nn.LayerNorm(embed_dim), eps 1e-5 (patch_embed.norm :524-528, self.norm :800) on the zero-padded token stream."""
    g = torch.Generator().manual_seed(C_)
    x = torch.zeros(npx, Cp)
    x[:, :C_] = torch.randn(npx, C_, generator=g) * 2 + 0.5
    gw, gb = torch.zeros(Cp), torch.zeros(Cp)
    gw[:C_], gb[:C_] = torch.randn(C_, generator=g) * 0.1 + 1, torch.randn(C_, generator=g) * 0.1
    ref = torch.zeros(npx, Cp)
    ref[:, :C_] = F.layer_norm(x[:, :C_], (C_,), gw[:C_], gb[:C_], 1e-5)
    y = torch.full((npx, Cp), float('nan')).cuda()
    xd, a, b = dev(x), dev(gw), dev(gb)
    rc = ctx._lib.ira_op_ln_affine(ctx.handle, ptr(xd), ptr(a), ptr(b), ptr(y), npx, C_, Cp, 1e-5)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    close(y.cpu(), ref, 2e-5, atol_scale=1e-5)


@pytest.mark.parametrize('C_,Cp,N,act,T,P', [(180, 192, 576, 0, 2, 70), (240, 256, 480, 1, 1, 45), (60, 64, 128, 1, 2, 33), (180, 192, 384, 1, 1, 64)])
def test_ln_linear_padded_gelu(ctx, C_, Cp, N, act, T, P):
    """LN over the true C of a zero-padded K + Linear (+ exact GELU): norm1+qkv (:246,116), norm2+fc1+act (:277, Mlp :24-27)."""
    g = torch.Generator().manual_seed(C_ + N)
    x = torch.zeros(T * P, Cp)
    x[:, :C_] = torch.randn(T * P, C_, generator=g) * 1.5 + 0.3
    Wm = torch.zeros(N, Cp)
    Wm[:, :C_] = torch.randn(N, C_, generator=g) / C_ ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    n = torch.zeros(T * P, Cp)
    n[:, :C_] = F.layer_norm(x[:, :C_], (C_,), None, None, 1e-5)
    ref = (r16(n).double() @ Wm.double().t()).float() + bias
    if act:
        ref = F.gelu(ref)
    NB = up32(N) // 32
    y = torch.full((T * P, N), float('nan'), dtype=torch.float16).cuda()
    xd, Wd, bd = dev(x), dev(split16(rowperm_pad(Wm, NB * 32))), dev(rowperm_pad(bias[:, None], NB * 32)[:, 0])
    rc = ctx._lib.ira_op_pw_kres_ex(ctx.handle, ptr(xd), 1, 2, Cp, ptr(Wd), ptr(bd), ptr(y), N, N, NB, T, P, 1e-5, C_, act)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    close(y.cpu().float(), ref, 2e-3)


# ------------------------------------------------------------------------------------------------ network + harness
def _u8(x):
    return O.img_as_ubyte(np.clip(x, 0, 1))


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(p)[7:-4] for p in CASES])
def test_forward_matches_reference_golden(path):
    z = np.load(path)
    cfg = SwinIRConfig(**json.loads(str(z['cfg'])))
    be = backend(cfg, int(z['seed']))
    y, ref = be.forward(z['input']), z['output']
    assert y.shape == ref.shape and np.isfinite(y).all()
    p = O.psnr_f(y, ref)
    x = z['input']
    if cfg.upscale > 1:
        x = np.repeat(np.repeat(x, cfg.upscale, 2), cfg.upscale, 3)
    d = abs(O.psnr_u8(_u8(y), _u8(x)) - O.psnr_u8(_u8(ref), _u8(x)))
    print(f"{os.path.basename(path)}: PSNR(build, reference) {p:.2f} dB, max abs err {np.abs(y - ref).max():.3e}, |dPSNR_u8| {d:.2e}")
    assert p >= 58.0, p
    assert d <= 1e-3, d


@pytest.mark.parametrize('name,cfgf,seed,h,w,tile,ov', [
    ('dn_whole', lambda: reduced_swinir_config(), 3, 40, 56, None, 32),
    ('dn_tiled_odd', lambda: reduced_swinir_config(), 3, 45, 61, 24, 8),
    ('jpeg_w7_gray_tiled', lambda: reduced_swinir_config(in_chans=1, window_size=7, img_size=28, img_range=255.), 4, 37, 50, 21, 7),
    ('sr_x4_3conv_tiled', lambda: reduced_swinir_config(upscale=4, upsampler='nearest+conv', resi_connection='3conv', embed_dim=64, num_heads=[2, 2]), 5, 30, 41, 16, 8),
    ('sr_x2_pixelshuffle', lambda: reduced_swinir_config(upscale=2, upsampler='pixelshuffle'), 6, 24, 33, None, 32),
])
def test_harness_matches_oracle(name, cfgf, seed, h, w, tile, ov):
    """SwinIR/inference.py:69-90,294-322: always-pad flip-concat, whole / overlap-average tiler with scale, clamp, round."""
    cfg = cfgf()
    be = backend(cfg, seed)
    sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(cfg, seed).items()}
    clean, img = synthetic_noisy_image(h, w, cfg.in_chans, sigma=25.0, seed=seed)
    got = be.restore(img, tile=tile, tile_overlap=ov)
    ref = S.swinir_restore_image(img, sd, cfg.to_dict(), tile, ov)
    assert got.shape == ref.shape == (h * cfg.upscale, w * cfg.upscale, cfg.in_chans)
    diff = np.abs(got.astype(int) - ref.astype(int))
    anchor = np.repeat(np.repeat(clean, cfg.upscale, 0), cfg.upscale, 1)
    d = abs(O.psnr_u8(got, anchor) - O.psnr_u8(ref, anchor))
    print(f"{name}: {100 * (diff > 0).mean():.2f}% px differ, max {diff.max()} LSB, |dPSNR| {d:.2e}")
    assert diff.max() <= 1 and (diff > 0).mean() < 0.05 and d <= 1e-3


def test_tile_parallel_restorer_world1_scaled():
    """ira_amd/dist.py on the SwinIR harness (model-aware plan, x4 tiles): world 1 equals the one-call harness."""
    from ira_amd.dist import TileParallelRestorer
    cfg = reduced_swinir_config(upscale=4, upsampler='nearest+conv', resi_connection='3conv', embed_dim=64, num_heads=[2, 2])
    be = backend(cfg, 5)
    _, img = synthetic_noisy_image(37, 52, 3, sigma=25.0, seed=8)
    assert be.plan(37, 52, 16, 8) == (40, 56, 16, 16, 4, 6, 4)
    out = TileParallelRestorer(be, 0, 1).restore_frame(torch.from_numpy(img).cuda(), 16, 8).cpu().numpy()
    assert out.shape == (148, 208, 3) and np.array_equal(out, be.restore(img, tile=16, tile_overlap=8))


def test_batch_and_errors():
    cfg = reduced_swinir_config()
    be = backend(cfg, 1)
    x = np.random.default_rng(0).random((3, 3, 24, 32), dtype=np.float32)
    y = be.forward(x)
    assert np.array_equal(be.forward(x[1:2])[0], y[1])
    from ira_amd._capi import IraError
    with pytest.raises(IraError):
        be.restore(np.zeros((40, 40, 3), np.uint8), tile=20)         # not a multiple of the window (inference.py:302)
    with pytest.raises(IraError):
        be.forward(np.zeros((1, 1, 16, 16), np.float32))
