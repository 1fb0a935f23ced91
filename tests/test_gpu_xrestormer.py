"""X-Restormer on a real MI355X, through the C ABI: the overlapping cross-attention kernel against the oracle's
own OCAB pieces (xrestormer_arch.py:209-232, RelPosEmb :64-82), the network against the reference module's goldens
(tests/golden/make_golden_xrestormer.py), and the pad-crop harness against the oracle."""
import glob
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ira_amd.synth import synthetic_noisy_image
from ira_amd.xrestormer import XRestormerConfig, reduced_xrestormer_config, seeded_state_dict
from oracle import restormer_oracle as O
from oracle import xrestormer_oracle as X
from oracle.padcrop_oracle import padcrop_restore_image
from test_gpu_ops import close, dev, ptr, r16

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(glob.glob(os.path.join(GOLD, 'xrestormer_*.npz')))
_cache = {}


def backend(cfg, seed):
    from ira_amd.backend import XRestormerBackend
    key = (json.dumps(cfg.to_dict(), sort_keys=True), seed)
    if key not in _cache:
        if len(_cache) >= 2:
            _, old = _cache.popitem()
            old.close()
        _cache[key] = XRestormerBackend(cfg, seeded_state_dict(cfg, seed), device=0)
    return _cache[key]


@pytest.mark.parametrize('heads,T,H,W', [(1, 2, 8, 16), (2, 1, 24, 16), (4, 1, 16, 32), (8, 1, 8, 8)])
def test_ocab_attention(heads, T, H, W):
    """unfold(12, stride 8, zero pad 2) keys/values, q.k + relative-position logits, softmax, P.v, window merge."""
    be = backend(reduced_xrestormer_config(), 1)
    g = torch.Generator().manual_seed(heads * 10 + H)
    dh, ws, r = 16, 8, 12
    inner = heads * dh
    q = r16(torch.randn(T, inner, H, W, generator=g) * 0.5)
    k = r16(torch.randn(T, inner, H, W, generator=g))
    v = r16(torch.randn(T, inner, H, W, generator=g))
    rel_w = torch.randn(2 * r - 1, dh, generator=g) * 0.25
    rel_h = torch.randn(2 * r - 1, dh, generator=g) * 0.25
    nh, nw = H // ws, W // ws
    qs = q.view(T, inner, nh, ws, nw, ws).permute(0, 2, 4, 3, 5, 1).reshape(T * nh * nw, ws * ws, inner)
    unf = lambda t: F.unfold(t, kernel_size=(r, r), stride=ws, padding=2).view(T, inner, r * r, nh * nw).permute(0, 3, 2, 1).reshape(T * nh * nw, r * r, inner)
    split = lambda t: t.view(t.shape[0], t.shape[1], heads, dh).permute(0, 2, 1, 3).reshape(t.shape[0] * heads, t.shape[1], dh)
    qh, kh, vh = split(qs).double(), split(unf(k)).double(), split(unf(v)).double()
    attn = qh @ kh.transpose(-2, -1) + X.rel_logits(qh, rel_w.double(), rel_h.double(), ws, r)
    o = (attn.softmax(-1) @ vh).float()
    ref = o.view(T, nh, nw, heads, ws, ws, dh).permute(0, 3, 6, 1, 4, 2, 5).reshape(T, inner, H, W).permute(0, 2, 3, 1)
    qkv = torch.cat([q, k, v], 1).permute(0, 2, 3, 1).contiguous().half()
    rel = torch.zeros(2, 32, dh)
    rel[0, :23], rel[1, :23] = rel_w, rel_h
    out = torch.full((T, H, W, inner), float('nan'), dtype=torch.float16).cuda()
    qd, rd = dev(qkv), dev(rel)
    rc = be._lib.ira_op_ocab_attn(be.handle, ptr(qd), ptr(rd), ptr(out), T, H, W, heads)
    assert rc == 0, be._lib.ira_last_error(be.handle)
    close(out.cpu().float(), ref, 3e-3, atol_scale=2e-3)


def _u8(x):
    return O.img_as_ubyte(np.clip(x, 0, 1))


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(p)[11:-4] for p in CASES])
def test_forward_matches_reference_golden(path):
    z = np.load(path)
    cfg = XRestormerConfig(**json.loads(str(z['cfg'])))
    be = backend(cfg, int(z['seed']))
    y, ref = be.forward(z['input']), z['output']
    assert y.shape == ref.shape and np.isfinite(y).all()
    p = O.psnr_f(y, ref)
    x = z['input']
    if cfg.scale > 1:
        x = np.repeat(np.repeat(x, cfg.scale, 2), cfg.scale, 3)
    d = abs(O.psnr_u8(_u8(y), _u8(x)) - O.psnr_u8(_u8(ref), _u8(x)))
    print(f"{os.path.basename(path)}: PSNR(build, reference) {p:.2f} dB, max abs err {np.abs(y - ref).max():.3e}, |dPSNR_u8| {d:.2e}")
    assert p >= 58.0, p
    assert d <= 1e-3, d


def test_sr_preset_padcrop_harness():
    """executor/super_resolution/configs/xrestormer.yml: scale 4, padding_size 16, untiled; and a tiled run."""
    cfg = reduced_xrestormer_config(scale=4)
    be = backend(cfg, 4)
    sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(cfg, 4).items()}
    _, img = synthetic_noisy_image(27, 40, 3, sigma=10.0, seed=6)
    fwd = lambda t: X.xrestormer_forward(t, sd, cfg.to_dict())
    for ts, tp in ((None, 32), (16, 16)):
        got = be.restore_padcrop(img, tile_size=ts, tile_pad=tp, padding_size=16)
        ref = padcrop_restore_image(img, fwd, ts, tp, 16, scale=4)
        assert got.shape == ref.shape == (108, 160, 3)
        diff = np.abs(got.astype(int) - ref.astype(int))
        print(f"sr x4 tile={ts}: {100 * (diff > 0).mean():.2f}% px differ, max {diff.max()} LSB")
        assert diff.max() <= 1 and (diff > 0).mean() < 0.05


def test_padcrop_harness_and_errors():
    """xrestormer_model.py:53-136 (padding_size 64, tile 64 / pad 32 -> tiles of 96 and 128, not allowed: must be multiples of 64)."""
    cfg = reduced_xrestormer_config()
    be = backend(cfg, 1)
    sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(cfg, 1).items()}
    clean, img = synthetic_noisy_image(100, 150, 3, sigma=25.0, seed=2)
    got = be.restore_padcrop(img, tile_size=None, tile_pad=32, padding_size=64)            # untiled: 128 x 192
    ref = padcrop_restore_image(img, lambda t: X.xrestormer_forward(t, sd, cfg.to_dict()), None, 32, 64)
    diff = np.abs(got.astype(int) - ref.astype(int))
    d = abs(O.psnr_u8(got, clean) - O.psnr_u8(ref, clean))
    print(f"untiled: {100 * (diff > 0).mean():.2f}% px differ, max {diff.max()} LSB, |dPSNR| {d:.2e}")
    assert diff.max() <= 1 and (diff > 0).mean() < 0.05 and d <= 1e-3
    got = be.restore_padcrop(img, tile_size=64, tile_pad=64, padding_size=64)              # tiles of 128 / 192 px
    ref = padcrop_restore_image(img, lambda t: X.xrestormer_forward(t, sd, cfg.to_dict()), 64, 64, 64)
    diff = np.abs(got.astype(int) - ref.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.05
    from ira_amd._capi import IraError
    with pytest.raises(IraError):
        be.forward(np.zeros((1, 3, 72, 64), np.float32))            # 72 is not a multiple of 64
    x = np.random.default_rng(0).random((2, 3, 64, 64), dtype=np.float32)
    y = be.forward(x)
    assert np.array_equal(be.forward(x[1:2])[0], y[1])
