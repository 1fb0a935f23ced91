"""Pad-crop tiler (xrestormer_model.py:53-136): oracle properties on CPU, HIP path against the oracle
on the GPU (reduced Restormer and the build's DnCNN as the networks)."""
import glob
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ira_amd.synth import synthetic_noisy_image
from oracle.padcrop_oracle import padcrop_restore, padcrop_restore_image
from oracle import restormer_oracle as O


GOLD = os.path.join(os.path.dirname(__file__), 'golden')
PADCROP = sorted(glob.glob(os.path.join(GOLD, 'padcrop_*.npz')))


def _fixture_net(z):
    """The oracle network (CPU restatement, itself pinned by its own fixtures) with the fixture's seeded weights."""
    cfgd, seed = json.loads(str(z['cfg'])), int(z['seed'])
    if str(z['kind']) == 'xrestormer':
        from ira_amd.xrestormer import XRestormerConfig, seeded_state_dict
        from oracle.xrestormer_oracle import xrestormer_forward
        cfg = XRestormerConfig(**cfgd)
        sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(cfg, seed).items()}
        return cfg, seeded_state_dict(cfg, seed), lambda t: xrestormer_forward(t, sd, cfg.to_dict())
    from ira_amd.rrdbnet import RRDBNetConfig, seeded_state_dict
    from oracle.rrdbnet_oracle import rrdbnet_forward
    cfg = RRDBNetConfig(**cfgd)
    sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(cfg, seed).items()}
    return cfg, seeded_state_dict(cfg, seed), lambda t: rrdbnet_forward(t, sd, cfg.to_dict())


@pytest.mark.parametrize('path', PADCROP, ids=[os.path.basename(p)[8:-4] for p in PADCROP])
def test_oracle_matches_the_references_own_tiler(path):
    """PIN of row a12: the fixtures are outputs of XRestormerModel.pre_process / tile_process / test / post_process
    (xrestormer_model.py:53-150) and of RealESRGANer.pre_process / process / tile_process / post_process (realesrganer.py:71-183),
    executed by tests/golden/make_golden_padcrop.py with the reference networks."""
    from oracle.padcrop_oracle import realesrganer_restore
    z = np.load(path)
    cfg, _, net = _fixture_net(z)
    x = torch.from_numpy(z['noisy']).float().div(255.).permute(2, 0, 1).unsqueeze(0)
    ts, tp = int(z['tile_size']), int(z['tile_pad'])
    with torch.no_grad():
        if str(z['kind']) == 'xrestormer':
            y = padcrop_restore(x, net, None if ts < 0 else ts, tp, int(z['padding_size']), cfg.scale)
        else:
            y = realesrganer_restore(x, net, cfg.scale, ts, tp, int(z['pre_pad']))
    ref = torch.from_numpy(z['output'])
    assert y.shape == ref.shape
    err = (y - ref).abs().max().item()
    print(f"{os.path.basename(path)}: max |oracle - reference tiler| = {err:.2e}")
    assert err < 2e-4 * max(1.0, ref.abs().max().item())


def test_identity_network_returns_the_image():
    _, img = synthetic_noisy_image(70, 101, 3, sigma=25.0, seed=1)
    for ts, tp, ps in ((None, 32, 64), (32, 8, 64), (48, 16, 16), (64, 0, 8)):
        assert np.array_equal(padcrop_restore_image(img, lambda t: t, ts, tp, ps), img)


def test_tiled_equals_untiled_for_a_local_network_with_enough_halo():
    """A 5x5 (two chained 3x3) zero-padded conv sees 2 pixels of context: with tile_pad >= 2 every core pixel whose
    window stays inside its haloed tile matches the untiled result; at frame borders both paths zero-pad identically."""
    g = torch.Generator().manual_seed(0)
    w1, w2 = torch.randn(4, 3, 3, 3, generator=g) * 0.2, torch.randn(3, 4, 3, 3, generator=g) * 0.2
    net = lambda t: F.conv2d(F.conv2d(t, w1, padding=1), w2, padding=1)
    x = torch.rand(1, 3, 75, 90, generator=g)
    a = padcrop_restore(x, net, None, 0, 16)
    b = padcrop_restore(x, net, 32, 2, 16)
    assert torch.allclose(a, b, atol=1e-6)
    c = padcrop_restore(x, net, 32, 1, 16)           # halo too small: must differ somewhere
    assert not torch.allclose(a, c, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('h,w,ts,tp,ps', [(70, 101, 32, 8, 64), (64, 64, None, 32, 64), (100, 60, 48, 16, 8), (33, 47, 16, 8, 16)])
def test_gpu_restormer_padcrop_matches_oracle(h, w, ts, tp, ps):
    from ira_amd.backend import RestormerBackend
    from ira_amd.config import reduced_config
    from ira_amd.weights import seeded_state_dict, to_torch
    cfg = reduced_config('BiasFree')
    sdn = seeded_state_dict(cfg, 4)
    sd = to_torch(sdn)
    clean, img = synthetic_noisy_image(h, w, 3, sigma=25.0, seed=h)
    with RestormerBackend(cfg, sdn, device=0) as be:
        got = be.restore_padcrop(img, ts, tp, ps)
    ref = padcrop_restore_image(img, lambda t: O.restormer_forward(t, sd, cfg.to_dict()), ts, tp, ps)
    diff = np.abs(got.astype(int) - ref.astype(int))
    d = abs(O.psnr_u8(got, clean) - O.psnr_u8(ref, clean))
    print(f"padcrop {h}x{w} ts={ts} pad={tp} ps={ps}: {100 * (diff > 0).mean():.2f}% px differ, max {diff.max()} LSB, |dPSNR| {d:.2e}")
    assert diff.max() <= 1 and (diff > 0).mean() < 0.05 and d <= 1e-3


@pytest.mark.gpu
def test_gpu_dncnn_padcrop_any_size():
    """DnCNN has no size constraint: odd tile / pad sizes."""
    from ira_amd.backend import DnCNNBackend
    from ira_amd.dncnn import DnCNNConfig, seeded_state_dict
    from oracle.dncnn_oracle import dncnn_forward
    cfg = DnCNNConfig(in_nc=1, nc=32, nb=4)
    sdn = seeded_state_dict(cfg, 1)
    sd = {k: torch.from_numpy(v) for k, v in sdn.items()}
    _, img = synthetic_noisy_image(61, 83, 1, sigma=25.0, seed=3)
    with DnCNNBackend(cfg, sdn, device=0) as be:
        got = be.restore_padcrop(img, 27, 5, 10)
        from ira_amd._capi import IraError
        with pytest.raises(IraError):
            be.restore_padcrop(img, 27, 5, 64 * 4)          # reflect pad >= image size
    ref = padcrop_restore_image(img, lambda t: dncnn_forward(t, sd, cfg.to_dict()), 27, 5, 10)
    diff = np.abs(got.astype(int) - ref.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize('path', [p for p in PADCROP if 'prepad' not in p], ids=[os.path.basename(p)[8:-4] for p in PADCROP if 'prepad' not in p])
def test_gpu_padcrop_matches_the_references_own_tiler(path):
    """The HIP harness (ira_restore_padcrop_u8) against what the reference's tiler methods produced (see above), uint8 level."""
    z = np.load(path)
    cfg, sdn, _ = _fixture_net(z)
    noisy = z['noisy']
    ts, tp = int(z['tile_size']), int(z['tile_pad'])
    ref = np.rint(np.clip(z['output'][0].transpose(1, 2, 0), 0, 1).astype(np.float32) * np.float32(255.0)).astype(np.uint8)
    if str(z['kind']) == 'xrestormer':
        from ira_amd.backend import XRestormerBackend
        with XRestormerBackend(cfg, sdn, device=0) as be:
            got = be.restore_padcrop(noisy, None if ts < 0 else ts, max(tp, 0), int(z['padding_size']))
    else:
        from ira_amd.backend import RRDBNetBackend
        mod = 2 if cfg.scale == 2 else (4 if cfg.scale == 1 else 1)
        with RRDBNetBackend(cfg, sdn, device=0) as be:
            got = be.restore_padcrop(noisy, ts if ts > 0 else None, tp, mod)
    assert got.shape == ref.shape
    diff = np.abs(got.astype(int) - ref.astype(int))
    anchor = np.repeat(np.repeat(noisy, cfg.scale, 0), cfg.scale, 1)
    d = abs(O.psnr_u8(got, anchor) - O.psnr_u8(ref, anchor))
    print(f"{os.path.basename(path)}: {100 * (diff > 0).mean():.2f}% px differ, max {diff.max()} LSB, |dPSNR| {d:.2e}")
    # the bar is max 1 LSB and |dPSNR| <= 1e-3 dB; the share of 1-LSB flips depends on the net (seeded X-Restormer: ~9 %, RRDBNet: < 1 %)
    assert diff.max() <= 1 and (diff > 0).mean() < 0.15 and d <= 1e-3
