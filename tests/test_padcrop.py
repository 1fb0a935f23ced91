"""Pad-crop tiler (xrestormer_model.py:53-136): oracle properties on CPU, HIP path against the oracle
on the GPU (reduced Restormer and the build's DnCNN as the networks)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ira_amd.synth import synthetic_noisy_image
from oracle.padcrop_oracle import padcrop_restore, padcrop_restore_image
from oracle import restormer_oracle as O


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
