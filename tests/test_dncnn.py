"""DnCNN (BASELINE config 1).  The reference has no DnCNN code, so parity versus the reference is
unpinned; what can be checked: the oracle against torch.nn modules of the documented layout (CPU), the packer's BatchNorm
fold (CPU), and the HIP path against the oracle on the config-1 shape, one 256x256 grayscale tile with sigma = 25 noise (GPU)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from ira_amd.dncnn import DnCNNConfig, seeded_state_dict, state_dict_spec, pack_dncnn_arrays
from ira_amd.synth import synthetic_noisy_image
from oracle.dncnn_oracle import dncnn_forward
from oracle import restormer_oracle as O


def nn_model(cfg):
    layers = [nn.Conv2d(cfg.in_nc, cfg.nc, 3, padding=1), nn.ReLU()]
    for _ in range(cfg.nb - 2):
        layers += [nn.Conv2d(cfg.nc, cfg.nc, 3, padding=1), nn.BatchNorm2d(cfg.nc), nn.ReLU()]
    layers += [nn.Conv2d(cfg.nc, cfg.in_nc, 3, padding=1)]

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = nn.Sequential(*layers)

        def forward(self, x):
            return x - self.model(x)
    return M().eval()


@pytest.mark.parametrize('cfg', [DnCNNConfig(), DnCNNConfig(in_nc=3, nc=32, nb=5)])
def test_oracle_matches_nn_modules(cfg):
    m = nn_model(cfg)
    want = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith('num_batches_tracked')}
    assert dict(state_dict_spec(cfg)) == want
    sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(cfg, 3).items()}
    m.load_state_dict(sd, strict=False)
    x = torch.rand(2, cfg.in_nc, 24, 20)
    with torch.no_grad():
        assert (m(x) - dncnn_forward(x, sd, cfg.to_dict())).abs().max() < 1e-5


def test_packer_folds_batchnorm():
    cfg = DnCNNConfig(nc=32, nb=4)
    sd = seeded_state_dict(cfg, 5)
    a = pack_dncnn_arrays(sd, cfg)
    assert a['mid.0_w'].shape == (2, 9, 1, 2, 64, 8) and a['last_w'].shape == (2, 9, 1, 2, 64, 8)
    scale = sd['model.3.weight'] / np.sqrt(sd['model.3.running_var'] + 1e-5)
    want_b = (sd['model.2.bias'] - sd['model.3.running_mean']) * scale + sd['model.3.bias']
    np.testing.assert_allclose(a['mid.0_b'][:32], want_b, rtol=1e-6)
    np.testing.assert_allclose(a['last_b'][:1], -sd['model.8.bias'], rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('cfg,B,H,W', [(DnCNNConfig(), 1, 256, 256), (DnCNNConfig(in_nc=3, nc=32, nb=5), 2, 37, 50)])
def test_gpu_matches_oracle(cfg, B, H, W):
    from ira_amd.backend import DnCNNBackend
    sdn = seeded_state_dict(cfg, 2)
    sd = {k: torch.from_numpy(v) for k, v in sdn.items()}
    clean, noisy = synthetic_noisy_image(H, W, cfg.in_nc, sigma=25.0, seed=9)
    x = torch.from_numpy(noisy).float().div(255.).permute(2, 0, 1).unsqueeze(0).repeat(B, 1, 1, 1).contiguous()
    if B > 1:
        x[1] = x[1].flip(-1)
    with DnCNNBackend(cfg, sdn, device=0) as be:
        y = be.forward(x.numpy())
    with torch.no_grad():
        ref = dncnn_forward(x, sd, cfg.to_dict()).numpy()
    p = O.psnr_f(y, ref)
    u8 = lambda a: O.img_as_ubyte(np.clip(a[0].transpose(1, 2, 0), 0, 1))
    d = abs(O.psnr_u8(u8(y), clean) - O.psnr_u8(u8(ref), clean))
    print(f"DnCNN {cfg.in_nc}ch nc{cfg.nc} nb{cfg.nb} {H}x{W}: PSNR(build, oracle) {p:.2f} dB, max abs err {np.abs(y - ref).max():.2e}, |dPSNR_u8| {d:.2e}")
    assert p >= 58.0 and d <= 1e-3
