"""RRDBNet / Real-ESRGAN on a real MI355X, through the C ABI: dense blocks computed in place in channel slices against the
reference module's goldens (tests/golden/make_golden_rrdbnet.py), and the RealESRGANer tensor path (mod-pad + tile_process) against the oracle."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from ira_amd.rrdbnet import RRDBNetConfig, reduced_rrdbnet_config, seeded_state_dict
from ira_amd.synth import synthetic_noisy_image
from oracle import restormer_oracle as O
from oracle.padcrop_oracle import padcrop_restore_image
from oracle.rrdbnet_oracle import rrdbnet_forward

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(glob.glob(os.path.join(GOLD, 'rrdbnet_*.npz')))


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_forward_matches_reference_golden(path):
    from ira_amd.backend import RRDBNetBackend
    z = np.load(path)
    cfg = RRDBNetConfig(**json.loads(str(z['cfg'])))
    be = RRDBNetBackend(cfg, seeded_state_dict(cfg, int(z['seed'])), device=0)
    y, ref = be.forward(z['input']), z['output']
    be.close()
    assert y.shape == ref.shape and np.isfinite(y).all()
    rel = np.abs(y - ref).max() / np.abs(ref).max()
    assert rel < 2e-3, rel                                   # fp32 streams, fp16 hi+lo MFMA operands
    assert O.psnr_f(y, ref, peak=float(np.abs(ref).max())) > 80.0


def _image_like(cfg, seed):
    """Seeded weights whose output is image-like (0.5 +- 0.1) so that clamp and uint8 rounding are exercised away from saturation."""
    sd = seeded_state_dict(cfg, seed)
    sd['conv_last.weight'] = (sd['conv_last.weight'] * 0.1).astype(np.float32)
    sd['conv_last.bias'] = np.full_like(sd['conv_last.bias'], 0.5)
    return sd


@pytest.mark.parametrize('scale,h,w,tile,pad', [(4, 37, 50, None, 10), (4, 61, 45, 32, 8), (2, 45, 63, 40, 10), (2, 50, 64, None, 10), (1, 30, 41, 16, 4)])
def test_realesrganer_tensor_path(scale, h, w, tile, pad):
    """realesrganer.py:71-183: reflect mod-pad to 2 (scale 2) / 4 (scale 1), tile_process with tile_pad, crop; uint8 out."""
    from ira_amd.backend import RRDBNetBackend
    cfg = reduced_rrdbnet_config(scale=scale)
    sd = _image_like(cfg, 11)
    be = RRDBNetBackend(cfg, sd, device=0)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    _, img = synthetic_noisy_image(h, w, 3, sigma=10.0, seed=h)
    mod = {4: 1, 2: 2, 1: 4}[scale]
    got = be.restore_padcrop(img, tile_size=tile, tile_pad=pad, padding_size=mod)
    be.close()
    ref = padcrop_restore_image(img, lambda t: rrdbnet_forward(t, tsd, cfg.to_dict()), tile, pad, mod, scale=scale)
    assert got.shape == (h * scale, w * scale, 3) and got.shape == ref.shape
    d = np.abs(got.astype(int) - ref.astype(int))
    assert 20 < ref.mean() < 235 and ref.std() > 3
    assert d.max() <= 1 and (d > 0).mean() < 5e-3, (d.max(), (d > 0).mean())


def test_forward_rejects_odd_sizes_for_scale_2():
    from ira_amd.backend import RRDBNetBackend
    cfg = reduced_rrdbnet_config(scale=2)
    be = RRDBNetBackend(cfg, seeded_state_dict(cfg, 1), device=0)
    with pytest.raises(RuntimeError, match='multiples of 2'):
        be.forward(np.zeros((1, 3, 21, 20), np.float32))
    be.close()
