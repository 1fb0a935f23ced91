"""MPRNet on a real MI355X, through the C ABI: the three-stage network against the reference module's goldens
(tests/golden/make_golden_mprnet.py), per-image channel attention across the batch, and the demo.py harness against the oracle."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from ira_amd.mprnet import MPRNetConfig, reduced_mprnet_config, seeded_state_dict
from ira_amd.synth import synthetic_noisy_image
from oracle import restormer_oracle as O
from oracle.mprnet_oracle import mprnet_forward

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(glob.glob(os.path.join(GOLD, 'mprnet_*.npz')))
_cache = {}


def backend(cfg, seed):
    from ira_amd.backend import MPRNetBackend
    key = (json.dumps(cfg.to_dict(), sort_keys=True), seed)
    if key not in _cache:
        if len(_cache) >= 2:
            _, old = _cache.popitem()
            old.close()
        _cache[key] = MPRNetBackend(cfg, seeded_state_dict(cfg, seed), device=0)
    return _cache[key]


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(p)[7:-4] for p in CASES])
def test_forward_matches_reference_golden(path):
    z = np.load(path)
    cfg = MPRNetConfig(**json.loads(str(z['cfg'])))
    be = backend(cfg, int(z['seed']))
    x, ref = z['input'], z['output']
    y = be.forward(x)
    assert y.shape == ref.shape and np.isfinite(y).all()
    # the network output is x + tail(...): judge the learned residual, not the skip
    res, res_ref = y - x, ref - x
    rel = np.abs(res - res_ref).max() / np.abs(res_ref).max()
    assert rel < 2e-3, rel                                   # fp32 streams, fp16 hi+lo MFMA operands
    assert O.psnr_f(y, ref) > 90.0


def test_batch_items_are_independent():
    """CALayer pools per image (MPRNet.py:34-37) and every patch split is per image: a batch equals its items run alone."""
    cfg = reduced_mprnet_config()
    be = backend(cfg, 3)
    g = np.random.default_rng(5)
    x = g.random((3, 3, 16, 24), dtype=np.float32)
    y = be.forward(x)
    for i in range(3):
        yi = be.forward(x[i:i + 1])
        assert np.array_equal(y[i:i + 1], yi)


def test_forward_rejects_sizes_the_reference_cannot_run():
    be = backend(reduced_mprnet_config(), 3)
    with pytest.raises(RuntimeError, match='multiples of 8'):
        be.forward(np.zeros((1, 3, 20, 24), np.float32))


@pytest.mark.parametrize('h,w', [(45, 52), (64, 40)])
def test_harness_matches_oracle(h, w):
    """demo.py:64-87: reflect-pad to a multiple of 8 only when needed, model(x)[0], clamp, crop, img_as_ubyte."""
    cfg = reduced_mprnet_config()
    be = backend(cfg, 3)
    sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(cfg, 3).items()}
    _, img = synthetic_noisy_image(h, w, 3, sigma=15.0, seed=h)
    ref = O.restore_image(img, None, None, tile=None, forward_fn=lambda t: mprnet_forward(t, sd, cfg.to_dict()))
    out = be.restore(img, tile=None)
    assert out.shape == ref.shape and out.dtype == np.uint8
    d = np.abs(out.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
