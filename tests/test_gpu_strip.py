"""Whole-frame Restormer split into horizontal bands on a real MI355X: the band schedule of the C ABI (halo rows,
interior-only Gram, one record per rank) against the single-GPU whole-frame path and the oracle.  The bands of one frame run on ONE GPU
here, one host thread + model handle each, exchanging through ThreadTransport (the box has one GPU; the RCCL transport is the same call
sites, covered by the gloo test in tests/test_dist_gloo.py)."""
import threading

import numpy as np
import pytest
import torch

from ira_amd.config import reduced_config
from ira_amd.dist import StripParallelRestorer
from ira_amd.test_transports import ThreadTransport
from ira_amd.weights import seeded_state_dict, to_torch
from oracle import restormer_oracle as O

pytestmark = pytest.mark.gpu


def _run_bands(cfg, seed, noisy, world, return_float=False):
    from ira_amd.backend import RestormerBackend
    dev = torch.device('cuda', 0)
    frame = torch.from_numpy(noisy).to(dev)
    shared = ThreadTransport.make_shared(world)
    bes = [RestormerBackend(cfg, seeded_state_dict(cfg, seed), device=0) for _ in range(world)]
    outs, errs = [None] * world, [None] * world

    def work(r):
        try:
            torch.cuda.set_device(0)
            rst = StripParallelRestorer(bes[r], r, world, transport=ThreadTransport(shared, r, world))
            outs[r] = rst.restore_frame(frame, return_float=return_float).cpu().numpy()
        except BaseException as e:
            errs[r] = e
            shared['barrier'].abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    for b in bes:
        b.close()
    for e in errs:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    assert all(o is not None for o in outs), errs
    return outs


def _whole_float(cfg, seed, noisy):
    """The single-GPU whole-frame forward on the SAME float input the band path builds (demo.py:120-131)."""
    import torch.nn.functional as F
    from ira_amd.backend import RestormerBackend
    h, w, _ = noisy.shape
    x = torch.from_numpy(noisy).cuda().permute(2, 0, 1).unsqueeze(0).float().div(255.)
    H, W = ((h + 8) // 8) * 8, ((w + 8) // 8) * 8
    x = F.pad(x, (0, W - w if w % 8 else 0, 0, H - h if h % 8 else 0), 'reflect') if (h % 8 or w % 8) else x
    be = RestormerBackend(cfg, seeded_state_dict(cfg, seed), device=0)
    y = be.forward(x.cpu().numpy())
    be.close()
    return np.transpose(y[0], (1, 2, 0))[:h, :w]


@pytest.mark.parametrize('ln,world,h,w', [('BiasFree', 1, 40, 56), ('BiasFree', 2, 72, 64), ('BiasFree', 3, 72, 64), ('WithBias', 2, 45, 52),
                                          ('BiasFree', 4, 64, 40), ('WithBias', 3, 96, 48)])
def test_bands_match_whole_frame(ln, world, h, w):
    """Float outputs: the only arithmetic difference to the one-GPU forward is the grouping of the Gram partial sums (one record per
    band), so the two agree to fp16-branch rounding noise everywhere, the seams between bands included."""
    cfg = reduced_config(ln)
    clean, noisy = O.synthetic_noisy_image(h, w, 3, seed=30 + world)
    whole = _whole_float(cfg, 9, noisy)
    outs = _run_bands(cfg, 9, noisy, world, return_float=True)
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])                   # every rank assembles the same frame
    d = np.abs(outs[0] - whole)
    assert np.isfinite(outs[0]).all() and O.psnr_f(outs[0], whole) > 75.0, (d.max(), O.psnr_f(outs[0], whole))
    rows = d.reshape(h, -1).max(1)
    assert rows.max() < 20 * np.median(rows) + 1e-4, (rows.argmax(), rows.max(), np.median(rows))     # no seam stands out


@pytest.mark.parametrize('world,h,w', [(2, 72, 64), (3, 45, 52)])
def test_bands_harness_matches_oracle(world, h, w):
    cfg = reduced_config('BiasFree')
    clean, noisy = O.synthetic_noisy_image(h, w, 3, seed=40 + world)
    outs = _run_bands(cfg, 9, noisy, world)
    ref = O.restore_image(noisy, to_torch(seeded_state_dict(cfg, 9)), cfg.to_dict(), tile=None)
    assert outs[0].dtype == np.uint8 and outs[0].shape == ref.shape
    dr = np.abs(outs[0].astype(int) - ref.astype(int))
    assert dr.max() <= 2 and abs(O.psnr_u8(outs[0], clean) - O.psnr_u8(ref, clean)) <= 1e-3


def test_strip_forward_argument_checks():
    from ira_amd.backend import RestormerBackend
    from ira_amd._capi import IraError
    cfg = reduced_config('BiasFree')
    be = RestormerBackend(cfg, seeded_state_dict(cfg, 9), device=0)
    x = torch.zeros((24, 16, 3), device='cuda')
    with pytest.raises(IraError, match='ira_strip_configure first'):
        be.strip_forward_device(x.data_ptr(), 24, 16, 0, 0, x.data_ptr())
    from ira_amd.test_transports import LocalTransport
    rst = StripParallelRestorer(be, 0, 1, transport=LocalTransport())
    rst._configure(16, x.device)
    with pytest.raises(IraError, match='upper neighbour'):
        be.strip_forward_device(x.data_ptr(), 24, 16, 8, 0, x.data_ptr())      # rank 0 of 1 has no neighbours
    be.close()
