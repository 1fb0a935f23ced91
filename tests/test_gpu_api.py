"""Agent-facing API on the real backend (MI355X): Tool contract, RestormerService, restore(), and the
tile-parallel restorer at world_size 1, all against the CPU oracle."""
import io

import numpy as np
import pytest
import torch
from PIL import Image

from ira_amd import service as S
from ira_amd import tools as T
from ira_amd.config import task_config
from ira_amd.weights import seeded_state_dict, to_torch
from oracle import restormer_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cache():
    c = T.BackendCache(device=0, state_dict_loader=T.seeded_state_dict_loader(7))
    yield c
    c.close()


def _oracle(img, task, tile=None, overlap=32):
    cfg = task_config(task)
    torch.set_num_threads(16)
    return O.restore_image(img, to_torch(seeded_state_dict(cfg, 7)), cfg.to_dict(), tile=tile, tile_overlap=overlap)


def _check(out, ref, clean):
    diff = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    assert diff.max() <= 2
    assert abs(O.psnr_u8(out, clean) - O.psnr_u8(ref, clean)) <= 1e-3


def test_tool_end_to_end(tmp_path, cache):
    clean, noisy = O.synthetic_noisy_image(72, 88, 3, seed=11)
    inp, out = tmp_path / 'in', tmp_path / 'out'
    inp.mkdir(); out.mkdir()
    Image.fromarray(noisy).save(inp / 'input.png')
    T.Restormer('motion_deblurring', cache=cache)(inp, out, silent=True)          # WithBias LN path
    got = np.asarray(Image.open(out / 'output.png'))
    _check(got, _oracle(noisy, 'Motion_Deblurring'), clean)


def test_service_and_restore_alias(cache):
    clean, noisy = O.synthetic_noisy_image(64, 80, 3, seed=12)
    svc = S.RestormerService(task='Real_Denoising', cache=cache)
    buf = io.BytesIO(); Image.fromarray(noisy).save(buf, format='PNG')
    png = svc.restore_image(buf.getvalue(), 'Real_Denoising', tile=48, tile_overlap=16)
    got = np.asarray(Image.open(io.BytesIO(png)))
    ref = _oracle(noisy, 'Real_Denoising', 48, 16)
    _check(got, ref, clean)
    assert np.array_equal(S.restore(noisy, task='denoising', tile=48, tile_overlap=16, cache=cache), got)
    # gray task
    cg, ng = O.synthetic_noisy_image(56, 56, 1, seed=13)
    gg = S.restore(ng, task='Gaussian_Gray_Denoising', cache=cache)
    _check(gg, _oracle(ng, 'Gaussian_Gray_Denoising'), cg)


def test_tile_parallel_restorer_world1(cache):
    from ira_amd.dist import TileParallelRestorer
    clean, noisy = O.synthetic_noisy_image(120, 136, 3, seed=14)
    be = cache.get('Real_Denoising')
    rst = TileParallelRestorer(be, 0, 1)
    out = rst.restore_frame(torch.from_numpy(noisy).cuda(), 64, 32).cpu().numpy()
    assert np.array_equal(out, be.restore(noisy, tile=64, tile_overlap=32))


def test_rccl_paths_with_one_rank():
    """The collective entry points of the C ABI on a real communicator (csrc/ira_comm.hip: librccl.so dlopen-ed, ncclCommInitRank with one
    rank, collectives enqueued on the model's stream): the tile-parallel frame and the band-parallel whole frame equal the single-GPU
    entry points bit for bit.  (More than one rank needs more than one GPU: the driver's scaling runs.)"""
    from ira_amd.backend import RestormerBackend
    from ira_amd.config import reduced_config
    from ira_amd.dist import StripParallelRestorer, TileParallelRestorer, init_comm
    from ira_amd.weights import seeded_state_dict
    cfg = reduced_config('BiasFree')
    clean, noisy = O.synthetic_noisy_image(100, 131, 3, seed=21)
    with RestormerBackend(cfg, seeded_state_dict(cfg, 1), device=0) as be:
        ref_tiled = be.restore(noisy, tile=48, tile_overlap=16)
        ref_whole = be.restore(noisy, tile=None)
        init_comm(be, 0, 1)
        assert be.has_comm
        frame = torch.from_numpy(noisy).cuda()
        out = TileParallelRestorer(be, 0, 1).restore_frame(frame, 48, 16).cpu().numpy()
        assert np.array_equal(out, ref_tiled)
        band = StripParallelRestorer(be, 0, 1).restore_frame(frame).cpu().numpy()
        assert band.shape == ref_whole.shape and band.dtype == np.uint8
        # one band = the whole frame; the band schedule sums its Gram records in another order than the one-launch path
        d = np.abs(band.astype(int) - ref_whole.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.02, (d.max(), (d > 0).mean())
        be.comm_destroy()
        assert not be.has_comm


def test_fused_and_unfused_ffn_agree(cache):
    be = cache.get('Real_Denoising')
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 64, 96, generator=g).numpy()
    y1 = be.forward(x)
    be.set_fused_ffn(False)
    try:
        y0 = be.forward(x)
    finally:
        be.set_fused_ffn(True)
    cfg = task_config('Real_Denoising')
    with torch.no_grad():
        yo = O.restormer_forward(torch.from_numpy(x), to_torch(seeded_state_dict(cfg, 7)), cfg.to_dict()).numpy()
    p1, p0, p10 = O.psnr_f(y1, yo), O.psnr_f(y0, yo), O.psnr_f(y1, y0)
    print(f"fused vs oracle {p1:.2f} dB, per-op vs oracle {p0:.2f} dB, fused vs per-op {p10:.2f} dB")
    # same numerics contract, different kernels and rounding points: each path is as close to the fp32 oracle as the other
    # (70.2 dB each when written), and they differ from each other by about as much as either differs from the oracle (70.18 dB with the second-generation fused kernel)
    assert p1 >= 66.0 and p0 >= 66.0 and abs(p1 - p0) <= 1.0 and p10 >= min(p1, p0) - 1.0
