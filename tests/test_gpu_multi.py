"""Two ranks on TWO GPUs over RCCL (the communicator lives in libira_hip.so: ira_comm_init, ira_restore_frame_distributed,
ira_strip_restore_u8_device).  Skipped where fewer than two devices are visible - the builder's box and the round-end test box have one;
what runs there instead: the same entry points with a one-rank communicator (tests/test_gpu_api.py::test_rccl_paths_with_one_rank), the
band schedule with several handles on one GPU (tests/test_gpu_strip.py) and the host logic at world 2-3 over gloo (tests/test_dist_gloo.py).
A rank that fails leaves through the parent's timeout: the workers are joined with a deadline and killed by exact PID."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, noisy, tile, overlap, out_q):
    try:
        os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
        torch.cuda.set_device(rank)
        from ira_amd.backend import RestormerBackend
        from ira_amd.config import reduced_config
        from ira_amd.dist import StripParallelRestorer, TileParallelRestorer, init_comm
        from ira_amd.weights import seeded_state_dict
        cfg = reduced_config('BiasFree')
        be = RestormerBackend(cfg, seeded_state_dict(cfg, 9), device=rank)
        init_comm(be, rank, world, '127.0.0.1', port)
        frame = torch.from_numpy(noisy).cuda(rank)
        tiled = TileParallelRestorer(be, rank, world).restore_frame(frame, tile, overlap).cpu().numpy()
        bands = StripParallelRestorer(be, rank, world).restore_frame(frame).cpu().numpy()
        be.close()
        out_q.put((rank, tiled, bands, None))
    except BaseException as e:      # report instead of leaving the parent to wait for the timeout
        out_q.put((rank, None, None, repr(e)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_ranks_over_rccl_match_one_gpu():
    import torch.multiprocessing as mp
    from ira_amd.backend import RestormerBackend
    from ira_amd.config import reduced_config
    from ira_amd.weights import seeded_state_dict
    from oracle import restormer_oracle as O
    world, tile, overlap = 2, 48, 16
    _, noisy = O.synthetic_noisy_image(104, 120, 3, seed=77)
    cfg = reduced_config('BiasFree')
    with RestormerBackend(cfg, seeded_state_dict(cfg, 9), device=0) as be:
        ref_tiled = be.restore(noisy, tile=tile, tile_overlap=overlap)
        ref_whole = be.restore(noisy, tile=None)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, noisy, tile, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            rank, tiled, bands, err = q.get(timeout=240)
            assert err is None, f"rank {rank}: {err}"
            got[rank] = (tiled, bands)
    finally:
        for p in procs:
            p.join(10)
            if p.is_alive():
                p.kill()
    for rank in range(world):
        tiled, bands = got[rank]
        assert np.array_equal(tiled, ref_tiled), f"rank {rank}: tile-parallel frame differs from the one-GPU tiler"     # same tiles, same blend order
        d = np.abs(bands.astype(int) - ref_whole.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.02, (rank, int(d.max()), float((d > 0).mean()))                    # Gram partial sums grouped per band
