"""Multi-rank tile-parallel path (ira_amd/dist.py) on CPU with the gloo backend, world_size 2 and 3:
partition + all-gather + row-major blend must reproduce the single-process reference tiler
(demo.py:132-153) bit for bit.  The per-tile 'network' is a cheap deterministic stand-in; the real
kernels are covered by the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from ira_amd.dist import TileParallelRestorer, block_partition
from oracle import restormer_oracle as O


def fake_net(t):                      # (n, th, tw, c) float32 -> same shape; depends on the whole tile (like attention)
    return t * 0.75 + 0.1 * t.mean(dim=(1, 2, 3), keepdim=True) + 0.05


def py_plan(h, w, tile, overlap):
    padh, padw = O.pad_amounts(h, w)
    Hp, Wp = h + padh, w + padw
    if not tile:
        return Hp, Wp, Hp, Wp, 1, 1
    t = min(tile, Hp, Wp)
    s = t - overlap
    return Hp, Wp, t, t, len(O.tile_index_list(Hp, t, s)), len(O.tile_index_list(Wp, t, s))


class CpuTileBackend:
    """restore_tiles_t / blend_tiles_t with the arithmetic of csrc k_extract_tiles / k_blend in torch."""

    def _geom(self, h, w, tile, overlap):
        Hp, Wp, th, tw, nr, nc = py_plan(h, w, tile, overlap)
        hi = O.tile_index_list(Hp, th, th - overlap) if tile else [0]
        wi = O.tile_index_list(Wp, tw, tw - overlap) if tile else [0]
        return Hp, Wp, th, tw, hi, wi

    def restore_tiles_t(self, frame, h, w, c, tile, overlap, first, n, out):
        Hp, Wp, th, tw, hi, wi = self._geom(h, w, tile, overlap)
        x = frame.float().div(255.).permute(2, 0, 1).unsqueeze(0)
        x = torch.nn.functional.pad(x, (0, Wp - w, 0, Hp - h), 'reflect')[0].permute(1, 2, 0)
        origins = [(y, xx) for y in hi for xx in wi][first:first + n]
        tiles = torch.stack([x[y:y + th, xx:xx + tw] for y, xx in origins])
        out[:n] = fake_net(tiles)

    def blend_tiles_t(self, tiles, h, w, c, tile, overlap, out):
        Hp, Wp, th, tw, hi, wi = self._geom(h, w, tile, overlap)
        E = torch.zeros(Hp, Wp, c); Wt = torch.zeros(Hp, Wp, c)
        k = 0
        for y in hi:
            for xx in wi:
                E[y:y + th, xx:xx + tw] += tiles[k]; Wt[y:y + th, xx:xx + tw] += 1; k += 1
        out.copy_(torch.from_numpy(O.img_as_ubyte(torch.clamp(E / Wt, 0, 1)[:h, :w].numpy())))


def sw_plan(h, w, tile, overlap, ws=8, scale=2):
    """SwinIR harness plan (SwinIR/inference.py:76-80,294-310) with the 7th value = scale, as ira_model_tile_plan returns it."""
    Hp, Wp = (h // ws + 1) * ws, (w // ws + 1) * ws
    if not tile:
        return Hp, Wp, Hp, Wp, 1, 1, scale
    t = min(tile, Hp, Wp)
    s = t - overlap
    return Hp, Wp, t, t, len(O.tile_index_list(Hp, t, s)), len(O.tile_index_list(Wp, t, s)), scale


def fake_sr(t):                       # (n, th, tw, c) -> (n, 2th, 2tw, c), tile-dependent like a real network
    up = t.repeat_interleave(2, 1).repeat_interleave(2, 2)
    return up * 0.8 + 0.1 * t.mean(dim=(1, 2, 3), keepdim=True)


class CpuSrBackend:
    """SwinIR-harness stand-in: always-pad by flip-concat, x2 outputs, overlap average on the scaled grid."""

    def _geom(self, h, w, tile, overlap):
        Hp, Wp, th, tw, nr, nc, s = sw_plan(h, w, tile, overlap)
        hi = O.tile_index_list(Hp, th, th - overlap) if tile else [0]
        wi = O.tile_index_list(Wp, tw, tw - overlap) if tile else [0]
        return Hp, Wp, th, tw, hi, wi, s

    def restore_tiles_t(self, frame, h, w, c, tile, overlap, first, n, out):
        from oracle.swinir_oracle import harness_pad
        Hp, Wp, th, tw, hi, wi, s = self._geom(h, w, tile, overlap)
        x = harness_pad(frame.float().div(255.).permute(2, 0, 1).unsqueeze(0), 8)[0].permute(1, 2, 0)
        origins = [(y, xx) for y in hi for xx in wi][first:first + n]
        out[:n] = fake_sr(torch.stack([x[y:y + th, xx:xx + tw] for y, xx in origins]))

    def blend_tiles_t(self, tiles, h, w, c, tile, overlap, out):
        Hp, Wp, th, tw, hi, wi, s = self._geom(h, w, tile, overlap)
        E = torch.zeros(Hp * s, Wp * s, c); Wt = torch.zeros(Hp * s, Wp * s, c)
        k = 0
        for y in hi:
            for xx in wi:
                E[y * s:(y + th) * s, xx * s:(xx + tw) * s] += tiles[k]; Wt[y * s:(y + th) * s, xx * s:(xx + tw) * s] += 1; k += 1
        out.copy_(torch.from_numpy(O.img_as_ubyte(torch.clamp(E / Wt, 0, 1)[:h * s, :w * s].numpy())))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _make(kind, rank, world):
    from ira_amd.test_transports import GlooTileGather
    gather = GlooTileGather() if world > 1 else None          # the product path all-gathers inside libira_hip.so (RCCL); this is the CPU stand-in
    if kind == 'sr':
        return TileParallelRestorer(CpuSrBackend(), rank, world, plan_fn=sw_plan, gather=gather)
    return TileParallelRestorer(CpuTileBackend(), rank, world, plan_fn=py_plan, gather=gather)


def _worker(rank, world, port, h, w, tile, overlap, q, kind='rst'):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _, noisy = O.synthetic_noisy_image(h, w, 3, seed=5)
    rst = _make(kind, rank, world)
    out = rst.restore_frame(torch.from_numpy(noisy), tile, overlap)
    out2 = rst.restore_frame(torch.from_numpy(noisy), tile, overlap)     # buffers are reused across frames
    assert torch.equal(out, out2)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,h,w,tile,overlap,kind', [(2, 150, 203, 64, 16, 'rst'), (3, 97, 130, 48, 8, 'rst'), (2, 70, 100, 512, 32, 'rst'),
                                                          (2, 61, 90, 32, 8, 'sr'), (3, 64, 64, 24, 8, 'sr')])
def test_tile_parallel_matches_single_process(world, h, w, tile, overlap, kind):
    """'rst': Restormer harness (pad to 8 if needed, scale 1); 'sr': SwinIR harness (always-pad flip-concat, x2 outputs)."""
    _, noisy = O.synthetic_noisy_image(h, w, 3, seed=5)
    single = _make(kind, 0, 1).restore_frame(torch.from_numpy(noisy), tile, overlap).numpy()
    assert single.shape == ((2 * h, 2 * w, 3) if kind == 'sr' else (h, w, 3))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, h, w, tile, overlap, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert np.array_equal(results[r], single), r


def test_world_gt_1_needs_a_communicator():
    from ira_amd._capi import IraError
    with pytest.raises(IraError, match='communicator'):
        TileParallelRestorer(CpuTileBackend(), 0, 2, plan_fn=py_plan)
    from ira_amd.dist import StripParallelRestorer
    with pytest.raises(IraError, match='communicator'):
        StripParallelRestorer(CpuStripBackend(), 0, 2)


def _uid_worker(rank, world, port, q):
    from ira_amd.dist import exchange_unique_id
    q.put((rank, exchange_unique_id(rank, world, lambda: bytes(range(128)), '127.0.0.1', port)))


def test_unique_id_travels_over_tcp():
    """The only thing the host does for the RCCL communicator: carry rank 0's 128-byte id to the other ranks (ira_amd.dist)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uid_worker, args=(r, 3, port, q)) for r in (2, 1, 0)]      # rank 0 last: the others retry until it listens
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(3))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert all(got[r] == bytes(range(128)) for r in range(3))


def test_block_partition_covers_in_order():
    for n in (1, 5, 6, 40, 41):
        for world in (1, 2, 3, 4, 8):
            got = []
            chunk = None
            for r in range(world):
                first, count, chunk = block_partition(n, world, r)
                assert count <= chunk
                got += list(range(first, first + count))
            assert got == list(range(n))
    assert block_partition(40, 8, 3) == (15, 5, 5)      # 5 tiles per GPU at 4K on 8 GPUs


# ---------------------------------------------------------------------------------------------------------------------
# Strip-parallel whole frame: the host side (band bounds, halo rows, the two exchange callbacks over torch.distributed,
# assembly) with a CPU stand-in for libira_hip's band forward that has the same two needs as Restormer: a 3x3 footprint (one halo row
# refreshed through comm.exchange) and a whole-frame statistic (one record per rank through comm.allgather).

class CpuStripBackend:
    def strip_buffer_bytes(self, Wp):
        return Wp * 3 * 4, 8

    def strip_configure(self, comm):
        self.comm = comm

    @staticmethod
    def _arr(ptr, shape, dtype=np.float32):
        import ctypes as C
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return np.frombuffer((C.c_char * n).from_address(ptr), dtype=dtype).reshape(shape)

    def strip_forward_device(self, in_ptr, H, W, top, bot, out_ptr):
        cm = self.comm
        x = self._arr(in_ptr, (H, W, 3)).copy()
        row = W * 3 * 4
        if top:
            x[:top] = np.nan                                                   # halo rows arrive through the exchange only
            self._arr(cm.send_top, (W, 3))[:] = x[top]
        if bot:
            x[H - bot:] = np.nan
            self._arr(cm.send_bot, (W, 3))[:] = x[H - bot - 1]
        if top or bot:
            assert cm.exchange(cm.ctx, row) == 0
        if top:
            x[top - 1] = self._arr(cm.recv_top, (W, 3))
        if bot:
            x[H - bot] = self._arr(cm.recv_bot, (W, 3))
        inner = x[top:H - bot]
        self._arr(cm.gather_send, (1,), np.int64)[0] = int((inner > 0.5).sum())
        assert cm.allgather(cm.ctx, 8) == 0
        total = int(self._arr(cm.gather_recv, (cm.world,), np.int64).sum())
        xz = np.pad(np.nan_to_num(x, nan=0.0), ((1, 1), (1, 1), (0, 0)))
        box = sum(xz[dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)) / 9.0
        y = np.where(np.isnan(x), 0.0, box) * 0.75 + 0.25 * ((total % 97) / 97.0)
        self._arr(out_ptr, (H, W, 3))[:] = y.astype(np.float32)


def _strip_worker(rank, world, port, h, w, q):
    import torch.distributed as dist
    from ira_amd.dist import StripParallelRestorer
    from ira_amd.test_transports import TorchDistTransport
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _, noisy = O.synthetic_noisy_image(h, w, 3, seed=8)
    rst = StripParallelRestorer(CpuStripBackend(), rank, world, transport=TorchDistTransport(rank, world))
    out = rst.restore_frame(torch.from_numpy(noisy))
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,h,w', [(2, 45, 52), (3, 72, 40), (2, 16, 24)])
def test_strip_parallel_matches_single_process(world, h, w):
    from ira_amd.dist import StripParallelRestorer
    _, noisy = O.synthetic_noisy_image(h, w, 3, seed=8)
    single = StripParallelRestorer(CpuStripBackend(), 0, 1).restore_frame(torch.from_numpy(noisy)).numpy()
    assert single.shape == (h, w, 3)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strip_worker, args=(r, world, port, h, w, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert np.array_equal(results[r], single), r


def test_strip_bounds():
    from ira_amd.dist import strip_bounds
    assert strip_bounds(2160, 8)[0] == (0, 272) and strip_bounds(2160, 8)[-1] == (1896, 2160)     # 270 groups of 8 rows: 6 x 34 + 2 x 33
    for Hp, world in ((2160, 1), (2160, 3), (64, 8), (72, 4)):
        b = strip_bounds(Hp, world)
        assert b[0][0] == 0 and b[-1][1] == Hp and all(x[1] == y[0] for x, y in zip(b, b[1:])) and all((e - s) % 8 == 0 and e > s for s, e in b)
    with pytest.raises(ValueError):
        strip_bounds(24, 4)
