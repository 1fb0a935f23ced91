"""Tile-parallel restoration across the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference runs the tiler as a Python double loop of batch-1 forwards on one GPU
(demo.py:143-153; agentic_api.py:2 pins one device).  Tiles are independent forwards (per-tile
attention statistics, per-tile zero padding), so the row-major tile list is block-partitioned over
ranks; each rank runs its tiles as ONE batched forward, then a single all-gather of the restored
tiles (float32 NHWC; 40 x 3 x 512 x 512 x 4 B = 126 MB for a 4K frame) gives every rank the whole
list, and the E/W overlap-average runs on each rank in the reference's row-major order (bit-stable).

torch.distributed is plumbing only: backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import torch

from . import _capi


def block_partition(n_tiles: int, world: int, rank: int):
    """Contiguous blocks of ceil(n/world) tiles; trailing ranks may be short or empty.
    With equal chunks the all-gathered buffer holds the tiles in row-major order already."""
    chunk = (n_tiles + world - 1) // world
    first = min(rank * chunk, n_tiles)
    count = max(0, min(chunk, n_tiles - first))
    return first, count, chunk


class TileParallelRestorer:
    """``backend`` needs restore_tiles_t(frame, h, w, c, tile, overlap, first, n, out) and
    blend_tiles_t(tiles, h, w, c, tile, overlap, out) on torch tensors (RestormerBackend provides them
    on device pointers; the CPU tests inject a fake)."""

    def __init__(self, backend, rank: int = 0, world: int = 1, group=None, plan_fn=None):
        self.be = backend
        self.rank, self.world, self.group = rank, world, group
        # the plan of the model's own harness (SwinIR: always-pad rule, outputs scaled by `upscale`); plan functions may
        # return 6 values (scale 1) or 7
        self.plan_fn = plan_fn or getattr(backend, 'plan', None) or _capi.tile_plan
        self._bufs = {}

    def _buffers(self, key, n_local, n_all, th, tw, c, device):
        if key not in self._bufs:
            local = torch.empty((max(n_local, 1), th, tw, c), dtype=torch.float32, device=device)
            allt = local if self.world == 1 else torch.empty((n_all, th, tw, c), dtype=torch.float32, device=device)
            self._bufs = {key: (local, allt)}
        return self._bufs[key]

    def restore_frame(self, frame_u8: torch.Tensor, tile, overlap: int = 32, out: torch.Tensor | None = None):
        """frame_u8: (h,w,c) uint8 tensor resident on this rank's device (every rank holds the frame).
        Returns the restored (h*scale, w*scale, c) uint8 tensor (identical on every rank)."""
        h, w, c = frame_u8.shape
        plan = tuple(self.plan_fn(h, w, tile, overlap))
        Hp, Wp, th, tw, n_rows, n_cols = plan[:6]
        s = plan[6] if len(plan) > 6 else 1
        nt = n_rows * n_cols
        first, count, chunk = block_partition(nt, self.world, self.rank)
        local, allt = self._buffers((h, w, c, tile, overlap), chunk, chunk * self.world, th * s, tw * s, c, frame_u8.device)
        if count > 0:
            self.be.restore_tiles_t(frame_u8, h, w, c, tile, overlap, first, count, local)
        if self.world > 1:
            import torch.distributed as dist
            # restore_tiles_t is host-synchronous, so `local` is complete before RCCL reads it
            if frame_u8.is_cuda and dist.get_backend(self.group) == 'gloo':      # rehearsal on a one-GPU box: stage through the host
                host = torch.empty(allt.numel(), dtype=allt.dtype)
                dist.all_gather_into_tensor(host, local[:chunk].reshape(-1).cpu(), group=self.group)
                allt.view(-1).copy_(host)
            else:
                dist.all_gather_into_tensor(allt.view(-1), local[:chunk].reshape(-1), group=self.group)
            if frame_u8.is_cuda:
                torch.cuda.synchronize(frame_u8.device)     # RCCL stream -> visible to the library stream
        if out is None:
            out = torch.empty((h * s, w * s, c), dtype=torch.uint8, device=frame_u8.device)
        self.be.blend_tiles_t(allt, h, w, c, tile, overlap, out)
        return out


# ---------------------------------------------------------------------------------------------------------------------
# Whole-frame (untiled) Restormer over the GPUs of a node: horizontal bands + halo rows + Gram all-gather.

def strip_bounds(Hp: int, world: int):
    """Rows [r0, r1) per rank: the Hp / 8 row groups of the coarsest U-Net level dealt out as evenly as possible, in order."""
    units = Hp // 8
    if Hp % 8 or units < world:
        raise ValueError(f"a padded height of {Hp} rows cannot be split into {world} bands of whole 8-row groups")
    base, rem = divmod(units, world)
    out, r = [], 0
    for i in range(world):
        n = 8 * (base + (1 if i < rem else 0))
        out.append((r, r + n))
        r += n
    return out


class TorchDistTransport:
    """The two exchange steps over torch.distributed (backend "nccl" = RCCL over xGMI).  ``via_host``: stage through host memory, for the
    gloo backend in tests."""

    def __init__(self, rank: int, world: int, group=None, via_host: bool = False):
        self.rank, self.world, self.group, self.via_host = rank, world, group, via_host

    def _peer(self, r):
        import torch.distributed as dist
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def exchange(self, send_top, recv_top, send_bot, recv_bot):
        import torch.distributed as dist
        h = (lambda t: t.cpu()) if self.via_host else (lambda t: t)
        ops, back = [], []
        if self.rank > 0:
            r = h(recv_top)
            ops += [dist.P2POp(dist.isend, h(send_top), self._peer(self.rank - 1), self.group), dist.P2POp(dist.irecv, r, self._peer(self.rank - 1), self.group)]
            back.append((recv_top, r))
        if self.rank < self.world - 1:
            r = h(recv_bot)
            ops += [dist.P2POp(dist.isend, h(send_bot), self._peer(self.rank + 1), self.group), dist.P2POp(dist.irecv, r, self._peer(self.rank + 1), self.group)]
            back.append((recv_bot, r))
        for w in dist.batch_isend_irecv(ops) if ops else []:
            w.wait()
        if self.via_host:
            for dst, src in back:
                dst.copy_(src)

    def allgather(self, send, recv):
        import torch.distributed as dist
        if self.via_host:
            r = recv.cpu()
            dist.all_gather_into_tensor(r, send.cpu(), group=self.group)
            recv.copy_(r)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)


class LocalTransport:
    """world == 1: nothing to exchange, the gather is a copy."""

    def exchange(self, send_top, recv_top, send_bot, recv_bot):
        pass

    def allgather(self, send, recv):
        recv.copy_(send)


class ThreadTransport:
    """Emulation for one GPU: every band is driven by its own host thread and model handle in ONE process, the exchange steps are
    device copies between the threads' staging tensors.  ``shared = ThreadTransport.make_shared(world)`` is common to all ranks."""

    @staticmethod
    def make_shared(world: int):
        import threading
        return {'barrier': threading.Barrier(world), 'slots': [None] * world}

    def __init__(self, shared, rank: int, world: int):
        self.sh, self.rank, self.world = shared, rank, world

    def _sync(self, t):
        if t.is_cuda:
            torch.cuda.synchronize(t.device)

    def exchange(self, send_top, recv_top, send_bot, recv_bot):
        self.sh['slots'][self.rank] = (send_top, send_bot)
        self.sh['barrier'].wait()
        if self.rank > 0:
            recv_top.copy_(self.sh['slots'][self.rank - 1][1])
        if self.rank < self.world - 1:
            recv_bot.copy_(self.sh['slots'][self.rank + 1][0])
        self._sync(recv_top)
        self.sh['barrier'].wait()

    def allgather(self, send, recv):
        self.sh['slots'][self.rank] = send
        self.sh['barrier'].wait()
        n = send.numel()
        for i in range(self.world):
            recv[i * n:(i + 1) * n].copy_(self.sh['slots'][i])
        self._sync(recv)
        self.sh['barrier'].wait()


class StripParallelRestorer:
    """Rank ``rank`` of ``world`` restores its band of the frame; ``restore_frame`` returns the whole restored frame on every rank.
    backend: RestormerBackend (strip_* methods).  transport: exchange(send_top, recv_top, send_bot, recv_bot) and allgather(send, recv)
    on torch tensors (default: TorchDistTransport over the default process group)."""

    def __init__(self, backend, rank: int = 0, world: int = 1, transport=None):
        self.be, self.rank, self.world = backend, rank, world
        self.tr = transport or (LocalTransport() if world == 1 else TorchDistTransport(rank, world))
        self._comm = None
        self._bufs = None
        self.error = None

    def _configure(self, Wp: int, device):
        edge, rec = self.be.strip_buffer_bytes(Wp)
        if self._bufs is not None and self._bufs['edge'] >= edge and self._bufs['device'] == device:
            return
        u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=device)
        b = {'edge': edge, 'rec': rec, 'device': device, 'st': u8(edge), 'rt': u8(edge), 'sb': u8(edge), 'rb': u8(edge),
             'gs': u8(rec), 'gr': u8(rec * self.world)}

        def guard(fn):
            def cb(_ctx, nbytes):
                try:
                    fn(int(nbytes))
                    if device.type == 'cuda':
                        torch.cuda.synchronize(device)
                    return 0
                except BaseException as e:      # must not propagate through the C frames
                    self.error = e
                    return 1
            return _capi.STRIP_FN(cb)
        ex = guard(lambda n: self.tr.exchange(b['st'][:n], b['rt'][:n], b['sb'][:n], b['rb'][:n]))
        ag = guard(lambda n: self.tr.allgather(b['gs'][:n], b['gr'][:n * self.world]))
        comm = _capi.IraStripComm(None, ex, ag, self.rank, self.world, b['st'].data_ptr(), b['rt'].data_ptr(), b['sb'].data_ptr(),
                                  b['rb'].data_ptr(), edge, b['gs'].data_ptr(), b['gr'].data_ptr(), rec)
        self.be.strip_configure(comm)
        self._bufs, self._comm, self._cbs = b, comm, (ex, ag)

    def restore_frame(self, frame_u8: torch.Tensor, return_float: bool = False) -> torch.Tensor:
        """frame_u8: (h,w,c) uint8 tensor on this rank's device (every rank holds the frame).  ``return_float``: the cropped float32
        network output instead of uint8 (tests).  The demo.py:120-161 harness in its
        whole-image mode (:129-131): pad to a multiple of 8 by reflection only where needed, forward, clamp, crop, img_as_ubyte."""
        import torch.nn.functional as F
        h, w, c = frame_u8.shape
        x = frame_u8.permute(2, 0, 1).unsqueeze(0).float().div(255.)
        H, W = ((h + 8) // 8) * 8, ((w + 8) // 8) * 8                                        # demo.py:123-126
        padh, padw = (H - h if h % 8 else 0), (W - w if w % 8 else 0)
        if padh or padw:
            x = F.pad(x, (0, padw, 0, padh), 'reflect')
        xp = x[0].permute(1, 2, 0).contiguous()                                             # (Hp, Wp, c) float32
        Hp, Wp = xp.shape[:2]
        bounds = strip_bounds(Hp, self.world)
        r0, r1 = bounds[self.rank]
        top, bot = (8 if self.rank > 0 else 0), (8 if self.rank < self.world - 1 else 0)
        self._configure(Wp, frame_u8.device)
        loc = xp[r0 - top:r1 + bot].contiguous()
        out_loc = torch.empty_like(loc)
        self.error = None
        try:
            self.be.strip_forward_device(loc.data_ptr(), loc.shape[0], Wp, top, bot, out_loc.data_ptr())
        except Exception:
            if self.error is not None:
                raise self.error
            raise
        rows_max = max(b - a for a, b in bounds)
        mine = torch.zeros((rows_max, Wp, c), dtype=torch.float32, device=frame_u8.device)
        mine[:r1 - r0] = out_loc[top:top + (r1 - r0)]
        if self.world > 1:
            allb = torch.empty((self.world * rows_max, Wp, c), dtype=torch.float32, device=frame_u8.device)
            self.tr.allgather(mine.view(-1).view(torch.uint8), allb.view(-1).view(torch.uint8))
            full = torch.cat([allb[i * rows_max:i * rows_max + (b - a)] for i, (a, b) in enumerate(bounds)], 0)
        else:
            full = mine[:Hp]
        if return_float:
            return full[:h, :w].clone()
        return full[:h, :w].clamp(0, 1).mul(255.).round().to(torch.uint8)
