"""Restoration of one frame across the GPUs of one node: one process per GPU, RCCL over xGMI, issued by libira_hip.so itself.

The reference runs the tiler as a Python double loop of batch-1 forwards on one GPU (demo.py:143-153; agentic_api.py:2 pins one device).
Tiles are independent forwards (per-tile attention statistics, per-tile zero padding), so the row-major tile list is block-partitioned over
ranks; each rank runs its tiles as ONE batched forward, a single ncclAllGather of the restored tiles (float32 NHWC; 40 x 3 x 512 x 512 x 4 B =
126 MB for a 4K frame) gives every rank the whole list, and the E/W overlap-average runs on each rank in the reference's row-major order
(bit-stable).  All of that is ``ira_restore_frame_distributed`` (include/ira.h); this module only creates the communicator - the 128-byte
RCCL id travels from rank 0 to the other ranks over a plain TCP socket - and mirrors the call.

No torch.distributed here: the classes take an optional TEST transport (ira_amd/test_transports.py) that replaces the library's RCCL calls
when there is no communicator (CPU tests with gloo, single-GPU emulations).
"""
from __future__ import annotations

import os
import socket
import struct
import time

import torch

from . import _capi


def exchange_unique_id(rank: int, world: int, make_id, addr: str | None = None, port: int | None = None, timeout: float = 120.0) -> bytes:
    """Rank 0 calls ``make_id()`` (-> 128 bytes) and serves it to the world - 1 other ranks over TCP; they connect (retrying while rank 0
    is not listening yet) and read it.  addr / port default to MASTER_ADDR / MASTER_PORT + 29 of the torchrun environment."""
    addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = port or int(os.environ.get('MASTER_PORT', '29500')) + 29
    if world == 1:
        return make_id()
    if rank == 0:
        uid = make_id()
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(struct.pack('<I', len(uid)) + uid)
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as s:
                n = struct.unpack('<I', _recvn(s, 4))[0]
                return _recvn(s, n)
        except (ConnectionRefusedError, socket.timeout, OSError):
            if time.time() > deadline:
                raise
            time.sleep(0.1)


def _recvn(s, n):
    buf = b''
    while len(buf) < n:
        chunk = s.recv(n - len(buf))
        if not chunk:
            raise ConnectionError('peer closed while sending the communicator id')
        buf += chunk
    return buf


def init_comm(backend, rank: int, world: int, addr: str | None = None, port: int | None = None):
    """Attach an RCCL communicator of ``world`` ranks to ``backend`` (one model handle on this rank's GPU)."""
    uid = exchange_unique_id(rank, world, _capi.comm_unique_id, addr, port)
    backend.comm_init(uid, rank, world)
    return backend


def block_partition(n_tiles: int, world: int, rank: int):
    """Contiguous blocks of ceil(n/world) tiles; trailing ranks may be short or empty.
    With equal chunks the all-gathered buffer holds the tiles in row-major order already.  (The same rule as
    csrc/ira_runtime.hip::ira_restore_frame_distributed.)"""
    chunk = (n_tiles + world - 1) // world
    first = min(rank * chunk, n_tiles)
    count = max(0, min(chunk, n_tiles - first))
    return first, count, chunk


class TileParallelRestorer:
    """One frame, tile-parallel.  With a communicator on the backend (``init_comm``) this is one call of
    ``ira_restore_frame_distributed``.  Without one: world 1 runs extract / forward / blend locally; world > 1 needs a test ``gather``
    (ira_amd/test_transports.py::GlooTileGather) and a backend with restore_tiles_t / blend_tiles_t (the CPU tests inject a fake)."""

    def __init__(self, backend, rank: int = 0, world: int = 1, plan_fn=None, gather=None):
        self.be = backend
        self.rank, self.world, self.gather = rank, world, gather
        # the plan of the model's own harness (SwinIR: always-pad rule, outputs scaled by `upscale`); plan functions may
        # return 6 values (scale 1) or 7
        self.plan_fn = plan_fn or getattr(backend, 'plan', None) or _capi.tile_plan
        self._bufs = {}
        if world > 1 and gather is None and not getattr(backend, 'has_comm', False):
            raise _capi.IraError("world > 1 needs an RCCL communicator on the backend (ira_amd.dist.init_comm) or a test gather transport")

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
        if out is None:
            out = torch.empty((h * s, w * s, c), dtype=torch.uint8, device=frame_u8.device)
        if self.gather is None and getattr(self.be, 'has_comm', False):
            self.be.restore_frame_distributed_t(frame_u8, tile, overlap, out)        # partition, forward, ncclAllGather, blend: in the library
            return out
        nt = n_rows * n_cols
        first, count, chunk = block_partition(nt, self.world, self.rank)
        local, allt = self._buffers((h, w, c, tile, overlap), chunk, chunk * self.world, th * s, tw * s, c, frame_u8.device)
        if count > 0:
            self.be.restore_tiles_t(frame_u8, h, w, c, tile, overlap, first, count, local)
        if self.world > 1:
            self.gather(local[:chunk].reshape(-1), allt.view(-1))
        self.be.blend_tiles_t(allt, h, w, c, tile, overlap, out)
        return out


# ---------------------------------------------------------------------------------------------------------------------
# Whole-frame (untiled) Restormer over the GPUs of a node: horizontal bands + halo rows + Gram all-gather.

def strip_bounds(Hp: int, world: int):
    """Rows [r0, r1) per rank: the Hp / 8 row groups of the coarsest U-Net level dealt out as evenly as possible, in order.
    (The same rule as csrc/ira_runtime.hip::strip_rows.)"""
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


class StripParallelRestorer:
    """Rank ``rank`` of ``world`` restores its band of the frame; ``restore_frame`` returns the whole restored frame on every rank.
    With a communicator on the backend (``init_comm``) and no transport this is one call of ``ira_strip_restore_u8_device``: band cut-out,
    /255, reflect padding, the band forward with ncclSend / ncclRecv / ncclAllGather on the model's stream, uint8 conversion and the
    gather of the bands all happen in the library.  ``transport`` (ira_amd/test_transports.py) selects the callback form of the C ABI
    instead (ira_strip_configure): exchange(send_top, recv_top, send_bot, recv_bot) and allgather(send, recv) on torch tensors."""

    def __init__(self, backend, rank: int = 0, world: int = 1, transport=None):
        self.be, self.rank, self.world = backend, rank, world
        self.use_comm = transport is None and getattr(backend, 'has_comm', False)
        if transport is None and not self.use_comm:
            if world > 1:
                raise _capi.IraError("world > 1 needs an RCCL communicator on the backend (ira_amd.dist.init_comm) or a test transport")
            from .test_transports import LocalTransport
            transport = LocalTransport()
        self.tr = transport
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
        if self.use_comm and not return_float:
            out = torch.empty_like(frame_u8)
            self.be.strip_restore_u8_t(frame_u8, out)
            return out
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
