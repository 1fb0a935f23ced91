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
            dist.all_gather_into_tensor(allt.view(-1), local[:chunk].reshape(-1), group=self.group)
            if frame_u8.is_cuda:
                torch.cuda.synchronize(frame_u8.device)     # RCCL stream -> visible to the library stream
        if out is None:
            out = torch.empty((h * s, w * s, c), dtype=torch.uint8, device=frame_u8.device)
        self.be.blend_tiles_t(allt, h, w, c, tile, overlap, out)
        return out
