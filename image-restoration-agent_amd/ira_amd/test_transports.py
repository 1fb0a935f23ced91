"""TEST TRANSPORTS for ira_amd.dist: stand-ins for the RCCL exchange steps that libira_hip.so issues itself (csrc/ira_comm.hip).

Product runs never import this module: with a communicator attached to the handle (``dist.init_comm``) the tile all-gather, the band halo
exchange and the Gram-record all-gather are ncclAllGather / ncclSend / ncclRecv calls on the model's stream inside the library.  What lives
here exists so that the host-side logic can be exercised without several GPUs:

* ``GlooTileGather`` / ``TorchDistTransport``: the same exchange steps over torch.distributed (gloo on CPU in tests/test_dist_gloo.py;
  staged through the host for rehearsals of bench.py with every rank on one GPU);
* ``ThreadTransport``: all bands of a frame driven by host threads of ONE process on one GPU (tests/test_gpu_strip.py);
* ``LocalTransport``: world == 1.
"""
from __future__ import annotations

import torch


class GlooTileGather:
    """all-gather of the equal tile chunks of TileParallelRestorer over torch.distributed.  ``via_host``: CUDA tensors are staged through host
    memory (gloo with every rank on one GPU)."""

    def __init__(self, group=None, via_host: bool = False):
        self.group, self.via_host = group, via_host

    def __call__(self, local_flat: torch.Tensor, all_flat: torch.Tensor):
        import torch.distributed as dist
        if self.via_host and local_flat.is_cuda:
            host = torch.empty(all_flat.numel(), dtype=all_flat.dtype)
            dist.all_gather_into_tensor(host, local_flat.cpu(), group=self.group)
            all_flat.copy_(host)
        else:
            dist.all_gather_into_tensor(all_flat, local_flat, group=self.group)
            if local_flat.is_cuda:
                torch.cuda.synchronize(local_flat.device)


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


