"""Multi-GPU ALM: one process per GPU, the Laplacian solve sharded over TIME MODES.

Why modes and not time slabs.  Step 1 dominates an ALM iteration (>= 80 % of the time) and is a PCG with
20-1000 iterations, each with two global inner products.  Sharding the unknowns over time slabs would put two
latency-bound all-reduces over xGMI (tens of microseconds each) into every PCG iteration whose own cost is
~20-400 us.  The time direction, however, is diagonalised exactly by the DCT-II basis (the reference does the
same with ``eigh``, utils/laplacian_inverse_socp.py:31): in mode space the T+1 surface problems are independent.
So rank p solves modes [p*m, (p+1)*m), m = ceil((T+1)/P), with NO communication inside the solve, and the ranks
exchange the mode-space solution ONCE per ALM iteration:

    every rank:   right-hand side (full, replicated)  ->  forward transform restricted to its own modes
                  ->  batched PCG (+ multigrid) on its modes
    all-gather    [V][m] doubles per rank  (N*8 bytes in total: 2.6 MB at 10k vertices, 26 MB at 100k; one RCCL
                  all-gather over xGMI, bandwidth-bound, not latency-bound)
    every rank:   inverse transform of the gathered solution -> phi;  cone projection, (q, lambda) + multiplier
                  update, KKT sums on the full state (replicated: these steps are < 15 % of an iteration and
                  replicating them removes every halo exchange and keeps all ranks bit-identical, so the host
                  control logic takes the same decisions everywhere without any further collective)

What scales: the solve (bandwidth per rank / P).  What does not: the replicated element-wise steps (Amdahl);
sharding those over time slabs too (three nearest-neighbour halos per iteration) is the next step and does not
change the exchange described here.

``ShardedAlmSolver`` is ``AlmSolver`` with ``_device_step`` replaced by begin / all-gather / end.
Communicators: ``TorchComm`` (torch.distributed; "nccl" = RCCL on ROCm, device tensors; with "gloo" the payload is
staged through the host) and ``ThreadComm`` (ranks as threads of one process, for single-GPU tests).
"""
from __future__ import annotations

import threading
import time

import numpy as np

from .socp.solver_socp import AlmSolver, DEFAULT_CG_TOL


def mode_partition(n_modes: int, n_ranks: int):
    """[(begin, count)] per rank, count <= stride = ceil(n_modes / n_ranks); trailing ranks may be empty."""
    stride = -(-n_modes // n_ranks)
    out = []
    for r in range(n_ranks):
        b = min(r * stride, n_modes)
        out.append((b, max(0, min(stride, n_modes - b))))
    return stride, out


class TorchComm:
    """all-gather / flag exchange over torch.distributed (backend "nccl" is RCCL on ROCm)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    def all_gather(self, recv, send, sync=True):
        """``sync=False`` (device buffers): the result is only ordered on the current torch stream."""
        import torch

        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(recv, send, group=self.group)
            if sync:
                torch.cuda.current_stream().synchronize()
        else:   # gloo: stage through the host
            h_send = send.detach().cpu()
            h_recv = torch.empty(recv.numel(), dtype=recv.dtype)
            self.dist.all_gather_into_tensor(h_recv, h_send, group=self.group)
            recv.copy_(h_recv.to(recv.device))
            if recv.is_cuda and sync:
                torch.cuda.current_stream().synchronize()

    def any_flag(self, flag: bool) -> bool:
        import torch

        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return bool(t.item())

    def barrier(self):
        self.dist.barrier(group=self.group)


class ThreadComm:
    """Ranks are threads of one process sharing one GPU (tests).  Create with ``ThreadComm.group(n)``."""

    class _Shared:
        def __init__(self, n):
            self.n = n
            self.barrier = threading.Barrier(n)
            self.slots = [None] * n
            self.flags = [False] * n

    def __init__(self, shared, rank):
        self.shared, self.rank, self.size, self.backend = shared, rank, shared.n, "thread"

    @classmethod
    def group(cls, n):
        shared = cls._Shared(n)
        return [cls(shared, r) for r in range(n)]

    def all_gather(self, recv, send, sync=True):
        import torch

        s = self.shared
        s.slots[self.rank] = send
        s.barrier.wait()      # every rank has ordered the (shared) torch stream after its own context by now
        recv.copy_(torch.cat([x.reshape(-1) for x in s.slots]))
        if recv.is_cuda and sync:
            torch.cuda.synchronize()
        s.barrier.wait()

    def any_flag(self, flag):
        s = self.shared
        s.flags[self.rank] = bool(flag)
        s.barrier.wait()
        out = any(s.flags)
        s.barrier.wait()
        return out

    def barrier(self):
        self.shared.barrier.wait()


class ShardedAlmSolver(AlmSolver):
    """``AlmSolver`` whose Laplacian solve is sharded over the time modes of ``comm.size`` ranks."""

    def __init__(self, n_time, geometry, comm=None, device=0, buffer_device=None, **kw):
        import torch

        if comm is None:
            comm = TorchComm()
        self.comm = comm
        kw.setdefault("cg_tol", DEFAULT_CG_TOL)
        kw.setdefault("lap_solver", "modal_direct")
        if kw["lap_solver"] not in ("modal_direct", "modal_pcg"):
            raise ValueError("the sharded solver works on the time modes: lap_solver must be 'modal_direct' or 'modal_pcg'")
        super().__init__(n_time, geometry, device=device, mode_shard=(comm.rank, comm.size), **kw)
        elems = self.dev.shard_elems()
        dev = torch.device(buffer_device) if buffer_device is not None else torch.device("cuda", device)
        self._send = torch.zeros(elems, dtype=torch.float64, device=dev)
        self._recv = torch.zeros(elems * comm.size, dtype=torch.float64, device=dev)
        self.comm_seconds = 0.0
        self.clock_exchanges = 0

    def _device_step(self, quiet=False):
        import torch

        self.dev.step_flags(skip_z_mid=quiet and not self.is_palm, palm=self.is_palm)
        on_device = self._send.is_cuda and self.direct
        if on_device:
            # no host wait anywhere: the context's stream and the stream the collective runs on are ordered by events
            other = torch.cuda.current_stream(self._send.device).cuda_stream
            self.dev.step_begin(self._send.data_ptr(), self._send.numel(), wait=False)
            self.dev.stream_wait(other, ctx_waits=False)
            self.comm.all_gather(self._recv, self._send, sync=False)
            self.dev.stream_wait(other, ctx_waits=True)
            self.dev.step_end(self._recv.data_ptr(), self._recv.numel(), wait=False)
            self.untimed_steps += 1
            return
        st1 = self.dev.step_begin(self._send.data_ptr(), self._send.numel())
        t0 = time.perf_counter()
        self.comm.all_gather(self._recv, self._send)
        self.comm_seconds += time.perf_counter() - t0
        st2 = self.dev.step_end(self._recv.data_ptr(), self._recv.numel())
        for name in ("ms_soc", "ms_q_lambda_multiplier", "ms_total", "alm_iterations"):
            setattr(st1, name, getattr(st1, name) + getattr(st2, name))
        self._account(st1)
        self.run_history.add_time("Exchange (all-gather of the mode-space solution)", 0.0)
        self.run_history.steps_time["Exchange (all-gather of the mode-space solution)"] = self.comm_seconds

    def _time_is_up(self, reads_back=True):
        """Every rank must leave the loop on the same iteration, so the clock decision is shared (one flag
        all-reduce) -- but only on iterations that synchronise with the host anyway (KKT evaluation or penalty
        update: at least every 37th, control.AdaptiveValidator): enqueue-only iterations see no collective and no
        host wait because of a finite time_limit."""
        if not reads_back or self.time_limit is None or self.time_limit > 1e8:
            return False
        self.clock_exchanges += 1
        return self.comm.any_flag((time.perf_counter() - self.start_time) > self.time_limit)


def solver_socp_sharded(n_time, geometry, comm=None, device=0, nit=1000, **kw):
    """``solver_socp`` on ``comm.size`` GPUs (call it from every rank; all ranks return the same result)."""
    alm = ShardedAlmSolver(n_time, geometry, comm=comm, device=device, nit=nit, **kw)
    try:
        for _ in range(nit):
            if alm.iterate():
                break
        return alm.finalize()
    finally:
        alm.close()
