"""Multi-GPU ALM: one process per GPU, the WHOLE state sharded over TIME SLABS, the Laplacian solve over TIME MODES.

Every operator of an ALM iteration is local in the time index except nearest-neighbour couplings
(``solver_socp.py:161-170`` ``kron(eye(n_time), .)``; ``:884``, ``:892-894`` grad / div in time; ``:934-940``, ``:955-957`` the
``B[t]`` / ``B[t+1]`` pairing of the corner variables), and the space-time Laplacian decouples exactly over the time MODES
(the reference's ``eigh``, ``utils/laplacian_inverse_socp.py:31-41``).  With ``s = ceil((T+1)/R)``, rank r

  * holds the nodes ``[r s, (r+1) s)`` of phi, B, E, the intervals that start at them of A, lambda_c, mu, z_*, beta_* and
    the corner entries compared with its nodes' B  (device bytes: state / R),
  * factorises and solves the time modes ``[r s, (r+1) s)``  (factor bytes / R, no communication inside the solve).

One iteration (``dots_slab_stage`` 0-3, include/dots_socp_hip.h) has three exchanges:

    stage 0    pack two V-sized halos                      -> neighbour exchange (xGMI point-to-point):
               (A + lambda_c - mu) of the last interval forward, the half of the cone norms that pairs with the first
               node's B backward
    stage 1    right-hand side + cone projection           -> all-gather of the right-hand side  (8 (T+1) V bytes in total;
               each rank appends its last interval's cone multipliers for the next slab)
    stage 2    forward transform of own modes + sweeps     -> all-gather of the mode-space solution  (8 (T+1) V bytes)
    stage 3    inverse transform for own nodes (+ the next slab's first node, computed redundantly: phi needs no halo),
               steps 2 and 3

The KKT residuals and the objective are sums over space-time: each rank forms the sums of its slab (after one more halo
exchange for the time stencils inside them) and ONE all-reduce of 24 doubles per evaluation adds them
(``solver_socp.py:433-559``); every rank then takes the same control decisions.  Iterates are bit-identical for every
number of ranks (wherever a value is computed, the same sum is formed in the same order); only those sums differ in
rounding.

``ShardedAlmSolver`` is ``AlmSolver`` with the device step and every read-back replaced.  Communicators: ``TorchComm``
(torch.distributed; "nccl" = RCCL on ROCm with device tensors; with "gloo" the payload is staged through the host) and
``ThreadComm`` (ranks as threads of one process sharing one GPU, for single-GPU tests).
"""
from __future__ import annotations

import threading
import time

import numpy as np

from .device import STATE_NAMES
from .socp.solver_socp import AlmSolver, DEFAULT_CG_TOL


def slab_partition(n_nodes: int, n_ranks: int):
    """(stride, [(begin, count)] per rank): count <= stride = ceil(n_nodes / n_ranks); trailing ranks may be empty."""
    stride = -(-n_nodes // n_ranks)
    out = []
    for r in range(n_ranks):
        b = min(r * stride, n_nodes)
        out.append((b, max(0, min(stride, n_nodes - b))))
    return stride, out


mode_partition = slab_partition      # the time modes are divided the same way


class TorchComm:
    """Exchanges over torch.distributed (backend "nccl" is RCCL on ROCm)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.calls = {"all_gather": 0, "exchange": 0, "all_reduce": 0, "flag": 0}

    def _peer(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def all_gather(self, recv, send, sync=True):
        """``sync=False`` (device buffers): the result is only ordered on the current torch stream."""
        import torch

        self.calls["all_gather"] += 1
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

    def exchange(self, n_active, fwd=None, bwd=None, sync=True):
        """Nearest-neighbour exchange among the ranks ``[0, n_active)``.  ``fwd = (send, recv)``: ``send`` goes to rank + 1,
        ``recv`` is filled from rank - 1; ``bwd = (send, recv)``: to rank - 1, from rank + 1.  Ranks >= n_active idle."""
        import torch

        self.calls["exchange"] += 1
        r, dist = self.rank, self.dist
        if r >= n_active:
            return
        ops, staged = [], []

        def tensor_for(t, receiving):
            if self.backend == "nccl":
                return t
            h = torch.empty(t.numel(), dtype=t.dtype) if receiving else t.detach().cpu()
            if receiving:
                staged.append((t, h))
            return h

        if fwd is not None:
            if r + 1 < n_active:
                ops.append(dist.P2POp(dist.isend, tensor_for(fwd[0], False), self._peer(r + 1), group=self.group))
            if r > 0:
                ops.append(dist.P2POp(dist.irecv, tensor_for(fwd[1], True), self._peer(r - 1), group=self.group))
        if bwd is not None:
            if r > 0:
                ops.append(dist.P2POp(dist.isend, tensor_for(bwd[0], False), self._peer(r - 1), group=self.group))
            if r + 1 < n_active:
                ops.append(dist.P2POp(dist.irecv, tensor_for(bwd[1], True), self._peer(r + 1), group=self.group))
        if not ops:
            return
        for req in dist.batch_isend_irecv(ops):
            req.wait()          # nccl: orders the current stream after the transfer; gloo: blocks the host
        for dst, h in staged:
            dst.copy_(h.to(dst.device))
        if self.backend == "nccl" and sync:
            torch.cuda.current_stream().synchronize()

    def all_reduce_sum(self, values: np.ndarray) -> np.ndarray:
        import torch

        self.calls["all_reduce"] += 1
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.as_tensor(np.ascontiguousarray(values, dtype=np.float64)).to(dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy().reshape(np.shape(values))

    def any_flag(self, flag: bool) -> bool:
        import torch

        self.calls["flag"] += 1
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
        self.calls = {"all_gather": 0, "exchange": 0, "all_reduce": 0, "flag": 0}

    @classmethod
    def group(cls, n):
        shared = cls._Shared(n)
        return [cls(shared, r) for r in range(n)]

    def all_gather(self, recv, send, sync=True):
        import torch

        self.calls["all_gather"] += 1
        s = self.shared
        if send.is_cuda:
            torch.cuda.current_stream().synchronize()     # the payload is complete before another thread's stream reads it
        s.slots[self.rank] = send
        s.barrier.wait()
        recv.copy_(torch.cat([x.reshape(-1) for x in s.slots]))
        if recv.is_cuda:
            torch.cuda.current_stream().synchronize()
        s.barrier.wait()

    def exchange(self, n_active, fwd=None, bwd=None, sync=True):
        import torch

        self.calls["exchange"] += 1
        s, r = self.shared, self.rank
        for pair in (fwd, bwd):
            if pair is not None and pair[0].is_cuda:
                torch.cuda.current_stream().synchronize()
        s.slots[r] = (fwd[0] if fwd is not None else None, bwd[0] if bwd is not None else None)
        s.barrier.wait()
        if r < n_active:
            if fwd is not None and r > 0:
                fwd[1].copy_(s.slots[r - 1][0])
            if bwd is not None and r + 1 < n_active:
                bwd[1].copy_(s.slots[r + 1][1])
            if torch.cuda.is_available():
                torch.cuda.current_stream().synchronize()
        s.barrier.wait()

    def all_reduce_sum(self, values):
        self.calls["all_reduce"] += 1
        s = self.shared
        s.slots[self.rank] = np.array(values, dtype=np.float64)
        s.barrier.wait()
        total = np.zeros_like(s.slots[0])
        for x in s.slots:          # fixed order: identical on every rank
            total = total + x
        s.barrier.wait()
        return total

    def any_flag(self, flag):
        self.calls["flag"] += 1
        s = self.shared
        s.flags[self.rank] = bool(flag)
        s.barrier.wait()
        out = any(s.flags)
        s.barrier.wait()
        return out

    def barrier(self):
        self.shared.barrier.wait()


class ShardedAlmSolver(AlmSolver):
    """``AlmSolver`` on ``comm.size`` ranks: the state in time slabs, the Laplacian solve over the time modes."""

    def __init__(self, n_time, geometry, comm=None, device=0, buffer_device=None, **kw):
        import torch

        if comm is None:
            comm = TorchComm()
        self.comm = comm
        kw.setdefault("cg_tol", DEFAULT_CG_TOL)
        kw.setdefault("lap_solver", "modal_direct")
        if kw["lap_solver"] not in ("modal_direct", "modal_pcg"):
            raise ValueError("the sharded solver works on the time modes: lap_solver must be 'modal_direct' or 'modal_pcg'")
        if kw.get("is_constant_scaling"):
            raise ValueError("is_constant_scaling needs norms of whole arrays before the first exchange: not available on time slabs")
        self._init_full = kw.pop("init_solution", None) or {}
        super().__init__(n_time, geometry, device=device, time_slab=(comm.rank, comm.size), **kw)
        dev = self.dev
        tdev = torch.device(buffer_device) if buffer_device is not None else torch.device("cuda", device)
        nv, nf = dev.slab_elems("vertex_halo"), dev.slab_elems("triangle_halo")
        nb, nx = dev.slab_elems("b_chunk"), dev.slab_elems("x_chunk")
        z = lambda n: torch.zeros(n, dtype=torch.float64, device=tdev)       # noqa: E731
        self.buf = {"send_x": z(nv), "send_nsq": z(nv), "recv_x": z(nv), "recv_nsq": z(nv), "b_send": z(nb), "b_recv": z(nb * comm.size),
                    "x_send": z(nx), "x_recv": z(nx * comm.size), "send_mu": z(nv), "send_b": z(nf), "recv_mu": z(nv), "recv_b": z(nf)}
        dev.slab_set_buffers(**{k: t.data_ptr() for k, t in self.buf.items()})
        self.n_active = dev.active_ranks
        self._on_device = tdev.type == "cuda"
        self._kkt_halo_fresh = False
        self.comm_seconds = 0.0
        self.clock_exchanges = 0
        if self._init_full:
            self._upload_initial_slab(self._init_full)

    # ---- initial state: every rank receives the whole arrays and keeps its slab
    def _upload_initial_state(self, init):
        if init:     # (AlmSolver.__init__ calls this with the popped, i.e. empty, dict)
            raise ValueError("pass init_solution to ShardedAlmSolver, not to the base class")

    def _upload_initial_slab(self, init):
        dev = self.dev
        need = {"phi", "A", "B"}
        if not need.issubset(k for k, v in init.items() if v is not None):
            raise ValueError("a sharded warm start needs phi, A and B in init_solution (the derived defaults need whole arrays)")
        unknown = set(init) - set(STATE_NAMES) - {"checkpoints"}
        if unknown:
            raise ValueError(f"unknown init_solution entries: {sorted(unknown)}")
        # the scalings applied in __init__ (initial z scaling) were applied to a zero state: redo them on the uploaded one
        have = {k: np.asarray(v, dtype=np.float64) for k, v in init.items() if k in STATE_NAMES and v is not None}
        r, sz = self.r, self.scale_z
        for k in ("phi", "A", "B", "lambda_c"):
            if k in have:
                dev.upload(k, dev.to_slab(k, have[k]))
        for k in ("z_fst", "z_mid", "z_end"):
            if k in have:
                dev.upload(k, dev.to_slab(k, sz * have[k]))
        betas = {k: (1.0 / (r * sz)) * have[k] for k in ("beta_fst", "beta_mid", "beta_end") if k in have}
        for k, v in betas.items():
            dev.upload(k, dev.to_slab(k, v))
        # mu and E follow from beta as in scale_variable_z (solver_socp.py:386-388)
        dev.scale_z(1.0, 1.0, sz)

    # ---- one iteration: four stages, three exchanges
    def _order(self, ctx_waits):
        import torch

        if self._on_device:
            self.dev.stream_wait(torch.cuda.current_stream(self.buf["b_send"].device).cuda_stream, ctx_waits=ctx_waits)
        elif not ctx_waits:
            self.dev.sync()        # host-staged buffers: the context's work must be complete before the host reads them

    def _device_step(self, quiet=False):
        dev, comm, buf = self.dev, self.comm, self.buf
        dev.step_flags(skip_z_mid=quiet and not self.is_palm, palm=self.is_palm)
        enqueue_only = quiet and self._on_device and self.direct
        wait = not enqueue_only
        self._kkt_halo_fresh = False
        t_comm = 0.0
        stats = []
        stats.append(dev.slab_stage(0, wait=wait))
        t0 = time.perf_counter()
        self._order(ctx_waits=False)
        comm.exchange(self.n_active, fwd=(buf["send_x"], buf["recv_x"]), bwd=(buf["send_nsq"], buf["recv_nsq"]), sync=wait)
        self._order(ctx_waits=True)
        t_comm += time.perf_counter() - t0
        stats.append(dev.slab_stage(1, wait=wait))
        t0 = time.perf_counter()
        self._order(ctx_waits=False)
        comm.all_gather(buf["b_recv"], buf["b_send"], sync=wait)
        self._order(ctx_waits=True)
        t_comm += time.perf_counter() - t0
        stats.append(dev.slab_stage(2, wait=wait))
        t0 = time.perf_counter()
        self._order(ctx_waits=False)
        comm.all_gather(buf["x_recv"], buf["x_send"], sync=wait)
        self._order(ctx_waits=True)
        t_comm += time.perf_counter() - t0
        stats.append(dev.slab_stage(3, wait=wait))
        if quiet:
            self.quiet_steps += 1
        if enqueue_only:
            self.untimed_steps += 1
            return
        st = stats[0]
        for other in stats[1:]:
            for name in ("ms_rhs", "ms_laplacian", "ms_soc", "ms_q_lambda_multiplier", "ms_total", "alm_iterations", "cg_iterations"):
                setattr(st, name, getattr(st, name) + getattr(other, name))
        st.cg_not_converged = sum(s.cg_not_converged for s in stats)
        self._account(st)
        self.comm_seconds += t_comm
        self.run_history.add_time("Exchange (halos + two all-gathers)", 0.0)
        self.run_history.steps_time["Exchange (halos + two all-gathers)"] = self.comm_seconds

    # ---- scaling tools change mu / B-independent duals in place: the KKT halos of the neighbours are stale afterwards
    def adjust_penalty(self, factor):
        super().adjust_penalty(factor)
        self._kkt_halo_fresh = False

    def scale_variable_z(self, scale_factor, msg="Scale z"):
        super().scale_variable_z(scale_factor, msg=msg)
        self._kkt_halo_fresh = False

    # ---- read-backs: sums over the slabs
    def _refresh_kkt_halos(self):
        if self._kkt_halo_fresh:
            return
        dev, buf = self.dev, self.buf
        dev.slab_stage(4, wait=False)
        self._order(ctx_waits=False)
        self.comm.exchange(self.n_active, fwd=(buf["send_mu"], buf["recv_mu"]), bwd=(buf["send_b"], buf["recv_b"]), sync=True)
        self._order(ctx_waits=True)
        self._kkt_halo_fresh = True

    def _kkt(self, conditions):
        self._refresh_kkt_halos()
        total = self.comm.all_reduce_sum(self.dev.kkt_sums(conditions))
        return self.dev.kkt_combine(conditions, total)

    def _objective(self):
        return self.dev.objective_combine(self.comm.all_reduce_sum(self.dev.objective_sums()))

    def _download(self, name):
        """The whole array: every rank contributes its slab (the entries of a whole array belong to exactly one rank)."""
        dev = self.dev
        full = np.zeros(dev.full_shape(name))
        if dev.nl > 0:
            dev.from_slab(name, dev.download(name), full)
        return self.comm.all_reduce_sum(full)

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
