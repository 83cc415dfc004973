"""Multi-GPU ALM: one process per GPU, the WHOLE state sharded over TIME SLABS, the Laplacian solve over TIME MODES.

Every operator of an ALM iteration is local in the time index except nearest-neighbour couplings
(``solver_socp.py:161-170`` ``kron(eye(n_time), .)``; ``:884``, ``:892-894`` grad / div in time; ``:934-940``, ``:955-957`` the
``B[t]`` / ``B[t+1]`` pairing of the corner variables), and the space-time Laplacian decouples exactly over the time MODES
(the reference's ``eigh``, ``utils/laplacian_inverse_socp.py:31-41``).  With ``s = ceil((T+1)/R)``, rank r

  * holds the nodes ``[r s, (r+1) s)`` of phi, B, E, the intervals that start at them of A, lambda_c, mu, z_*, beta_* and
    the corner entries compared with its nodes' B  (device bytes: state / R),
  * factorises and solves the time modes ``[r s, (r+1) s)``  (factor bytes / R, no communication inside the solve).

One iteration (``dots_slab_stage`` 0, 6, 5, 2, 3; include/dots_socp_hip.h) has three exchanges:

    stage 0    pack two V-sized halos                      -> neighbour exchange (xGMI point-to-point):
               (A + lambda_c - mu) of the last interval forward, the half of the cone norms that pairs with the first
               node's B backward
    stage 6    right-hand side                             -> all-gather of the right-hand side  (8 (T+1) V bytes in total)
    stage 5    cone projection, on the context's stream WHILE that all-gather runs (it reads nothing the exchange writes)
    stage 2    forward transform of own modes + sweeps     -> all-gather of the mode-space solution  (8 (T+1) V bytes; each
               rank appends its last interval's cone multipliers for the next slab's steps 2+3)
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

from ._lib import KKT_N_SUMS
from .device import STATE_NAMES, place_slab
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


def _nbytes(t):
    return int(t.numel()) * int(t.element_size())


class TorchComm:
    """Exchanges over torch.distributed (backend "nccl" is RCCL on ROCm).  ``device``: where the small payloads this class
    builds itself (flags, sums handed over as numpy arrays) live with the nccl backend -- the solver's own device
    (``ShardedAlmSolver`` sets it; default: torch's current device at the time of the call).  ``calls`` / ``bytes`` count the
    collectives and what this rank hands to them (send side; an all-gather counts its own chunk)."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device = device
        self.calls = {"all_gather": 0, "exchange": 0, "all_reduce": 0, "flag": 0, "gather_array": 0}
        self.bytes = {"all_gather": 0, "exchange": 0, "all_reduce": 0, "flag": 0, "gather_array": 0}

    def _peer(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def _payload_device(self):
        import torch

        if self.backend != "nccl":
            return torch.device("cpu")
        return torch.device(self.device) if self.device is not None else torch.device("cuda", torch.cuda.current_device())

    def all_gather(self, recv, send, sync=True):
        """``sync=False`` (device buffers): the result is only ordered on the current torch stream."""
        import torch

        self.calls["all_gather"] += 1
        self.bytes["all_gather"] += _nbytes(send)
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(recv, send, group=self.group)
            if sync:
                torch.cuda.current_stream(recv.device).synchronize()
        else:   # gloo: stage through the host
            h_send = send.detach().cpu()
            h_recv = torch.empty(recv.numel(), dtype=recv.dtype)
            self.dist.all_gather_into_tensor(h_recv, h_send, group=self.group)
            recv.copy_(h_recv.to(recv.device))
            if recv.is_cuda and sync:
                torch.cuda.current_stream(recv.device).synchronize()

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
            if not receiving:
                self.bytes["exchange"] += _nbytes(t)
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
            torch.cuda.current_stream((fwd or bwd)[0].device).synchronize()

    def all_reduce_sum(self, values) -> np.ndarray:
        """Sum over the ranks of a small vector; the result as a numpy array on every rank.  ``values``: a numpy array, or a
        torch tensor -- with the nccl backend a DEVICE tensor is reduced where it lies (the KKT sums go from the kernel that
        formed them straight to RCCL: no host round trip before the collective; the tensor holds the total afterwards)."""
        import torch

        self.calls["all_reduce"] += 1
        if isinstance(values, torch.Tensor):
            t = values if (self.backend == "nccl") == values.is_cuda else values.detach().to(self._payload_device())
            shape = tuple(values.shape)
        else:
            shape = np.shape(values)
            t = torch.as_tensor(np.ascontiguousarray(values, dtype=np.float64)).to(self._payload_device())
        self.bytes["all_reduce"] += _nbytes(t)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy().reshape(shape)

    def all_gather_array(self, part: np.ndarray) -> np.ndarray:
        """Every rank's ``part`` (same shape and dtype on all ranks), stacked along a new first axis, on every rank."""
        import torch

        self.calls["gather_array"] += 1
        part = np.ascontiguousarray(part)
        send = torch.from_numpy(part).reshape(-1).to(self._payload_device())
        self.bytes["gather_array"] += _nbytes(send)
        recv = torch.empty(send.numel() * self.size, dtype=send.dtype, device=send.device)
        self.dist.all_gather_into_tensor(recv, send, group=self.group)
        return recv.cpu().numpy().reshape((self.size,) + part.shape)

    def any_flag(self, flag: bool) -> bool:
        import torch

        self.calls["flag"] += 1
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self._payload_device())
        self.bytes["flag"] += _nbytes(t)
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
        self.device = None
        self.calls = {"all_gather": 0, "exchange": 0, "all_reduce": 0, "flag": 0, "gather_array": 0}
        self.bytes = {"all_gather": 0, "exchange": 0, "all_reduce": 0, "flag": 0, "gather_array": 0}

    @classmethod
    def group(cls, n):
        shared = cls._Shared(n)
        return [cls(shared, r) for r in range(n)]

    def all_gather(self, recv, send, sync=True):
        import torch

        self.calls["all_gather"] += 1
        self.bytes["all_gather"] += _nbytes(send)
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
        if r < n_active:
            self.bytes["exchange"] += (_nbytes(fwd[0]) if fwd is not None and r + 1 < n_active else 0) + (_nbytes(bwd[0]) if bwd is not None and r > 0 else 0)
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
        if not isinstance(values, np.ndarray) and hasattr(values, "detach"):      # a torch tensor (device or host)
            values = values.detach().cpu().numpy()
        s.slots[self.rank] = np.array(values, dtype=np.float64)
        self.bytes["all_reduce"] += s.slots[self.rank].nbytes
        s.barrier.wait()
        total = np.zeros_like(s.slots[0])
        for x in s.slots:          # fixed order: identical on every rank
            total = total + x
        s.barrier.wait()
        return total

    def all_gather_array(self, part):
        self.calls["gather_array"] += 1
        s = self.shared
        s.slots[self.rank] = np.ascontiguousarray(part)
        self.bytes["gather_array"] += s.slots[self.rank].nbytes
        s.barrier.wait()
        out = np.stack(list(s.slots))
        s.barrier.wait()
        return out

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

    EXCHANGE_TAG = "Exchange (halos + two all-gathers)"

    def __init__(self, n_time, geometry, comm=None, device=0, buffer_device=None, **kw):
        import torch

        if comm is None:
            comm = TorchComm()
        self.comm = comm
        kw.setdefault("cg_tol", DEFAULT_CG_TOL)
        kw.setdefault("lap_solver", "modal_direct")
        if kw["lap_solver"] not in ("modal_direct", "modal_pcg"):
            raise ValueError("the sharded solver works on the time modes: lap_solver must be 'modal_direct' or 'modal_pcg'")
        self._init_full = kw.pop("init_solution", None) or {}
        if kw.get("is_constant_scaling") and self._init_full:
            raise ValueError("a sharded warm start cannot be combined with is_constant_scaling (the initial scaling is applied to the zero state)")
        tdev = torch.device(buffer_device) if buffer_device is not None else torch.device("cuda", device)
        # the small payloads the communicator builds itself must live on THIS solver's device (RCCL works on the device of
        # its tensors: a rank that only passed device=k must not all-reduce on GPU 0)
        if tdev.type == "cuda":
            if getattr(comm, "device", None) is None:
                comm.device = tdev
            elif torch.device(comm.device) != tdev:
                raise ValueError(f"the communicator works on {comm.device}, the solver on {tdev}")
        super().__init__(n_time, geometry, device=device, time_slab=(comm.rank, comm.size), **kw)
        dev = self.dev
        nv, nf = dev.slab_elems("vertex_halo"), dev.slab_elems("triangle_halo")
        nb, nx = dev.slab_elems("b_chunk"), dev.slab_elems("x_chunk")
        z = lambda n: torch.zeros(n, dtype=torch.float64, device=tdev)       # noqa: E731
        self.buf = {"send_x": z(nv), "send_nsq": z(nv), "recv_x": z(nv), "recv_nsq": z(nv), "b_send": z(nb), "b_recv": z(nb * comm.size),
                    "x_send": z(nx), "x_recv": z(nx * comm.size), "send_mu": z(nv), "send_b": z(nf), "recv_mu": z(nv), "recv_b": z(nf)}
        dev.slab_set_buffers(**{k: t.data_ptr() for k, t in self.buf.items()})
        self.kkt_buf = z(KKT_N_SUMS)          # the slab's KKT sums, handed to the all-reduce where the kernels leave them
        self.n_active = dev.active_ranks
        self._on_device = tdev.type == "cuda"
        self._tdev = tdev
        self._kkt_halo_fresh = False
        self.comm_seconds = 0.0
        self.clock_exchanges = 0
        self._comm_events = []        # (kind, [(start, end) torch events of the three exchanges]) of timed iterations in flight
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
            self.dev.stream_wait(torch.cuda.current_stream(self._tdev).cuda_stream, ctx_waits=ctx_waits)
        elif not ctx_waits:
            self.dev.sync()        # host-staged buffers: the context's work must be complete before the host reads them

    def _device_step(self, quiet=False):
        """With device buffers and the direct solver the four stages and the three exchanges are only ENQUEUED (context stream
        and communication stream ordered by events, no host wait: read-back iterations wait once, at their KKT sums).  Sampled
        iterations bracket the stages (DOTS_STEP_TIMED) and the exchanges (events on the communication stream)."""
        import torch

        dev, comm, buf = self.dev, self.comm, self.buf
        kind = "quiet" if quiet else "read-back"
        sample = self.step_timers.begin(kind)
        async_ok = self._on_device and self.direct
        timed = sample and async_ok and len(self._timed_in_flight) <= 55
        dev.step_flags(skip_z_mid=quiet and not self.is_palm, palm=self.is_palm, timed=timed, carry=self._carry)
        wait = not async_ok
        self._kkt_halo_fresh = False
        if quiet:
            self.quiet_steps += 1
        if async_ok and not timed:
            self.untimed_steps += 1
        events = []

        def exchange(fn, between=None):
            """``between``: device work enqueued on the context's stream AFTER the exchange has been handed to the communication
            stream and BEFORE the context waits for it: the two overlap."""
            t0 = time.perf_counter()
            self._order(ctx_waits=False)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream(self._tdev))
            fn()
            if timed:
                e1.record(torch.cuda.current_stream(self._tdev))
                events.append((e0, e1))
            dt = time.perf_counter() - t0
            if between is not None:
                between()
            t1 = time.perf_counter()
            self._order(ctx_waits=True)
            return dt + time.perf_counter() - t1

        try:
            self._stages_and_exchanges(dev, comm, buf, wait, timed, kind, exchange, events)
        except Exception:
            self._timed_in_flight.clear()      # (see AlmSolver._device_step: the ring and these lists no longer match)
            self._comm_events.clear()
            raise

    def _stages_and_exchanges(self, dev, comm, buf, wait, timed, kind, exchange, events):
        t_comm = 0.0
        stats = [dev.slab_stage(0, wait=wait)]
        t_comm += exchange(lambda: comm.exchange(self.n_active, fwd=(buf["send_x"], buf["recv_x"]), bwd=(buf["send_nsq"], buf["recv_nsq"]), sync=wait))
        # stage 1 in two halves: the all-gather of the right-hand side starts as soon as it is packed (stage 6) and the cone
        # projection (stage 5: it reads nothing the exchange or the solve writes) runs behind it on the context's stream
        stats.append(dev.slab_stage(6, wait=wait))
        t_comm += exchange(lambda: comm.all_gather(buf["b_recv"], buf["b_send"], sync=wait), between=lambda: stats.append(dev.slab_stage(5, wait=wait)))
        stats.append(dev.slab_stage(2, wait=wait))
        t_comm += exchange(lambda: comm.all_gather(buf["x_recv"], buf["x_send"], sync=wait))
        stats.append(dev.slab_stage(3, wait=wait))
        if timed:
            self._timed_in_flight.extend([kind] * 5)
            self._comm_events.append((kind, events))
        if wait:        # host-staged exchanges or the PCG: every stage was waited for and timed, the exchanges by the host's clock
            for st in stats:
                self._account(st, kind)
            self.step_timers.add(kind, self.EXCHANGE_TAG, t_comm)
            self.comm_seconds += t_comm

    def _collect_step_times(self, wait=False):
        fresh = False
        while self._comm_events and (wait or self._comm_events[0][1][-1][1].query()):
            fresh = True
            kind, events = self._comm_events.pop(0)
            if wait:
                events[-1][1].synchronize()
            seconds = 1e-3 * sum(e0.elapsed_time(e1) for e0, e1 in events)
            self.step_timers.add(kind, self.EXCHANGE_TAG, seconds)
            self.comm_seconds += seconds
        super()._collect_step_times(wait=wait)
        if fresh and not wait:
            self.step_timers.publish()

    # ---- scaling tools change mu / B-independent duals in place: the KKT halos of the neighbours are stale afterwards
    def adjust_penalty(self, factor):
        super().adjust_penalty(factor)
        self._kkt_halo_fresh = False

    def scale_variable_z(self, scale_factor, msg="Scale z"):
        super().scale_variable_z(scale_factor, msg=msg)
        self._kkt_halo_fresh = False

    def scale_prim_dual(self, scale_factor=None):
        super().scale_prim_dual(scale_factor)
        self._kkt_halo_fresh = False

    # ---- is_constant_scaling on time slabs (solver_socp.py:324-365): every norm is a sum over space-time -- each rank forms
    # the share of its slab (dots_norm_square on a slab), ONE small all-reduce adds the shares of all the norms asked for
    def _norm_squares(self, requests):
        shares = np.array([self.dev.norm_square(name, part) for name, part in requests])
        return [float(x) for x in self.comm.all_reduce_sum(shares)]

    def _boundary_gradient_norm(self, bt, phi0):
        """sqrt(|d_t b|^2 + |d_x b|^2) of the boundary term b (nonzero at the first and the last node only) in the norms of
        solver_socp.py:215-218: in closed form on the host, from the plan every rank holds (the base class uploads b and asks the
        device; a slab holds only its own nodes)."""
        p, T = self.dev.plan, self.n_time
        h = 1.0 / T
        mass, area = p.mass_vert, p.area_tri
        b0, bT = -p.mu0 / (h * mass), p.mu1 / (h * mass)
        if T == 1:
            n1 = float(np.sum(((bT - b0) / h) ** 2 * mass)) / T
        else:
            n1 = float(np.sum((b0 / h) ** 2 * mass) + np.sum((bT / h) ** 2 * mass)) / T
        hat, tri = p.hat_grad.reshape(-1, 3, 3), p.triangles.reshape(-1, 3)

        def grad_space_sq(x):
            g = np.einsum("fkc,fk->fc", hat, x[tri])
            return float(np.sum(g ** 2 * area[:, None]))

        return float(np.sqrt(n1 + (grad_space_sq(b0) + grad_space_sq(bT)) / (T + 1)))

    # ---- read-backs: sums over the slabs
    def _refresh_kkt_halos(self):
        if self._kkt_halo_fresh:
            return
        dev, buf = self.dev, self.buf
        dev.slab_stage(4, wait=False)
        self._order(ctx_waits=False)
        self.comm.exchange(self.n_active, fwd=(buf["send_mu"], buf["recv_mu"]), bwd=(buf["send_b"], buf["recv_b"]), sync=not self._on_device)
        self._order(ctx_waits=True)
        self._kkt_halo_fresh = True

    def _kkt(self, conditions):
        self._refresh_kkt_halos()
        if self._on_device:       # kernels -> device buffer -> all-reduce -> ONE copy to the host
            self.dev.kkt_sums_device(conditions, self.kkt_buf.data_ptr())
            self._order(ctx_waits=False)
            total = self.comm.all_reduce_sum(self.kkt_buf)
        else:
            total = self.comm.all_reduce_sum(self.dev.kkt_sums(conditions))
        return self.dev.kkt_combine(conditions, total)

    def _objective(self):
        return self.dev.objective_combine(self.comm.all_reduce_sum(self.dev.objective_sums()))

    def _download(self, name):
        """The whole array on every rank: an all-gather of the slabs (each entry of a whole array belongs to exactly one rank;
        a rank hands over its own 1 / R of the array, padded to the common slab length)."""
        dev = self.dev
        stride, parts = slab_partition(self.n_time + 1, self.comm.size)
        part = np.zeros((stride,) + tuple(dev.full_shape(name)[1:]))
        if dev.nl > 0:
            own = dev.download(name)
            part[:own.shape[0]] = own
        gathered = self.comm.all_gather_array(part)
        full = np.zeros(dev.full_shape(name))
        for begin, count in parts:
            if count > 0:
                place_slab(name, gathered[begin // stride], full, begin, count, max(0, min(count, self.n_time - begin)))
        return full

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
