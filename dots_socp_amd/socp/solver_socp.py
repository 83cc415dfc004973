"""``solver_socp``: the reference's ALM driver with the numerics on the GPU.

Same signature, return contract and iteration-by-iteration decisions as
``dot_surface_socp/socp/solver_socp.py:25-871``; every array operation of the loop
(steps 1-3, KKT residuals, objective, scaling) runs in HIP kernels behind the C ABI
(``include/dots_socp_hip.h``), the state never leaves HBM between iterations and the host
only sees scalars.  Extra keyword arguments (all optional) select device-side choices:

    lap_solver   how step 1's Laplacian is solved: "modal_direct" (default; time eigen-modes + multifrontal Cholesky
                 sweeps, the reference's eigh + sparse-LU algorithm), "modal_pcg" (batched multigrid-PCG on the
                 modes) or "spacetime_pcg" (Jacobi-PCG on the coupled operator)
    cg_tol       relative PCG tolerance (default 1e-8; see DESIGN.md for the parity budget)
    cg_max_iter  PCG iteration cap
    device       HIP device ordinal
    reorder      locality renumbering of the mesh (default True)

There is no CPU path: without libdotsocp_hip.so and a GPU this raises.
"""
from __future__ import annotations

import logging
import math
import os
import time

import numpy as np

from ..control import (AdaptiveValidator, AdjustAdmmParam, ConditionValidator, ErrorCondition, RunningHistory, SampledStepTimers,
                       KKT_LABELS, KKT_SHORT_LABELS, max_of_list_with_none, safe_rescale_ratio)
from .. import _lib
from .._lib import env_choice
from ..device import DeviceProblem, STATE_NAMES

logger = logging.getLogger("dots_socp_amd")

KKT_QUEUE_ORDER = [6, 2, 0, 3, 1, 4, 5]     # solver_socp.py:644
KKT_STOP, KKT_PRIM, KKT_DUAL = [0, 2, 4, 5], [0, 1], [2, 3]   # :299-301
PRIMAL = ("phi", "A", "B", "lambda_c")
Z_VARS = ("z_fst", "z_mid", "z_end")
DUAL_QE = ("mu", "E")
BETAS = ("beta_fst", "beta_mid", "beta_end")
CARRY_MIN_NODES = 0       # space-time nodes from which quiet iterations carry the next iteration's gathers (DOTS_CARRY_MIN)
DEFAULT_CG_TOL = 1e-8     # parity study: profiles/studies/cg_tol_parity.txt (cost within 1e-9 of the reference, budget 1e-6)


def _validate_checkpoints(tol_checkpoints, tol):
    """solver_socp.py:85-94."""
    if tol_checkpoints is None:
        return None
    if not isinstance(tol_checkpoints, list) or not tol_checkpoints:
        raise ValueError("tol_checkpoints must be a non-empty list")
    for i, cp in enumerate(tol_checkpoints):
        if not (isinstance(cp, (int, float)) and 0 < cp < 1):
            raise ValueError(f"Invalid checkpoint value at index {i}: {cp}. Must be between 0 and 1")
        if cp < tol:
            raise ValueError(f"Checkpoint value must be greater than tol. However, checkpoint ({cp}) < tol ({tol})")
    return sorted(tol_checkpoints, reverse=True)


class AlmSolver:
    """The reference's solver as an object: ``__init__`` = setup (solver_socp.py:96-652),
    ``iterate()`` = one pass of the main loop (:656-823), ``finalize()`` = :826-871.

    The scalars the reference keeps in closure variables (r, prim/dual scale, constant_d,
    scale_factor_z, congestion, norm constants) live here on the host and are pushed to the
    device context whenever they change; arrays live on the device only.
    """

    def __init__(self, n_time, geometry, congestion=0.0, nit=1000, eps=0.0, tol=1e-4, tau=1.90, is_z_scaling=True,
                 is_constant_scaling=False, check_kkt_step_by_step=False, init_solution=None, tol_checkpoints=None,
                 time_limit=1000, is_palm=False, lap_solver="modal_direct", cg_tol=DEFAULT_CG_TOL, cg_max_iter=20000, device=0, reorder=True,
                 preconditioner="multigrid", mg_coarsest=256, time_slab=None, nd_leaf=16):
        self.tol_checkpoints = _validate_checkpoints(tol_checkpoints, tol)
        self.checkpoint_solutions = []
        self.n_time, self.nit, self.tol, self.time_limit = int(n_time), int(nit), tol, time_limit
        self.is_z_scaling, self.is_constant_scaling = is_z_scaling, is_constant_scaling
        self.check_kkt_step_by_step = check_kkt_step_by_step
        self.is_palm = bool(is_palm)      # an extra (q, lambda_c) solve opens every iteration (solver_socp.py:668-672)
        self.direct = direct = lap_solver == "modal_direct"
        self.untimed_steps = 0          # iterations whose phases were not timed (run_history.steps_time is estimated from the others)
        self.quiet_steps = 0            # iterations after which nothing was read back
        self._timed_in_flight = []      # kinds of the timed iterations whose events have not been collected yet (one entry per slot)
        # the right-hand side + cone projection of the iteration after a read-back may be enqueued ahead (iterate(); DOTS_RHS_AHEAD=0
        # never).  Round 2 only did so on small problems (the projection then ran apart from the right-hand side: one launch more);
        # the projection now rides in the launch ahead, its results in alternate buffers until the step takes them: the same launch
        # the iteration would start with, only earlier -- on for every size
        ahead = env_choice("DOTS_RHS_AHEAD", ("0", "1", "2"), "1")
        self._rhs_ahead_ok = (direct and time_slab is None and not self.is_palm and not check_kkt_step_by_step
                              and not is_constant_scaling and ahead != "0")
        self._rhs_ahead = False
        self._carry = False             # DOTS_STEP_CARRY for the next device step (iterate())
        # ... which pays where the iteration is bandwidth-bound: below DOTS_CARRY_MIN space-time nodes (V (T + 1)) the launches are
        # latency-bound and the bytes saved in the right-hand side / projection only balance what the exchange through LDS costs
        # steps 2+3 (A/B of the round: DESIGN.md section 5)
        nodes = int(np.asarray(geometry["vertices"]).shape[0]) * (int(n_time) + 1)
        self._carry_ok = direct and not self.is_palm and nodes >= int(env_choice("DOTS_CARRY_MIN", None, str(CARRY_MIN_NODES), integer=(0, 1 << 40)))
        self._fused_kkt = False         # the last device step formed the KKT sums it holds in registers (DOTS_STEP_KKT_SUMS)
        if direct and reorder is True:
            reorder = "nd"      # the elimination order of the factor doubles as the locality numbering
        self.dev = dev = DeviceProblem(n_time, geometry, lap_solver="modal_pcg" if direct else lap_solver, device=device,
                                       reorder=reorder, time_slab=time_slab, nd_leaf=nd_leaf)

        p = dev.params
        self.r = 1.0
        self.prim_scale = self.dual_scale = self.boundary_scale = 1.0
        self.const_d = self.scale_z = 1.0
        self.norm_d, self.norm_boundary = p.norm_d, p.norm_boundary
        self.congestion0 = float(congestion)
        self.congestion = float(congestion)
        self.tau, self.eps = float(tau), float(eps)
        dev.set_params(cg_tol=cg_tol, cg_max_iter=int(cg_max_iter))
        self._push()
        if preconditioner not in ("multigrid", "jacobi"):
            raise ValueError("preconditioner must be 'multigrid' or 'jacobi'")
        self.mg_summary = self.front_summary = None
        self.lap_solver_fallback = None      # why the direct solve was not used although it was asked for
        if direct:
            # Memory regime: the factor holds ~70 entries per vertex and time mode (8.9 GB at 500k vertices x 32 modes).  When it
            # does not fit beside the state, step 1 runs the batched multigrid-PCG on the same modes instead -- the reference has
            # no counterpart (laplacian_inverse_socp.py:34-41 just factorises); the choice is logged and reported in solver_stats.
            try:
                self.front_summary = dev.setup_frontal(eps=self.eps)
            except _lib.HipLibraryError as exc:
                if exc.status != _lib.ERR_MEMORY or time_slab is not None:
                    raise
                self.lap_solver_fallback = str(exc)
                logger.warning("modal_direct -> modal_pcg with the multigrid preconditioner: %s", exc)
                self.direct = direct = False
                self._rhs_ahead_ok = False
                if preconditioner == "multigrid":
                    self.mg_summary = dev.setup_multigrid(eps=self.eps, coarsest=mg_coarsest)
        elif preconditioner == "multigrid" and lap_solver == "modal_pcg":
            self.mg_summary = dev.setup_multigrid(eps=self.eps, coarsest=mg_coarsest)
        init_solution = init_solution or {}
        self._upload_initial_state(init_solution)

        self.run_history = RunningHistory(max_record_numbers=self.nit, kkt_labels=KKT_LABELS,
                                          kkt_short_labels=KKT_SHORT_LABELS, name="SOCP")
        # per-step timers without a host wait in the loop (control.SampledStepTimers); DOTS_TIME_EVERY=1 times every iteration.
        # The five events of a timed iteration cost ~30 us of stream time at knot (measured on the driver's 20-step window:
        # 8 190 it/s without sampling, 7 870 with the first 4 + every 8th of a kind, profiles/tools/driver_window.py): the first 2
        # of a kind and every 32nd are sampled (< 1 % of a 95-us iteration)
        self.step_timers = SampledStepTimers(self.run_history, first=2, every=int(env_choice("DOTS_TIME_EVERY", None, "32", integer=(1, 1 << 20))))
        self.adjust_params = AdjustAdmmParam()
        self.is_org_kkt = False
        self.cg_total = self.cg_fail = 0
        self.counter_main = -1
        self.finished = False

        self.run_history.start()
        self.prim_gap = 1.0 + 1.0 * math.exp(-100 * congestion)      # :568
        if is_z_scaling:
            self.scale_variable_z(2.0, msg="Initially scale z")          # :571-572
        if is_constant_scaling:
            self._initial_constant_scaling(init_solution)                # :574-586

        self._kkt_cache = {}
        conditions = [ErrorCondition((lambda i=i: self._kkt_value(i)), tol, KKT_SHORT_LABELS[i]) for i in range(7)]
        self.kkt_validator = AdaptiveValidator(ConditionValidator(conditions, KKT_QUEUE_ORDER))   # :589-645
        self.start_time = time.perf_counter()

    # ---- parameters ------------------------------------------------------------------------
    def _push(self):
        self.dev.set_params(r=self.r, scale_z=self.scale_z, const_d=self.const_d, norm_d=self.norm_d,
                            norm_boundary=self.norm_boundary, congestion=self.congestion, tau=self.tau, eps=self.eps,
                            prim_scale=self.prim_scale, dual_scale=self.dual_scale, boundary_scale=self.boundary_scale)

    def _upload_initial_state(self, init):
        """solver_socp.py:239-250: missing entries default to zero / to the derived expressions."""
        dev, r = self.dev, self.r
        if not init:
            return   # device state is zero-initialised: phi = 0 -> A = B = 0, all multipliers 0
        unknown = set(init) - set(STATE_NAMES) - {"checkpoints"}
        if unknown:
            raise ValueError(f"unknown init_solution entries: {sorted(unknown)}")
        have = {k: np.asarray(v, dtype=np.float64) for k, v in init.items() if k in STATE_NAMES and v is not None}
        phi = have.get("phi", np.zeros(dev.shape("phi")))
        dev.upload("phi", phi)
        dev.upload("A", have["A"] if "A" in have else dev.apply_operator("grad_time", phi))
        dev.upload("B", have["B"] if "B" in have else dev.apply_operator("grad_space", phi))
        for k in ("lambda_c", "z_fst", "z_end", "z_mid"):
            if k in have:
                dev.upload(k, have[k])
        betas = {k: (1.0 / r) * have.get(k, np.zeros(dev.shape(k))) for k in BETAS}
        for k in BETAS:
            dev.upload(k, betas[k])
        mu = (1.0 / r) * (have["mu"] if "mu" in have else r * (betas["beta_fst"] - betas["beta_end"]))
        dev.upload("mu", mu)
        if "E" in have:
            dev.upload("E", (1.0 / r) * have["E"])
        else:
            dev.upload("E", -dev.apply_operator("decouple_adjoint", betas["beta_mid"], 1.0))

    # ---- scaling tools (solver_socp.py:324-412) -------------------------------------------
    def adjust_penalty(self, factor):
        self._kkt_cache = {}
        self.r *= factor
        self.dev.adjust_penalty(factor)
        self._push()

    def scale_variable_z(self, scale_factor, msg="Scale z"):
        logger.log(12, "%s with z factor: %s", msg, scale_factor)
        self._kkt_cache = {}
        self.scale_z *= scale_factor
        self.const_d *= scale_factor
        self.norm_d *= scale_factor
        # the reference multiplies by the *cumulative* factor (:383-384)
        self.dev.scale_z(self.scale_z, 1.0 / self.scale_z, self.scale_z)
        self._push()

    def scale_prim_dual(self, scale_factor=None):
        self._kkt_cache = {}
        dev = self.dev
        if scale_factor is None:
            names = [("phi", 1), ("phi", 2), ("A", 0), ("B", 0), ("z_fst", 0), ("z_mid", 0), ("z_end", 0), ("mu", 0), ("E", 0),
                     ("beta_fst", 0), ("beta_mid", 0), ("beta_end", 0)]
            n2 = dict(zip(names, self._norm_squares(names)))
            prim = [
                math.sqrt(n2["phi", 1] + n2["phi", 2]),
                math.sqrt(n2["A", 0] + n2["B", 0]),
                math.sqrt(n2["z_fst", 0] + n2["z_mid", 0] + n2["z_end", 0]),
            ]
            dual = [
                self.r * math.sqrt(n2["mu", 0] + n2["E", 0]),
                self.r * math.sqrt(n2["beta_fst", 0] + n2["beta_mid", 0] + n2["beta_end", 0]),
            ]
            prim_rescale, dual_rescale = AdjustAdmmParam.compute_scale_factor(prim, dual)
        else:
            prim_rescale, dual_rescale = scale_factor
        if max(prim_rescale, dual_rescale) / min(prim_rescale, dual_rescale) > 2.0:
            self.prim_scale *= prim_rescale
            self.dual_scale *= dual_rescale
            dev.scale_arrays(PRIMAL + Z_VARS, 1.0 / prim_rescale)
            dual_factor = dual_rescale ** 2 / prim_rescale
            dev.scale_arrays(DUAL_QE + BETAS, 1.0 / dual_factor)
            # the boundary term is divided by dual_factor while r is multiplied by dual/prim; the device
            # rebuilds the boundary from r, so only the remaining 1/dual goes into boundary_scale
            self.boundary_scale /= dual_rescale
            self.r *= dual_rescale / prim_rescale
            self.congestion *= dual_rescale / prim_rescale
            self.const_d /= prim_rescale
            self.norm_d /= prim_rescale
            self.norm_boundary /= dual_rescale
            self._push()

    def _initial_constant_scaling(self, init_solution):
        dev, n_time = self.dev, self.n_time
        h = 1.0 / n_time
        plan = dev.plan
        inv = np.arange(dev.V)
        if plan.perm_vert is not None:
            inv = np.empty(dev.V, dtype=np.int64)
            inv[plan.perm_vert] = np.arange(dev.V)
        mass, mu0, mu1 = plan.mass_vert[inv], plan.mu0[inv], plan.mu1[inv]
        bt = np.zeros((n_time + 1, dev.V))        # r * boundary / mass at r = 1
        bt[0], bt[-1] = -mu0 / (h * mass), mu1 / (h * mass)
        norm_c = math.sqrt(float(np.sum(bt ** 2 * mass[None, :])) / (n_time + 1))
        norm_ac = self._boundary_gradient_norm(bt, init_solution.get("phi"))
        self.scale_prim_dual(scale_factor=(self.norm_d, math.sqrt(n_time) * norm_c ** 2 / norm_ac))
        self.adjust_penalty(1.0 / self.r)

    def _boundary_gradient_norm(self, bt, phi0):
        """sqrt(|d_t b|^2 + |d_x b|^2) of the boundary term (:574-586): the device's own operators on an uploaded copy."""
        dev = self.dev
        dev.upload("phi", bt)
        norm_ac = math.sqrt(dev.norm_square("phi", 1) + dev.norm_square("phi", 2))
        dev.upload("phi", np.zeros_like(bt) if phi0 is None else np.asarray(phi0, dtype=np.float64))
        return norm_ac

    def _norm_squares(self, requests):
        """norm_square_* (:875-878) of the listed (array, part) pairs; the multi-GPU solver adds the slabs' shares."""
        return [self.dev.norm_square(name, part) for name, part in requests]

    def recovered(self, name, arr):
        """recorver_scaled_solution (:397-405)."""
        if name in PRIMAL:
            return self.prim_scale * arr
        if name in Z_VARS:
            return (self.prim_scale / self.scale_z) * arr
        if name in DUAL_QE:
            return (self.r * self.dual_scale) * arr
        return (self.r * self.scale_z * self.dual_scale) * arr

    # ---- KKT residuals of the current iterate, evaluated at most once each: the conditions an iteration is known to need
    # (the four primal / dual ones before a penalty update, all seven in step-by-step mode and at the end) are fetched in
    # ONE device call (one pass of the KKT kernels, one host round trip) instead of one call per condition
    # After a step with DOTS_STEP_KKT_SUMS the conditions FUSED_KKT cost no pass over the state (their sums were formed by steps
    # 2+3) and Dual(alpha) one vertex pass: whatever of them the lazy validator may ask for next (its queue runs 6, 2, 0, 3, 1)
    # comes with the first round trip.  Which values the validator LOOKS at -- and so the NaN pattern of the history -- is
    # unchanged: the cache only holds more than was asked for.  Conditions 4 and 5 gather over corner lists and keep their own pass.
    FUSED_KKT = (0, 1, 3, 6)

    def _widen(self, conditions):
        want = set(conditions)
        if not self._fused_kkt or want & {4, 5}:
            return list(conditions)
        if want & {2, 6}:      # (6 passes whenever there is no congestion, and 2 follows it in the queue)
            want.add(2)
        want.update(self.FUSED_KKT)
        return sorted(want - set(self._kkt_cache))

    def _kkt_value(self, i):
        if i not in self._kkt_cache:
            self._kkt_cache.update(self._kkt(self._widen([i])))
        return self._kkt_cache[i]

    def _kkt_prefetch(self, conditions):
        missing = [i for i in conditions if i not in self._kkt_cache]
        if missing:
            self._kkt_cache.update(self._kkt(self._widen(missing)))

    # ---- what the multi-GPU solver overrides: everything that reads numbers or arrays back from the device(s)
    def _kkt(self, conditions):
        return self.dev.kkt(conditions)

    def _objective(self):
        return self.dev.objective()

    def _download(self, name):
        """The whole array ``name`` in the reference layout."""
        return self.dev.download(name)

    def _device_step(self, quiet=False):
        """Steps 1-3 on the device; the multi-GPU solver overrides this with begin / all-gather / end.

        ``quiet``: nothing is read back after this iteration (no KKT evaluation, not the last one): z_mid
        need not be stored, and with the direct solver the host does not wait for the device either."""
        # is_palm's step 0 reads z_mid of the previous iteration: it is stored every iteration then
        # With the direct solver an iteration needs no host round trip: it is only enqueued, and on iterations that read
        # back the KKT kernels follow it on the stream (one wait, at the read-back).  The phase timers of the history
        # (Step 1-1 ...) come from SAMPLED iterations whose phases are bracketed by events on the stream, collected at the
        # next read-back and scaled to all iterations of their kind (control.SampledStepTimers).
        kind = "quiet" if quiet else "read-back"
        sample = self.step_timers.begin(kind)
        if quiet:
            self.quiet_steps += 1
        if self.direct:
            timed = sample and len(self._timed_in_flight) < 60       # (the ring of the library holds 64 slots)
            self.dev.step_flags(skip_z_mid=quiet and not self.is_palm, palm=self.is_palm, rhs_ahead=self._rhs_ahead and not quiet, timed=timed,
                                carry=self._carry, kkt_sums=not quiet)
            self._fused_kkt = not quiet
            try:
                self.dev.step(1, wait=False)
            except Exception:
                # the library gave the timing slot of a failed step back: what this side still expects of the ring can no longer
                # be matched to kinds -- drop it, so that the error that surfaces is the step's own
                self._timed_in_flight.clear()
                raise
            if timed:
                self._timed_in_flight.append(kind)
            else:
                self.untimed_steps += 1
        else:       # the PCG waits for its convergence flags anyway: every iteration is timed
            self.dev.step_flags(skip_z_mid=quiet and not self.is_palm, palm=self.is_palm)
            self._account(self.dev.step(1), kind)

    def _collect_step_times(self, wait=False):
        """Phase times of the timed iterations that have finished (at a read-back: all of them), into the history."""
        fresh = False
        if self._timed_in_flight:
            for st in self.dev.step_times(wait=wait):
                self._account(st, self._timed_in_flight.pop(0))
                fresh = True
        if fresh or wait:       # (the estimate also scales with the iteration counts: brought up to date whenever it is asked for with wait=True)
            self.step_timers.publish()

    def _time_is_up(self, reads_back=True):
        """``reads_back``: this iteration synchronises with the host anyway (the multi-GPU driver only shares the
        clock decision on those iterations)."""
        return (time.perf_counter() - self.start_time) > self.time_limit

    def _account(self, st, kind):
        """One dots_step_stats (a whole iteration, or one stage of a time slab: the iteration is complete with the record that
        carries alm_iterations = 1) of an iteration of ``kind`` into the sampled timers."""
        tm = self.step_timers
        self.cg_total += st.cg_iterations
        self.cg_fail += st.cg_not_converged
        done = int(st.alm_iterations)
        tm.add(kind, "Step 1-1 (Laplacian)", 1e-3 * (st.ms_rhs + st.ms_laplacian), done)
        tm.add(kind, "Step 1-2 (SOC-Projection)", 1e-3 * st.ms_soc, done)
        tm.add(kind, "Step 2+3 (Q & Lambda, Multiplier)", 1e-3 * st.ms_q_lambda_multiplier, done)

    # ---- one pass of the main loop (:656-823); returns True when the loop must stop ------------
    def iterate(self):
        if self.finished:
            return True
        self.counter_main += 1
        it, dev, hist, params = self.counter_main, self.dev, self.run_history, self.adjust_params
        if self.is_constant_scaling and params.is_to_scale(it):
            self.scale_prim_dual()
        if self.is_z_scaling and params.is_to_scale_matrix(it, hist.get_current_kkt_errors()):
            rescale_z = safe_rescale_ratio(self.prim_gap, hist.get_current_kkt_errors())
            if rescale_z > 1.25:
                self.scale_variable_z(rescale_z, msg=f"Rescale z at iteration {it}")

        # The reference looks at the clock after the step (:725); here before it, so that it is known in advance whether
        # this iteration's results are read back (a wall-clock limit has no parity to keep).
        validator = self.kkt_validator
        reads_back = (self.check_kkt_step_by_step or it + 1 >= self.nit or params.peek_adjust(it) or validator.will_validate_next()
                      or (self.is_constant_scaling and params.is_to_scale(it + 1)))   # the next iteration opens with norms of z
        is_time_used_up = self._time_is_up(reads_back)
        quiet = not (is_time_used_up or reads_back)
        self._kkt_cache = {}
        # An iteration that reads residuals back but changes nothing afterwards (no penalty update: that is known from the
        # schedule) is followed by an iteration that starts from the state it leaves, unless the run stops or z is rescaled:
        # the device starts on that iteration's right-hand side while the host waits for the residuals (DOTS_STEP_RHS_AHEAD;
        # dropped by the library if anything changes in between).
        self._rhs_ahead = (self._rhs_ahead_ok and reads_back and not is_time_used_up and it + 1 < self.nit
                           and not params.peek_adjust(it))
        # The next iteration starts from the state this one leaves unless the penalty is updated in between (known from the schedule;
        # a z rescaling or a stop simply drop what was carried): steps 2+3 then also store the per-corner sums that iteration's
        # right-hand side and cone projection would gather from B, E and beta_mid (DOTS_STEP_CARRY: one pass over beta_mid less).
        self._carry = self._carry_ok and it + 1 < self.nit and not params.peek_adjust(it)
        self._device_step(quiet)                                                # steps 1-3 (:674-722)

        adjust = params.is_to_adjust(it) or is_time_used_up
        required = KKT_PRIM + KKT_DUAL if adjust else None
        validator = self.kkt_validator

        history = None
        if not self.check_kkt_step_by_step:
            if adjust:
                validator.reset_counter()
                if self._rhs_ahead_ok and not is_time_used_up and it + 1 < self.nit:
                    # the library takes the decision below itself as soon as the residuals have arrived and starts the next iteration's
                    # first launch with it, while this side is still on its way there (dots_penalty_ahead; confirmed -- or dropped -- by the
                    # adjust_penalty / set_params calls further down: results never depend on it)
                    self.dev.penalty_ahead(self.tol, self.is_org_kkt, params._sigma_lower_bound, params._sigma_upper_bound, params.FACTOR_STEPS)
                self._kkt_prefetch(required)
            passed, _info = validator.validate(required)
            org, scaled = validator.collect()
            if adjust:
                validator.reset_counter()
        else:
            self._kkt_prefetch(range(7))
            passed, _info = validator.validator.validate(list(range(7)))
            org, scaled = validator.collect()
            cost, lagr = self._objective()
            history = {"Transportation cost": cost, "Objective value": lagr}
        error = max_of_list_with_none([org[i] for i in KKT_STOP])

        cps = self.tol_checkpoints
        if cps and error is not None and error <= cps[0]:                      # :790-801
            self.checkpoint_solutions.append({
                "mu": (self.r * self.dual_scale) * self._download("mu"),
                "E": (self.r * self.dual_scale) * self._download("E"),
                "iteration": it, "time": hist.get_running_time(), "kkt": np.array(org, dtype=object),
            })
            cps.pop(0)

        stop = passed or is_time_used_up
        if stop or it + 1 >= self.nit:
            self.finished = True
        if not stop:
            max_scaled = max_of_list_with_none(scaled)
            if max_scaled is not None and max_scaled < 5 * self.tol:
                self.is_org_kkt = True
            if adjust:                                                         # :813-823
                # enqueued BEFORE the records below are written (the reference records first, :776-789): nothing the records
                # hold depends on it, and the device divides its arrays while the host does its bookkeeping
                src = org if self.is_org_kkt else scaled
                prim_error = max_of_list_with_none([src[i] for i in KKT_PRIM])
                dual_error = max_of_list_with_none([src[i] for i in KKT_DUAL])
                self.adjust_penalty(params.get_updated_value(self.r, prim_error / dual_error) / self.r)
        hist.record(current_it=it, kkt_errors=org, history=history)
        if not quiet:
            self._collect_step_times()      # (the read-back above waited for the stream: every timed iteration so far is complete)
        if not self.check_kkt_step_by_step and error is not None:
            validator.set_error_and_tolerance(error, self.tol)
        return self.finished

    # ---- final record and solution (:826-871) ------------------------------------------------
    def finalize(self, download=True):
        dev, hist, validator = self.dev, self.run_history, self.kkt_validator
        self._kkt_prefetch(range(7))
        validator.validator.validate(list(range(7)))
        org, _ = validator.collect()
        cost, lagr = self._objective()
        hist.record(current_it=self.counter_main, kkt_errors=org,
                    history={"Transportation cost": cost, "Objective value": lagr})
        self._collect_step_times(wait=True)
        hist.end()
        hist.solver_stats = {
            "cg_iterations": int(self.cg_total), "cg_not_converged": int(self.cg_fail), "lap_solver": dev.lap_solver,
            "device_bytes": dev.device_bytes(), "final_r": self.r, "final_scale_z": self.scale_z,
        }
        if self.lap_solver_fallback:
            hist.solver_stats.update(lap_solver="modal_pcg (asked for modal_direct)", lap_solver_fallback=self.lap_solver_fallback)
        if self.cg_fail:
            logger.warning("PCG hit its iteration cap in %d solves", self.cg_fail)
        solution = {}
        if download:
            solution = {name: self.recovered(name, self._download(name)) for name in STATE_NAMES}
        solution["checkpoints"] = self.checkpoint_solutions if self.checkpoint_solutions else None
        logger.info("Number of iterations: %d   Iteration time: %.2f", self.counter_main, hist.running_time)
        return solution, hist

    def close(self):
        self.dev.close()


def solver_socp(
        n_time,
        geometry,
        congestion=0.0,
        nit=1000,
        eps=0.0,
        tol=1e-4,
        tau=1.90,
        is_palm=False,
        is_multi_threads=True,
        is_z_scaling=True,
        is_constant_scaling=False,
        check_kkt_step_by_step=False,
        init_solution=None,
        tol_checkpoints=None,
        time_limit=1000,
        *,
        lap_solver="modal_direct",
        cg_tol=DEFAULT_CG_TOL,
        cg_max_iter=20000,
        device=0,
        reorder=True,
        preconditioner="multigrid",
        mg_coarsest=256,
        nd_leaf=16,
):
    """SOCP for dynamical optimal transport on a discrete surface, on the GPU.

    Returns ``(solution, run_history)``: ``solution`` is a dict with the twelve arrays of
    ``SolutionSocpData`` (reference layouts, un-scaled as at solver_socp.py:397-405) plus
    ``checkpoints``; ``run_history`` is a :class:`RunningHistory`.
    ``is_multi_threads`` is accepted and ignored (the GPU path has no host threads to split).
    """
    alm = AlmSolver(n_time, geometry, congestion=congestion, nit=nit, eps=eps, tol=tol, tau=tau, is_z_scaling=is_z_scaling,
                    is_constant_scaling=is_constant_scaling, check_kkt_step_by_step=check_kkt_step_by_step,
                    init_solution=init_solution, tol_checkpoints=tol_checkpoints, time_limit=time_limit, is_palm=is_palm,
                    lap_solver=lap_solver, cg_tol=cg_tol, cg_max_iter=cg_max_iter, device=device, reorder=reorder,
                    preconditioner=preconditioner, mg_coarsest=mg_coarsest, nd_leaf=nd_leaf)
    try:
        for _ in range(nit):
            if alm.iterate():
                break
        return alm.finalize()
    finally:
        alm.close()
