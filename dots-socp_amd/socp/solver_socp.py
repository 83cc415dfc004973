"""``solver_socp``: the reference's ALM driver with the numerics on the GPU.

Same signature, return contract and iteration-by-iteration decisions as
``dot_surface_socp/socp/solver_socp.py:25-871``; every array operation of the loop
(steps 1-3, KKT residuals, objective, scaling) runs in HIP kernels behind the C ABI
(``include/dots_socp_hip.h``), the state never leaves HBM between iterations and the host
only sees scalars.  Extra keyword arguments (all optional) select device-side choices:

    lap_solver   "modal_pcg" (default) or "spacetime_pcg"  -- how step 1's Laplacian is solved
    cg_tol       relative PCG tolerance (default 1e-10; see DESIGN.md for the parity budget)
    cg_max_iter  PCG iteration cap
    device       HIP device ordinal
    reorder      locality renumbering of the mesh (default True)

There is no CPU path: without libdotsocp_hip.so and a GPU this raises.
"""
from __future__ import annotations

import logging
import math
import time

import numpy as np

from .. import _lib
from ..control import (AdaptiveValidator, AdjustAdmmParam, ConditionValidator, ErrorCondition, RunningHistory,
                       KKT_LABELS, KKT_SHORT_LABELS, max_of_list_with_none, safe_rescale_ratio)
from ..device import DeviceProblem, STATE_NAMES

logger = logging.getLogger("dots_socp_amd")

KKT_QUEUE_ORDER = [6, 2, 0, 3, 1, 4, 5]     # solver_socp.py:644
KKT_STOP, KKT_PRIM, KKT_DUAL = [0, 2, 4, 5], [0, 1], [2, 3]   # :299-301
PRIMAL = ("phi", "A", "B", "lambda_c")
Z_VARS = ("z_fst", "z_mid", "z_end")
DUAL_QE = ("mu", "E")
BETAS = ("beta_fst", "beta_mid", "beta_end")


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


class _Scalars:
    """Host copies of the scalars the reference keeps as closure variables (:97, :318-321)."""

    def __init__(self, dev: DeviceProblem, congestion, tau, eps):
        p = dev.params
        self.r = 1.0
        self.prim_scale = self.dual_scale = 1.0
        self.boundary_scale = 1.0
        self.const_d = 1.0
        self.scale_z = 1.0
        self.norm_d = p.norm_d
        self.norm_boundary = p.norm_boundary
        self.congestion = float(congestion)
        self.tau, self.eps = float(tau), float(eps)

    def push(self, dev: DeviceProblem):
        dev.set_params(r=self.r, scale_z=self.scale_z, const_d=self.const_d, norm_d=self.norm_d,
                       norm_boundary=self.norm_boundary, congestion=self.congestion, tau=self.tau, eps=self.eps,
                       prim_scale=self.prim_scale, dual_scale=self.dual_scale, boundary_scale=self.boundary_scale)


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
        lap_solver="modal_pcg",
        cg_tol=1e-10,
        cg_max_iter=20000,
        device=0,
        reorder=True,
):
    """SOCP for dynamical optimal transport on a discrete surface, on the GPU.

    Returns ``(solution, run_history)``: ``solution`` is a dict with the twelve arrays of
    ``SolutionSocpData`` (reference layouts, un-scaled as at solver_socp.py:397-405) plus
    ``checkpoints``; ``run_history`` is a :class:`RunningHistory`.
    ``is_multi_threads`` is accepted and ignored (the GPU path has no host threads to split).
    """
    if is_palm:
        raise NotImplementedError("is_palm=True (the plain-ALM variant) is outside the scoped hot path")
    tol_checkpoints = _validate_checkpoints(tol_checkpoints, tol)
    checkpoint_solutions = []

    dev = DeviceProblem(n_time, geometry, lap_solver=lap_solver, device=device, reorder=reorder)
    try:
        return _run(dev, n_time, congestion, nit, eps, tol, tau, is_z_scaling, is_constant_scaling, check_kkt_step_by_step,
                    init_solution or {}, tol_checkpoints, checkpoint_solutions, time_limit, cg_tol, cg_max_iter)
    finally:
        dev.close()


def _upload_initial_state(dev, init, r):
    """solver_socp.py:239-250: missing entries default to zero / to the derived expressions."""
    if not init:
        return   # device state is zero-initialised: phi=0 -> A=B=0, all multipliers 0
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


def _run(dev, n_time, congestion, nit, eps, tol, tau, is_z_scaling, is_constant_scaling, check_kkt_step_by_step,
         init_solution, tol_checkpoints, checkpoint_solutions, time_limit, cg_tol, cg_max_iter):
    sc = _Scalars(dev, congestion, tau, eps)
    dev.set_params(cg_tol=cg_tol, cg_max_iter=int(cg_max_iter))
    sc.push(dev)
    _upload_initial_state(dev, init_solution, sc.r)

    run_history = RunningHistory(max_record_numbers=nit, kkt_labels=KKT_LABELS, kkt_short_labels=KKT_SHORT_LABELS, name="SOCP")
    adjust_params = AdjustAdmmParam()
    is_org_kkt = False
    cg_total = cg_fail = 0

    # ---- scaling tools (solver_socp.py:324-412) -------------------------------------------
    def adjust_penalty(factor):
        sc.r *= factor
        dev.adjust_penalty(factor)
        sc.push(dev)

    def scale_variable_z(scale_factor, msg="Scale z"):
        logger.log(12, "%s with z factor: %s", msg, scale_factor)
        sc.scale_z *= scale_factor
        sc.const_d *= scale_factor
        sc.norm_d *= scale_factor
        # the reference multiplies by the *cumulative* factor (:383-384)
        dev.scale_z(sc.scale_z, 1.0 / sc.scale_z, sc.scale_z)
        sc.push(dev)

    def scale_prim_dual(scale_factor=None):
        if scale_factor is None:
            nt, ns, nd = dev.norm_square, dev.norm_square, dev.norm_square
            prim = [
                math.sqrt(nt("phi", 1) + ns("phi", 2)),
                math.sqrt(nt("A") + ns("B")),
                math.sqrt(nt("z_fst") + nd("z_mid") + nt("z_end")),
            ]
            dual = [
                sc.r * math.sqrt(nt("mu") + ns("E")),
                sc.r * math.sqrt(nt("beta_fst") + nd("beta_mid") + nt("beta_end")),
            ]
            prim_rescale, dual_rescale = AdjustAdmmParam.compute_scale_factor(prim, dual)
        else:
            prim_rescale, dual_rescale = scale_factor
        if max(prim_rescale, dual_rescale) / min(prim_rescale, dual_rescale) > 2.0:
            sc.prim_scale *= prim_rescale
            sc.dual_scale *= dual_rescale
            dev.scale_arrays(PRIMAL + Z_VARS, 1.0 / prim_rescale)
            dual_factor = dual_rescale ** 2 / prim_rescale
            dev.scale_arrays(DUAL_QE + BETAS, 1.0 / dual_factor)
            # boundary /= dual_factor while r *= dual/prim: the device rebuilds the boundary from r,
            # so only the extra 1/dual goes into boundary_scale
            sc.boundary_scale /= dual_rescale
            sc.r *= dual_rescale / prim_rescale
            sc.congestion *= dual_rescale / prim_rescale
            sc.const_d /= prim_rescale
            sc.norm_d /= prim_rescale
            sc.norm_boundary /= dual_rescale
            sc.push(dev)

    def recovered(name, arr):
        """recorver_scaled_solution (:397-405)."""
        if name in PRIMAL:
            return sc.prim_scale * arr
        if name in Z_VARS:
            return (sc.prim_scale / sc.scale_z) * arr
        if name in DUAL_QE:
            return (sc.r * sc.dual_scale) * arr
        return (sc.r * sc.scale_z * sc.dual_scale) * arr

    # ---- main computation ------------------------------------------------------------------
    run_history.start()
    counter_main = -1
    prim_gap = 1.0 + 1.0 * math.exp(-100 * congestion)

    if is_z_scaling:
        scale_variable_z(2.0, msg="Initially scale z")

    if is_constant_scaling:   # :574-586
        # r * boundary / mass is (-mu0, +mu1) / (h mass) at the two end nodes (boundary_scale = 1 here)
        h = 1.0 / n_time
        plan = dev.plan
        bt = np.zeros((n_time + 1, dev.V))
        inv = np.empty(dev.V, dtype=np.int64)
        if plan.perm_vert is not None:
            inv[plan.perm_vert] = np.arange(dev.V)
        else:
            inv = np.arange(dev.V)
        mass = plan.mass_vert[inv]
        bt[0] = -plan.mu0[inv] / (h * mass)
        bt[-1] = plan.mu1[inv] / (h * mass)
        norm_c = math.sqrt(float(np.sum(bt ** 2 * mass[None, :])) / (n_time + 1))
        dev.upload("phi", bt)
        norm_ac = math.sqrt(dev.norm_square("phi", 1) + dev.norm_square("phi", 2))
        dev.upload("phi", np.zeros_like(bt) if "phi" not in init_solution else np.asarray(init_solution["phi"], dtype=np.float64))
        scale_prim_dual(scale_factor=(sc.norm_d, math.sqrt(n_time) * norm_c ** 2 / norm_ac))
        adjust_penalty(1.0 / sc.r)

    conditions = [ErrorCondition((lambda i=i: dev.kkt([i])[i]), tol, KKT_SHORT_LABELS[i]) for i in range(7)]
    kkt_validator = AdaptiveValidator(ConditionValidator(conditions, KKT_QUEUE_ORDER))

    def step_once():
        nonlocal cg_total, cg_fail
        st = dev.step(1)
        cg_total += st.cg_iterations
        cg_fail += st.cg_not_converged
        run_history.add_time("Step 1-1 (Laplacian)", 1e-3 * (st.ms_rhs + st.ms_laplacian))
        run_history.add_time("Step 1-2 (SOC-Projection)", 1e-3 * st.ms_soc)
        run_history.add_time("Step 2+3 (Q & Lambda, Multiplier)", 1e-3 * st.ms_q_lambda_multiplier)

    start_time = time.perf_counter()
    for counter_main in range(nit):
        if is_constant_scaling and adjust_params.is_to_scale(counter_main):
            scale_prim_dual()

        if is_z_scaling and adjust_params.is_to_scale_matrix(counter_main, run_history.get_current_kkt_errors()):
            rescale_z = safe_rescale_ratio(prim_gap, run_history.get_current_kkt_errors())
            if rescale_z > 1.25:
                scale_variable_z(rescale_z, msg=f"Rescale z at iteration {counter_main}")

        step_once()     # steps 1-3 (:674-722)

        is_time_used_up = (time.perf_counter() - start_time) > time_limit
        whether_adjust_sigma = adjust_params.is_to_adjust(counter_main) or is_time_used_up
        required_conditions = KKT_PRIM + KKT_DUAL if whether_adjust_sigma else None

        if not check_kkt_step_by_step:
            if whether_adjust_sigma:
                kkt_validator.reset_counter()
            kkt_validation_passed, _info = kkt_validator.validate(required_conditions)
            org_kkt_errors, kkt_errors = kkt_validator.collect()
            if whether_adjust_sigma:
                kkt_validator.reset_counter()
            run_history.record(current_it=counter_main, kkt_errors=org_kkt_errors)
            error = max_of_list_with_none([org_kkt_errors[i] for i in KKT_STOP])
            if error is not None:
                kkt_validator.set_error_and_tolerance(error, tol)
        else:
            kkt_validation_passed, _info = kkt_validator.validator.validate(list(range(7)))
            org_kkt_errors, kkt_errors = kkt_validator.collect()
            trans_cost, lagrangian = dev.objective()
            run_history.record(current_it=counter_main, kkt_errors=org_kkt_errors,
                               history={"Transportation cost": trans_cost, "Objective value": lagrangian})
            error = max_of_list_with_none([org_kkt_errors[i] for i in KKT_STOP])

        if tol_checkpoints and error is not None and error <= tol_checkpoints[0]:    # :790-801
            checkpoint_solutions.append({
                "mu": (sc.r * sc.dual_scale) * dev.download("mu"),
                "E": (sc.r * sc.dual_scale) * dev.download("E"),
                "iteration": counter_main,
                "time": run_history.get_running_time(),
                "kkt": np.array(org_kkt_errors, dtype=object),
            })
            tol_checkpoints.pop(0)

        if kkt_validation_passed or is_time_used_up:
            break

        max_kkt_errors = max_of_list_with_none(kkt_errors)
        if max_kkt_errors is not None and max_kkt_errors < 5 * tol:
            is_org_kkt = True

        if whether_adjust_sigma:    # :813-823
            src = org_kkt_errors if is_org_kkt else kkt_errors
            prim_error = max_of_list_with_none([src[i] for i in KKT_PRIM])
            dual_error = max_of_list_with_none([src[i] for i in KKT_DUAL])
            r_factor = adjust_params.get_updated_value(sc.r, prim_error / dual_error) / sc.r
            adjust_penalty(r_factor)

    # ---- final record (:826-845) -------------------------------------------------------------
    kkt_validator.validator.validate(list(range(7)))
    org_kkt_errors, _ = kkt_validator.collect()
    trans_cost, lagrangian = dev.objective()
    run_history.record(current_it=counter_main, kkt_errors=org_kkt_errors,
                       history={"Transportation cost": trans_cost, "Objective value": lagrangian})
    run_history.end()
    run_history.solver_stats = {
        "cg_iterations": int(cg_total), "cg_not_converged": int(cg_fail), "lap_solver": dev.lap_solver,
        "device_bytes": dev.device_bytes(), "final_r": sc.r, "final_scale_z": sc.scale_z,
    }
    if cg_fail:
        logger.warning("PCG hit its iteration cap in %d solves", cg_fail)

    solution = {name: recovered(name, dev.download(name)) for name in STATE_NAMES}
    solution["checkpoints"] = checkpoint_solutions if checkpoint_solutions else None
    logger.info("Number of iterations: %d   Iteration time: %.2f", counter_main, run_history.running_time)
    return solution, run_history
