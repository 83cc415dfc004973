"""Record the headline runs of BASELINE.json's configurations (build container only; takes tens of minutes).

    python tests/golden/make_headline.py reference sphere10k knot knot63     # the REFERENCE itself (via ref_shim)
    python tests/golden/make_headline.py oracle torus100k                    # the CPU oracle (the reference would
                                                                              # need hours at this size)
    python tests/golden/make_headline.py reference torus100k_ref10 torus65k_T127   # the reference, first 10 iterations
    python tests/golden/make_headline.py --verify reference knot knot63      # regenerate into a temp dir, compare bit for bit

Geometry comes from dots_socp_amd/meshes.py (deterministic generators), so a fixture stores
only a checksum of it, the stopping iteration, the cost / objective / KKT histories and a sample of the solution
(every 40th vertex of mu, plus per-layer sums and norms): ``headline_<workload>.npz``."""
from __future__ import annotations

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from dots_socp_amd import meshes  # noqa: E402

WORKLOADS = {   # same table as bench.py
    "sphere10k": dict(example="sphere", kw=dict(level=5), n_time=31, congestion=0.0, tol=1e-3),
    "knot": dict(example="knot", kw={}, n_time=31, congestion=0.0, tol=1e-3),
    "knot63": dict(example="knot", kw={}, n_time=63, congestion=0.1, tol=1e-3),
    "torus100k": dict(example="torus", kw=dict(nu=400, nv=250), n_time=31, congestion=0.0, tol=1e-3),
    # BASELINE configs[3] (the north star's size) recorded from the REFERENCE itself, truncated like configs[4] below: the full
    # 282 iterations would take the reference many hours; `headline_torus100k.npz` (the oracle's full-length run) stays beside it
    "torus100k_ref10": dict(example="torus", kw=dict(nu=400, nv=250), n_time=31, congestion=0.0, tol=1e-3, nit=10, check_kkt_step_by_step=True),
    # BASELINE configs[4] stand-in at its full size; the reference needs ~1 min per iteration there, so the run is
    # TRUNCATED: the first `nit` iterations with every KKT residual and the objective recorded each iteration
    "torus65k_T127": dict(example="torus", kw=dict(nu=360, nv=180), n_time=127, congestion=0.0, tol=1e-5, nit=10, check_kkt_step_by_step=True),
}


def same_fixture(a, b):
    """Two fixture files hold the same recorded run: every array bit for bit, the wall-clock `seconds` aside."""
    if set(a.files) != set(b.files):
        return False, f"keys differ: {sorted(set(a.files) ^ set(b.files))}"
    for k in a.files:
        if k == "seconds":
            continue
        if a[k].shape != b[k].shape or not np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"):
            return False, f"{k} differs"
    return True, "identical"


def main(argv):
    # --verify: record into a temporary directory and compare with the committed fixtures instead of overwriting them
    #   python tests/golden/make_headline.py --verify reference knot knot63      (exit code 1 if any array differs)
    verify = "--verify" in argv
    argv = [a for a in argv if a != "--verify"]
    who, names = argv[0], argv[1:]
    out_dir = HERE
    if verify:
        import tempfile

        out_dir = tempfile.mkdtemp(prefix="headline_verify_")
    if who == "reference":
        import ref_shim
        from make_golden import run_reference

        ref = ref_shim.load_reference()
        solve = lambda T, g, **kw: run_reference(ref, g, T, **kw)        # noqa: E731
    else:
        import importlib.util

        spec = importlib.util.spec_from_file_location("dots_oracle", os.path.join(ROOT, "oracle", "dots_oracle.py"))
        O = importlib.util.module_from_spec(spec)
        sys.modules["dots_oracle"] = O
        spec.loader.exec_module(O)
        solve = lambda T, g, **kw: O.solver_socp(T, g, **kw)              # noqa: E731
    failed = False
    for name in names:
        wl = WORKLOADS[name]
        geom, scale = meshes.example(wl["example"], **wl["kw"])
        t0 = time.time()
        extra = {k: wl[k] for k in ("check_kkt_step_by_step",) if k in wl}
        sol, hist = solve(wl["n_time"], geom, nit=wl.get("nit", 20000), tol=wl["tol"], congestion=wl["congestion"], time_limit=1e9, **extra)
        sec = time.time() - t0
        mu = np.asarray(sol["mu"])
        out = dict(
            source=np.array(who), n_time=np.array(wl["n_time"]), tol=np.array(wl["tol"]), congestion=np.array(wl["congestion"]),
            scale_factor=np.array(scale), seconds=np.array(sec), nit=np.array(wl.get("nit", 20000)),
            check_kkt_step_by_step=np.array(bool(wl.get("check_kkt_step_by_step", False))),
            vertices_checksum=np.array([geom["vertices"].sum(), np.abs(geom["vertices"]).sum(), float(geom["triangles"].sum())]),
            mu0_checksum=np.array([np.dot(geom["mu0"], np.arange(geom["mu0"].size)), np.dot(geom["mu1"], np.arange(geom["mu1"].size))]),
            last_iteration=np.array(int(hist.kkt_iteration[-1])),
            hist_kkt_errors=np.asarray(hist.kkt_errors, dtype=np.float64), hist_kkt_iteration=np.asarray(hist.kkt_iteration, dtype=np.float64),
            mu_sample=mu[:, ::40].copy(), mu_layer_sum=mu.sum(axis=1), mu_layer_norm=np.sqrt((mu * mu).sum(axis=1)),
            E_norm=np.array(np.sqrt((np.asarray(sol["E"]) ** 2).sum())), A_norm=np.array(np.sqrt((np.asarray(sol["A"]) ** 2).sum())),
        )
        for k, val in hist.history.items():
            out["hist_" + k.replace(" ", "_")] = np.asarray(val, dtype=np.float64)
        np.savez_compressed(os.path.join(out_dir, f"headline_{name}.npz"), **out)
        print("wrote", os.path.join(out_dir, f"headline_{name}.npz") if verify else f"headline_{name}.npz", who, "last iteration", out["last_iteration"],
              "cost", hist.history["Transportation cost"][-1], "seconds", round(sec, 1), flush=True)
        if verify:
            ok, why = same_fixture(np.load(os.path.join(out_dir, f"headline_{name}.npz")), np.load(os.path.join(HERE, f"headline_{name}.npz")))
            print("verify", name, "OK" if ok else "MISMATCH", why, flush=True)
            failed = failed or not ok
    if verify and failed:
        raise SystemExit(1)


if __name__ == "__main__":
    main(sys.argv[1:])
