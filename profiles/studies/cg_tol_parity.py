"""How tight must the PCG tolerance be for parity with the reference's recorded runs?
For every golden run and cg_tol: does the solver stop at the reference's iteration, and how far are the
cost / KKT values / mu from the recorded ones.   python profiles/studies/cg_tol_parity.py  (needs a GPU)"""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dots_socp_amd.socp import solver_socp

runs = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "run_*tol1e*.npz")))
print(f"{'run':42s} {'cg_tol':>8s} {'same it':>8s} {'d cost':>10s} {'d kkt':>10s} {'d mu':>10s} {'pcg its':>8s}")
for f in runs:
    g = np.load(f)
    geom = dict(vertices=g["vertices"], triangles=g["triangles"], mu0=g["mu0"], mu1=g["mu1"])
    kw = {k[3:]: (g[k].tolist() if g[k].ndim else g[k].item()) for k in g.files if k.startswith("kw_")}
    kw.pop("tol_checkpoints", None)
    for tol in (1e-6, 1e-7, 1e-8, 1e-9, 1e-10, 1e-11):
        sol, hist = solver_socp(int(g["n_time"]), geom, cg_tol=tol, mg_coarsest=24, **kw)
        same = int(hist.kkt_iteration[-1]) == int(g["last_iteration"])
        c, cw = hist.history["Transportation cost"][-1], g["hist_Transportation_cost"][-1]
        dk = float("nan")
        if same and hist.kkt_errors.shape == g["hist_kkt_errors"].shape:
            m = ~np.isnan(g["hist_kkt_errors"]) & ~np.isnan(hist.kkt_errors)
            dk = float(np.max(np.abs(hist.kkt_errors[m] - g["hist_kkt_errors"][m]) / np.maximum(np.abs(g["hist_kkt_errors"][m]), 1e-12)))
        dmu = float(np.max(np.abs(sol["mu"] - g["sol_mu"])) / np.max(np.abs(g["sol_mu"])))
        print(f"{os.path.basename(f)[4:-4]:42s} {tol:8.0e} {str(same):>8s} {abs(c-cw)/abs(cw):10.2e} {dk:10.2e} {dmu:10.2e} {hist.solver_stats['cg_iterations']:8d}")
