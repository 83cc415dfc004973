"""Versus-exact study on the plane example (reference: interface.py:386-480 with data/settings/plane.py): the
optimal transport between two equal Gaussians is a translation, squared W2 distance 0.08, dynamic cost 0.04.
Runs the GPU solver on refining meshes / time grids and prints cost, density errors against the displacement
interpolation, mass conservation and negative mass.   usage: python profiles/studies/versus_exact.py [tol]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dots_socp_amd import evaluate, meshes  # noqa: E402
from dots_socp_amd.socp import solver  # noqa: E402

tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
print(f"tol {tol:g}")
print(f"{'n':>5s} {'V':>7s} {'T':>4s} {'its':>6s} {'sec':>7s} {'cost':>12s} {'cost-0.04':>10s} {'L1':>9s} {'L2':>9s} {'Linf':>9s} {'mass':>9s} {'neg':>9s}")
for n, T in ((20, 31), (40, 31), (80, 63), (160, 63), (240, 127)):
    geom, scale = meshes.example("plane", n=n)
    t0 = time.perf_counter()
    sol, hist = solver(T, geom, tol=tol, nit=20000)
    sec = time.perf_counter() - t0
    cost = hist.history["Transportation cost"][-1] / scale ** 2
    tt = np.linspace(0.0, 1.0, sol["mu"].shape[0])           # time-centred grid: the T + 1 nodes
    exact = evaluate.plane_exact_transportation(tt, geom["vertices"] / scale, geom["area_vertices"])
    err = evaluate.compare_with_exact_transportation(sol["mu"], exact, geom)
    mass = evaluate.check_mass_conservation(sol["mu"])
    neg, _ = evaluate.check_negative_mass(sol["mu"])
    print(f"{n:5d} {geom['vertices'].shape[0]:7d} {T:4d} {int(hist.kkt_iteration[-1]) + 1:6d} {sec:7.2f} {cost:12.8f} {cost - 0.04:10.2e} "
          f"{err['l1']:9.2e} {err['l2']:9.2e} {err['linf']:9.2e} {mass:9.2e} {neg:9.2e}", flush=True)
