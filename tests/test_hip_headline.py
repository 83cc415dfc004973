"""BASELINE.json's configurations at FULL size on the GPU against the runs recorded by tests/golden/make_headline.py
(the reference itself for sphere10k / knot / knot63, the CPU oracle for torus100k): same stopping iteration, same
lazy-KKT pattern, KKT values / cost / objective within 1e-6, sampled solution within 1e-5."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, has_gpu
from dots_socp_amd import meshes

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a GPU")]

WORKLOADS = {
    "sphere10k": dict(example="sphere", kw=dict(level=5)),
    "knot": dict(example="knot", kw={}),
    "knot63": dict(example="knot", kw={}),
    "torus100k": dict(example="torus", kw=dict(nu=400, nv=250)),
}
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "headline_*.npz")))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[9:-4] for p in FIXTURES])
def test_headline_configuration(path):
    from dots_socp_amd.socp import solver_socp

    g = np.load(path)
    name = os.path.basename(path)[9:-4]
    wl = WORKLOADS[name]
    geom, scale = meshes.example(wl["example"], **wl["kw"])
    # the fixture was recorded on exactly this geometry
    chk = np.array([geom["vertices"].sum(), np.abs(geom["vertices"]).sum(), float(geom["triangles"].sum())])
    assert np.allclose(chk, g["vertices_checksum"], rtol=1e-13)
    assert abs(scale - float(g["scale_factor"])) < 1e-14
    sol, hist = solver_socp(int(g["n_time"]), geom, nit=20000, tol=float(g["tol"]), congestion=float(g["congestion"]), time_limit=1e9)
    assert int(hist.kkt_iteration[-1]) == int(g["last_iteration"])
    want, got = g["hist_kkt_errors"], hist.kkt_errors
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want)), "lazy KKT schedule differs"
    m = ~np.isnan(want)
    assert np.allclose(got[m], want[m], rtol=1e-6, atol=1e-13)
    for key in ("Transportation cost", "Objective value"):
        assert np.allclose(hist.history[key], g["hist_" + key.replace(" ", "_")], rtol=1e-6, atol=0, equal_nan=True), key
    mu = sol["mu"]
    assert np.max(np.abs(mu[:, ::40] - g["mu_sample"])) < 1e-5 * np.max(np.abs(g["mu_sample"]))
    assert np.allclose(mu.sum(axis=1), g["mu_layer_sum"], rtol=1e-6)
    assert np.allclose(np.sqrt((mu * mu).sum(axis=1)), g["mu_layer_norm"], rtol=1e-6)
    assert abs(np.sqrt((sol["E"] ** 2).sum()) - float(g["E_norm"])) < 1e-6 * float(g["E_norm"])
