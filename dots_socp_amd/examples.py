"""Boundary densities of the reference's packaged examples and the example loader around them.

The reference keeps one ``get_mu(area_vertices, vertices)`` per example in ``dot_surface_socp/data/settings/*.py``
(Python loops over the vertices) and normalises the two densities to unit mass in
``data/load_example.py:100-151``.  Here every recipe is a short table of vectorised terms built from the three
primitives of ``data/util.py:6-30``:

    cut_off(x, sigma)    1 for x <= 0, 0 for x >= sigma, (s - 1)^2 (s + 1)^2 with s = x / sigma in between
    gaussian(v, c, s)    exp(-|v - c|^2 / s)
    cut(cond, val)       val where cond else 0

so that a real ``knots_5.off`` (the packaged meshes are Git-LFS blobs) runs unchanged once it is present:

    geometry = load_example("knots_5", "/path/to/knots_5.off")
    solution, history = dots_socp_amd.socp.solver(31, geometry, tol=1e-4)

``tests/golden/settings_get_mu.npz`` holds the reference's own ``get_mu`` outputs for every recipe on one synthetic
mesh; ``tests/test_host_cpu.py`` compares this module with them.
"""
from __future__ import annotations

import os

import numpy as np

from . import meshes


# ---- primitives (data/util.py:6-30) ---------------------------------------------------------------
def cut_off(x, sigma):
    s = np.asarray(x, dtype=np.float64) / sigma
    return np.where(s <= 0.0, 1.0, np.where(s >= 1.0, 0.0, (s - 1.0) ** 2 * (s + 1.0) ** 2))


def _dist(v, c):
    return np.sqrt(np.sum((v - np.asarray(c, dtype=np.float64)[None, :]) ** 2, axis=1))


def gaussian(v, c, scale):
    return np.exp(-_dist(v, c) ** 2 / scale)


def _ball_gaussian(v, c, radius, scale):
    """cut(|v - c| < radius, gaussian(v, c, scale))  (knots_3.py:18-20, knots_5.py:18-20)."""
    return np.where(_dist(v, c) < radius, gaussian(v, c, scale), 0.0)


def _step(cond):
    return np.where(cond, 1.0, 0.0)


# ---- the recipes: name -> f(vertices) -> (shape0, shape1); the densities are area_vertices * shape ---------------
def _airplane(v):          # settings/airplane.py:10-12
    return cut_off(-(v[:, 2] - 0.5), 0.3), cut_off(v[:, 2] + 0.1, 0.3)


def _armadillo(v):         # settings/armadillo.py:10-12
    return cut_off(-v[:, 0] + 0.1, 0.15), cut_off(v[:, 0] + 0.1, 0.15)


def _audi(v):              # settings/audi.py:10-34
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    m0 = cut_off(x + 0.357, 0.007) * cut_off(y + 0.9, 0.1) * cut_off(-z + 0.02748, 0.00422)
    m1 = cut_off(-x + 0.715, 0.0143) * cut_off(y + 0.9, 0.1) * cut_off(z + 0.2389, 0.02114)
    m1 = m1 + cut_off(-x + 0.715, 0.0143) * cut_off(y + 0.9, 0.1) * cut_off(-z + 0.3023, 0.02114)
    m1 = m1 + cut_off(-x + 0.286, 0.0143) * cut_off(y + 0.9, 0.1) * cut_off(z + 1.0844, 0.02114)
    return m0, m1


def _bunny(v):             # settings/bunny.py:12-14
    x, y = v[:, 0], v[:, 1]
    return _step(x > 0.03), cut_off(-y + 0.3, 0.5) * _step(x < -0.06) * _step(y < 0.11) * _step(y > 0.05)


def _default(v):           # settings/default.py:10-12, settings/robot.py:10-12
    return cut_off(v[:, 0], 0.1), cut_off(v[:, 1], 0.1)


def _eight(v):             # settings/eight.py:10-27
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    m0 = cut_off(x + 0.2626, 0.01) * cut_off(y + 0.9108, 0.1012)
    m1 = cut_off(-x + 0.9696, 0.0202) * cut_off(y + 0.9108, 0.1012) * cut_off(z + 0.3371, 0.0337)
    m1 = m1 + cut_off(-x + 0.9696, 0.0202) * cut_off(y + 0.9108, 0.1012) * cut_off(z + 0.4383, 0.0337)
    return m0, m1


def _face_like(centers):
    """settings/face.py:14-26, settings/refined_face.py:14-34: a smooth window on the front of the head against
    Gaussian blobs (scale 0.1^2) around fixed vertices."""
    def recipe(v):
        alpha = 0.1 * v[:, 0] + v[:, 1]
        beta = -v[:, 0] + 0.1 * v[:, 1]
        m0 = np.where(v[:, 2] >= -0.1,
                      cut_off(-0.2 - alpha, 0.3) * cut_off(alpha - 0.15, 0.3) * cut_off(0.1 - beta, 0.3) * cut_off(beta - 0.45, 0.3), 0.0)
        m1 = np.zeros(v.shape[0])
        for c in centers:
            m1 = m1 + np.exp(-_dist(v, v[c]) ** 2 / 0.1 ** 2)
        return m0, m1
    return recipe


def _hand(v):              # settings/hand.py:12-14
    return _step(v[:, 1] < -0.5), _step(v[:, 1] > 0.4)


def _refined_hand(v):      # settings/refined_hand.py:13-20
    m0 = np.exp(-_dist(v, v[5982]) ** 2 / 0.1 ** 2) + np.exp(-_dist(v, v[1347]) ** 2 / 0.1 ** 2)
    return m0, _step(v[:, 1] > 0.4)


def _hills(v):             # settings/hills.py:10-16
    return gaussian(v, v[1191], 1.0), gaussian(v, v[9505], 1.0)


def _knots_3(v):           # settings/knots_3.py:11-20
    c0, c11, c12 = [0.0888, 1.282, 0.512], [-1.035, -1.087, 0.300], [1.212, -0.594, 0.455]
    return _ball_gaussian(v, c0, 0.5, 0.3), _ball_gaussian(v, c11, 0.3, 0.3) + _ball_gaussian(v, c12, 0.3, 0.3)


def _knots_5(v):           # settings/knots_5.py:11-20
    return _ball_gaussian(v, v[2786], 0.5, 0.5), _ball_gaussian(v, v[1232], 0.5, 0.5) + _ball_gaussian(v, v[406], 0.5, 0.5)


def _plane(v):             # settings/plane.py:6-11,20-25
    return gaussian(v, [0.4, 0.4, 0.0], 2 * 0.1 ** 2), gaussian(v, [0.6, 0.6, 0.0], 2 * 0.1 ** 2)


def _punctured_ball(v):    # settings/punctured_ball.py:10-12
    return cut_off(-v[:, 1] + 0.875, 0.1), cut_off(v[:, 1] + 0.875, 0.1)


def _ring(v):              # settings/ring.py:10-12
    return cut_off(v[:, 0] - 0.5, 0.5), cut_off(v[:, 0] + 0.7, 0.5)


def _square_regular(v):    # settings/square_regular.py:12-19
    x0, x10, x11 = [0.33, 0.5, 0.0], [0.8, 0.2, 0.0], [0.8, 0.8, 0.0]
    m1 = cut_off((_dist(v, x10) - 0.1) * 2.0, 0.1) + cut_off((_dist(v, x11) - 0.1) * 2.0, 0.1)
    return cut_off(_dist(v, x0) - 0.1, 0.1), m1


RECIPES = {
    "default": _default, "robot": _default,
    "airplane": _airplane, "refined_airplane": _airplane,
    "armadillo": _armadillo, "refined_armadillo": _armadillo,
    "audi": _audi,
    "bunny": _bunny, "refined_bunny": _bunny,
    "eight": _eight,
    "face": _face_like((4492, 4225)), "refined_face": _face_like((10129, 9458, 11792, 12638, 3146)),
    "hand": _hand, "refined_hand": _refined_hand,
    "hills": _hills,
    "knots_3": _knots_3, "knots_5": _knots_5,
    "plane": _plane,
    "punctured_ball": _punctured_ball, "refined_punctured_ball": _punctured_ball,
    "ring": _ring,
    "square_regular": _square_regular,
}
# example name -> packaged mesh file (data/load_example.py, data/meshes/); "sphere" reads its densities from
# data/settings/data_mu/sphere_puncture_data_mu{0,1}.txt (settings/sphere.py:4-8)
MESH_FILES = {
    "airplane": "airplane_62.off", "refined_airplane": "refined_airplane_62.off", "armadillo": "armadillo.off",
    "refined_armadillo": "refined_armadillo.off", "audi": "audi.off", "bunny": "bunny.off", "refined_bunny": "refined_bunny.off",
    "eight": "eight.off", "face": "face_vector_field_319.off", "refined_face": "refined_face_vector_field_319.off",
    "hand": "hand_3k.off", "refined_hand": "refined_hand_3k.off", "hills": "hills.off", "knots_3": "knots_3.off",
    "knots_5": "knots_5.off", "punctured_ball": "punctured_ball.off", "refined_punctured_ball": "refined_punctured_ball.off",
    "ring": "ring.off", "robot": "robot.off", "sphere": "sphere_puncture.off", "square_regular": "square_regular_100.off",
}


def get_mu(name, area_vertices, vertices, data_dir=None):
    """The un-normalised boundary densities ``(mub0, mub1)`` of example ``name`` (``settings/<name>.get_mu``).

    ``area_vertices`` is the reference's vertex area (the SUM of the incident triangle areas,
    surface_pre_computations_socp.py:120-132).  ``sphere`` loads its densities from ``data_dir``."""
    v = np.asarray(vertices, dtype=np.float64)
    a = np.asarray(area_vertices, dtype=np.float64)
    if name == "sphere":
        if data_dir is None:
            raise ValueError("the 'sphere' example reads sphere_puncture_data_mu0/1.txt: pass data_dir")
        return (np.loadtxt(os.path.join(data_dir, "sphere_puncture_data_mu0.txt")),
                np.loadtxt(os.path.join(data_dir, "sphere_puncture_data_mu1.txt")))
    if name not in RECIPES:
        raise ValueError(f"unknown example {name!r}; known: {sorted(RECIPES) + ['sphere']}")
    s0, s1 = RECIPES[name](v)
    return a * s0, a * s1


def load_example(name, mesh_file, data_dir=None, normalize=True):
    """``load_example`` + ``normalize_geometry`` (data/load_example.py:100-151, socp/data_preprocessing.py:5-37):
    read the OFF mesh, build the densities of example ``name`` on the ORIGINAL coordinates, normalise each to unit
    mass, then scale the geometry to the unit box.  Returns ``(geometry, scale_factor)``; the reported transport
    cost is ``history["Transportation cost"] / scale_factor**2`` (interface.py:303-308)."""
    vertices, triangles = meshes.read_off(mesh_file)[:2]
    area_t = meshes.triangle_areas(vertices, triangles)
    area_v = meshes.vertex_areas(vertices.shape[0], triangles, area_t)
    mu0, mu1 = get_mu(name, area_v, vertices, data_dir=data_dir)
    if not (mu0.sum() > 0 and mu1.sum() > 0):
        raise ValueError(f"example {name!r}: a boundary density has no mass on this mesh")
    geometry, scale = meshes.make_geometry(vertices, triangles, normalize=normalize)
    geometry["mu0"], geometry["mu1"] = mu0 / mu0.sum(), mu1 / mu1.sum()
    return geometry, scale
