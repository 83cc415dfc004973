"""Deterministic surface-mesh and boundary-density generators.

The reference's packaged meshes are Git-LFS pointers in the mount
(e.g. ``dot_surface_socp/data/meshes/knots_5.off:1-3``), so every workload in
BASELINE.json is run on a generated stand-in of the same size class:

    icosphere(level=5)            V=10 242  F=20 480   "sphere ~10k"
    torus(400, 250)               V=100 000 F=200 000  "torus ~100k"
    torus_knot_tube(2, 5, ...)    V~4.3k    F~8.6k     knots_5 stand-in
    plane(n)                      flat hexagonal patch, analytic answer 0.04

Densities follow the reference's recipe (``data/settings/knots_5.py:15-20``,
``data/util.py:6-13``): area-weighted Gaussian bumps cut at a radius around fixed
vertices, each normalised to unit mass (``data/load_example.py:138-139``).
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------- #
# meshes
# --------------------------------------------------------------------------- #
def _unique_edges(triangles):
    t = np.asarray(triangles)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]], axis=0)
    e.sort(axis=1)
    return np.unique(e, axis=0)


def icosphere(level: int = 3, radius: float = 1.0):
    """Subdivided icosahedron projected on the sphere: V = 10*4^level + 2."""
    g = (1.0 + np.sqrt(5.0)) / 2.0
    v = np.array(
        [[-1, g, 0], [1, g, 0], [-1, -g, 0], [1, -g, 0], [0, -1, g], [0, 1, g],
         [0, -1, -g], [0, 1, -g], [g, 0, -1], [g, 0, 1], [-g, 0, -1], [-g, 0, 1]],
        dtype=np.float64,
    )
    f = np.array(
        [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
         [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
         [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]],
        dtype=np.int64,
    )
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        e.sort(axis=1)
        ue, inv = np.unique(e, axis=0, return_inverse=True)
        inv = np.asarray(inv).reshape(-1)
        mid = v[ue[:, 0]] + v[ue[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = v.shape[0]
        v = np.concatenate([v, mid], axis=0)
        nf = f.shape[0]
        m01, m12, m20 = base + inv[:nf], base + inv[nf:2 * nf], base + inv[2 * nf:]
        f = np.concatenate(
            [
                np.stack([f[:, 0], m01, m20], axis=1),
                np.stack([f[:, 1], m12, m01], axis=1),
                np.stack([f[:, 2], m20, m12], axis=1),
                np.stack([m01, m12, m20], axis=1),
            ],
            axis=0,
        )
    return radius * v, f


def _periodic_grid_triangles(nu: int, nv: int):
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    i1, j1 = (i + 1) % nu, (j + 1) % nv
    a, b, c, d = i * nv + j, i1 * nv + j, i1 * nv + j1, i * nv + j1
    return np.concatenate(
        [np.stack([a, b, c], axis=-1).reshape(-1, 3), np.stack([a, c, d], axis=-1).reshape(-1, 3)], axis=0
    ).astype(np.int64)


def torus(nu: int = 64, nv: int = 40, R: float = 1.0, r: float = 0.4):
    """Torus of major radius R, minor radius r on an nu x nv periodic grid: V = nu*nv, F = 2V."""
    u = 2 * np.pi * np.arange(nu) / nu
    w = 2 * np.pi * np.arange(nv) / nv
    uu, ww = np.meshgrid(u, w, indexing="ij")
    x = (R + r * np.cos(ww)) * np.cos(uu)
    y = (R + r * np.cos(ww)) * np.sin(uu)
    z = r * np.sin(ww)
    return np.stack([x, y, z], axis=-1).reshape(-1, 3), _periodic_grid_triangles(nu, nv)


def torus_knot_tube(p: int = 2, q: int = 5, nu: int = 216, nv: int = 20, R: float = 1.0, r: float = 0.45,
                    tube: float = 0.12):
    """Tube of radius ``tube`` around the (p, q) torus knot: stand-in for the reference's knots_5."""
    s = 2 * np.pi * np.arange(nu) / nu

    def curve(s):
        rad = R + r * np.cos(q * s)
        return np.stack([rad * np.cos(p * s), rad * np.sin(p * s), r * np.sin(q * s)], axis=-1)

    c = curve(s)
    ds = 1e-4
    tan = curve(s + ds) - curve(s - ds)
    tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    # a smooth periodic frame: normal = component of the radial direction orthogonal to the tangent
    radial = c.copy()
    radial[:, 2] = 0.0
    radial /= np.linalg.norm(radial, axis=1, keepdims=True)
    n1 = radial - np.sum(radial * tan, axis=1, keepdims=True) * tan
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(tan, n1)
    w = 2 * np.pi * np.arange(nv) / nv
    v = c[:, None, :] + tube * (np.cos(w)[None, :, None] * n1[:, None, :] + np.sin(w)[None, :, None] * n2[:, None, :])
    return v.reshape(-1, 3), _periodic_grid_triangles(nu, nv)


def plane(n: int = 20):
    """Flat patch of [0,1]^2 tiled by equilateral triangles of side 1/n (rows offset by half a side).

    Same family as the reference's procedural mesh (``data/meshes/plane.py:3-69``): n+1 vertices
    per row, rows spaced sqrt(3)/(2n) apart.  Written independently; vertex numbering is row-major.
    """
    dx = 1.0 / n
    dy = dx * np.sqrt(3.0) / 2.0
    rows = int(1.0 / dy) + 1
    cols = n + 1
    jj, ii = np.meshgrid(np.arange(cols), np.arange(rows))
    x = jj * dx + np.where(ii % 2 == 1, dx / 2.0, 0.0)
    y = ii * dy
    verts = np.stack([x, y, np.zeros_like(x)], axis=-1).reshape(-1, 3)
    idx = lambda i, j: i * cols + j  # noqa: E731
    tris = []
    for i in range(rows - 1):
        for j in range(cols - 1):
            if i % 2 == 0:
                tris.append([idx(i, j), idx(i, j + 1), idx(i + 1, j)])
                tris.append([idx(i, j + 1), idx(i + 1, j + 1), idx(i + 1, j)])
            else:
                tris.append([idx(i, j), idx(i + 1, j + 1), idx(i + 1, j)])
                tris.append([idx(i, j), idx(i, j + 1), idx(i + 1, j + 1)])
    return verts, np.asarray(tris, dtype=np.int64)


# --------------------------------------------------------------------------- #
# geometry dict (GeometryData of the reference, utils/type.py:6-13)
# --------------------------------------------------------------------------- #
def triangle_areas(vertices, triangles):
    v, t = np.asarray(vertices, dtype=np.float64), np.asarray(triangles)
    return 0.5 * np.linalg.norm(np.cross(v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 1]]), axis=1)


def vertex_areas(n_vertices, triangles, area_triangles):
    """Sum of incident triangle areas (the reference's un-divided ``area_vertices``,
    ``surface_pre_computations_socp.py:123``)."""
    t = np.asarray(triangles)
    out = np.zeros(n_vertices)
    for k in range(3):
        np.add.at(out, t[:, k], area_triangles)
    return out


def bump_density(vertices, area_vertices, centers, radius=0.5, sigma=0.5):
    """Sum of area-weighted truncated Gaussians around ``vertices[c]`` for c in centers, mass 1."""
    v = np.asarray(vertices, dtype=np.float64)
    mu = np.zeros(v.shape[0])
    for c in centers:
        d = np.linalg.norm(v - v[c], axis=1)
        mu += area_vertices * np.where(d < radius, np.exp(-d ** 2 / sigma), 0.0)
    return mu / mu.sum()


def gaussian_density(vertices, area_vertices, center, scale):
    """Un-truncated area-weighted Gaussian exp(-|x-c|^2/scale), mass 1 (``data/settings/plane.py:14-25``)."""
    d2 = np.sum((np.asarray(vertices, dtype=np.float64) - np.asarray(center)[None, :]) ** 2, axis=1)
    mu = area_vertices * np.exp(-d2 / scale)
    return mu / mu.sum()


def make_geometry(vertices, triangles, mu0=None, mu1=None, normalize=True):
    """Build the geometry dict the solver receives; optionally normalised into the unit box.

    Normalisation restates ``socp/data_preprocessing.py:5-37``: translate and scale so the
    bounding box is [0, s]^3 with longest side 1 (the trimesh centroid shift cancels in the
    final min-subtraction).  Returns ``(geometry, scale_factor)``.
    """
    v = np.asarray(vertices, dtype=np.float64).copy()
    t = np.asarray(triangles, dtype=np.int64)
    scale = 1.0
    if normalize:
        scale = 1.0 / (v.max(axis=0) - v.min(axis=0)).max()
        v = (v - v.min(axis=0)) * scale
    area_t = triangle_areas(v, t)
    area_v = vertex_areas(v.shape[0], t, area_t)
    geom = {
        "vertices": v,
        "triangles": t,
        "edges": _unique_edges(t),
        "area_triangles": area_t,
        "area_vertices": area_v,
    }
    if mu0 is not None:
        geom["mu0"] = np.asarray(mu0, dtype=np.float64)
        geom["mu1"] = np.asarray(mu1, dtype=np.float64)
    return geom, scale


def farthest_vertices(vertices, start: int, count: int):
    """Greedy farthest-point vertex indices (deterministic centres for the bumps)."""
    v = np.asarray(vertices, dtype=np.float64)
    picked = [int(start)]
    dist = np.linalg.norm(v - v[start], axis=1)
    for _ in range(count - 1):
        nxt = int(np.argmax(dist))
        picked.append(nxt)
        dist = np.minimum(dist, np.linalg.norm(v - v[nxt], axis=1))
    return picked


def example(name: str, **kw):
    """Named synthetic workloads (stand-ins for BASELINE.json's configs).

    Returns ``(geometry, scale_factor)`` with mu0 (one bump) and mu1 (two bumps) set, normalised.
    """
    if name == "plane":
        n = kw.get("n", 20)
        v, t = plane(n)
        at = triangle_areas(v, t)
        av = vertex_areas(v.shape[0], t, at)
        mu0 = gaussian_density(v, av, [0.4, 0.4, 0.0], 2 * 0.1 ** 2)   # data/settings/plane.py:5-11
        mu1 = gaussian_density(v, av, [0.6, 0.6, 0.0], 2 * 0.1 ** 2)
        return make_geometry(v, t, mu0, mu1, normalize=kw.get("normalize", True))
    if name == "sphere":
        v, t = icosphere(kw.get("level", 5))
    elif name == "torus":
        v, t = torus(kw.get("nu", 400), kw.get("nv", 250))
    elif name == "knot":
        v, t = torus_knot_tube(2, 5, kw.get("nu", 216), kw.get("nv", 20))
    else:
        raise ValueError(f"unknown example {name!r}")
    geom, scale = make_geometry(v, t)
    c = farthest_vertices(geom["vertices"], 0, 3)
    radius, sigma = kw.get("radius", 0.35), kw.get("sigma", 0.05)
    # recipe of data/settings/knots_5.py:15-20 on the normalised mesh (radius/sigma sized for the unit box)
    geom["mu0"] = bump_density(geom["vertices"], geom["area_vertices"], [c[0]], radius, sigma)
    geom["mu1"] = bump_density(geom["vertices"], geom["area_vertices"], [c[1], c[2]], radius, sigma)
    return geom, scale


def read_off(path):
    """Triangle mesh from an OFF file: ``(vertices (V,3) float64, triangles (F,3) int64, edges (3F,2) int64)``.

    Same contract as the reference's reader (``data/util.py:73-144``): first line ``OFF``, then the vertex and face
    counts, then V coordinate lines and F lines ``3 i j k``; empty lines are skipped; edges are the three directed
    sides of every triangle in file order; a malformed file raises ``ValueError``.  Written independently: the
    payload is parsed with numpy in two blocks instead of line by line (a 100k-vertex mesh reads in ~0.1 s)."""
    try:
        with open(path, "r") as fh:
            lines = [ln.split() for ln in fh]
    except OSError as exc:
        raise ValueError(f"Error reading .off file: {exc}") from exc
    lines = [ln for ln in lines if ln]
    if not lines or lines[0] != ["OFF"]:
        raise ValueError("Error reading .off file: Not a valid .off file")
    if len(lines) < 2 or len(lines[1]) < 2:
        raise ValueError("Error reading .off file: Invalid file format: missing vertex/triangle counts")
    try:
        nv, nf = int(lines[1][0]), int(lines[1][1])
        body = lines[2:]
        faces = [ln for ln in body if ln[0] == "3"]
        verts = [ln for ln in body if ln[0] != "3"]
        if len(verts) != nv:
            raise ValueError(f"Expected {nv} vertices but found {len(verts)}")
        if len(faces) != nf:
            raise ValueError(f"Expected {nf} triangles but found {len(faces)}")
        if any(len(ln) < 3 for ln in verts) or any(len(ln) < 4 for ln in faces):
            raise ValueError("Invalid vertex / triangle data")
        vertices = np.array([ln[:3] for ln in verts], dtype=np.float64).reshape(nv, 3)
        triangles = np.array([ln[1:4] for ln in faces], dtype=np.int64).reshape(nf, 3)
    except ValueError as exc:
        raise ValueError(f"Error reading .off file: {exc}") from exc
    edges = np.stack([triangles[:, [0, 1]], triangles[:, [1, 2]], triangles[:, [2, 0]]], axis=1).reshape(-1, 2)
    return vertices, triangles, edges


def write_off(path, vertices, triangles):
    """Write a triangle mesh as OFF (the inverse of ``read_off``; 17 significant digits)."""
    v, t = np.asarray(vertices, dtype=np.float64), np.asarray(triangles)
    with open(path, "w") as fh:
        fh.write("OFF\n%d %d 0\n" % (v.shape[0], t.shape[0]))
        for row in v:
            fh.write("%.17g %.17g %.17g\n" % tuple(row))
        for row in t:
            fh.write("3 %d %d %d\n" % tuple(int(i) for i in row))
