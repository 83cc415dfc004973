"""Checks on a DOT solution: mass conservation, negative mass, distance to an exact transport
(the reference's ``utils/evaluate_solution.py:7-75`` and ``utils/util.py`` norms; written independently, vectorised).

``mu`` is a mass array (n_layers, V) as returned by ``dots_socp_amd.socp.solver`` / ``solver_raw``."""
from __future__ import annotations

import numpy as np


def check_mass_conservation(mu):
    """RMS over the time layers of (total mass of the layer - 1)."""
    mass = np.asarray(mu, dtype=np.float64).sum(axis=1)
    return float(np.linalg.norm(mass - 1.0) / np.sqrt(mass.size)), mass


def check_negative_mass(mu):
    """RMS over the time layers of the (non-positive) sum of the negative entries of the layer."""
    mu = np.asarray(mu, dtype=np.float64)
    neg = np.where(mu < 0.0, mu, 0.0).sum(axis=1)
    return float(np.linalg.norm(neg) / np.sqrt(neg.size)), neg


def compare_with_exact_transportation(mu, mu_exact, area_vertices):
    """Relative L1 / L2 / Linf distance between the densities mu / (area_v / 3) of two mass arrays
    (evaluate_solution.py:48-58: each norm is divided by 1 + the norm of the exact density)."""
    w = np.asarray(area_vertices, dtype=np.float64)[None, :] / 3.0
    rho, rho_x = np.asarray(mu) / w, np.asarray(mu_exact) / w
    d = rho - rho_x
    l1 = lambda a: float(np.sum(np.abs(a) * w))                     # noqa: E731
    l2 = lambda a: float(np.sqrt(np.sum(a * a * w)))                # noqa: E731
    linf = lambda a: float(np.max(np.abs(a)))                       # noqa: E731
    return {"l1": l1(d) / (1.0 + l1(rho_x)), "l2": l2(d) / (1.0 + l2(rho_x)), "linf": linf(d) / (1.0 + linf(rho_x))}


def plane_exact_transportation(t_array, vertices, area_vertices, center0=(0.4, 0.4, 0.0), center1=(0.6, 0.6, 0.0),
                               scale0=2 * 0.1 ** 2, scale1=2 * 0.1 ** 2):
    """Displacement interpolation between two isotropic Gaussians on the plane (data/settings/plane.py:29-46):
    centre moves linearly, scale^(1/4) interpolates linearly; masses area_v * gaussian with every layer
    normalised to total mass 1, as the solver's mu0 / mu1 are (data/load_example.py:138-139)."""
    from .meshes import gaussian_density

    c0, c1 = np.asarray(center0, dtype=np.float64), np.asarray(center1, dtype=np.float64)
    out = np.empty((len(t_array), np.asarray(vertices).shape[0]))
    for k, t in enumerate(np.asarray(t_array, dtype=np.float64)):
        sigma = ((1.0 - t) * scale0 ** 0.25 + t * scale1 ** 0.25) ** 4
        out[k] = gaussian_density(vertices, area_vertices, (1.0 - t) * c0 + t * c1, sigma)
    return out
