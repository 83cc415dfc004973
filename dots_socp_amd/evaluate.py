"""Checks on a DOT solution: mass conservation, negative mass, distance to an exact transport
(the reference's ``utils/evaluate_solution.py:7-75`` and ``utils/util.py`` norms; written independently, vectorised).

``mu`` is a mass array (n_layers, V) as returned by ``dots_socp_amd.socp.solver`` / ``solver_raw``."""
from __future__ import annotations

import logging

import numpy as np

logger = logging.getLogger("dots_socp_amd")


def _l1(v, weight):
    """utils/util.py:32-46: a 2-D (layers, V) array is a time integral: the weighted sum times 1 / n_layers."""
    s = float(np.sum(np.abs(v) * weight))
    return s / v.shape[0] if v.ndim == 2 else s


def _l2(v, weight):
    """utils/util.py:48-62: sqrt(weighted sum of squares [* 1 / n_layers for a 2-D array])."""
    s = float(np.sum(v * v * weight))
    return float(np.sqrt(s / v.shape[0] if v.ndim == 2 else s))


def check_mass_conservation(mu, verbose=False):
    """RMS over the time layers of (total mass of the layer - 1)   (evaluate_solution.py:7-22; returns the scalar)."""
    mass = np.asarray(mu, dtype=np.float64).sum(axis=1)
    err = float(np.linalg.norm(mass - 1.0) / np.sqrt(mass.shape[0]))
    if verbose:
        logger.info("Mass Conservation Violation: %.2e", err)
    return err


def check_negative_mass(mu, verbose=False):
    """(RMS over the time layers of the sum of the negative entries of the layer, those sums)   (evaluate_solution.py:24-45)."""
    mu = np.asarray(mu, dtype=np.float64)
    neg = np.where(mu < 0.0, mu, 0.0).sum(axis=1)
    err = float(np.linalg.norm(neg) / np.sqrt(neg.shape[0]))
    if verbose:
        logger.info("Non-Negative Mass Violation: %.2e", err)
    return err, neg


def compare_with_exact_transportation(mu, mu_exact, geometry, verbose=False):
    """Relative L1 / L2 / Linf distance between the densities mu / (area_v / 3) of two mass arrays
    (evaluate_solution.py:47-58 with the norms of utils/util.py:32-67: each norm is divided by 1 + the norm of the
    exact density; the L1 and L2 norms of a (layers, V) array carry the time step 1 / layers).
    ``geometry``: the geometry dict (its "area_vertices" is used), as in the reference."""
    w = np.asarray(geometry["area_vertices"], dtype=np.float64) / 3.0
    mu, mu_exact = np.asarray(mu, dtype=np.float64), np.asarray(mu_exact, dtype=np.float64)
    if mu.ndim == 2:
        w = w[np.newaxis, :]
    rho, rho_x = mu / w, mu_exact / w
    d = rho - rho_x
    linf = lambda a: float(np.max(np.abs(a)))                       # noqa: E731
    err = {"l1": _l1(d, w) / (1.0 + _l1(rho_x, w)), "l2": _l2(d, w) / (1.0 + _l2(rho_x, w)), "linf": linf(d) / (1.0 + linf(rho_x))}
    if verbose:
        logger.info("L_1 Error: %.2e  L_2 Error: %.2e  L_Inf Error: %.2e", err["l1"], err["l2"], err["linf"])
    return err


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
