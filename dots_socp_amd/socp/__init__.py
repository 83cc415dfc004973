"""Solver plug-ins with the reference's names (``dot_surface_socp/socp/__init__.py:6-11``).

``solver_raw``  SOCP solver + conversion to DOT units          (socp/solver_decorator.py:10-27, utils/type.py:48-65)
``solver``      ``solver_raw`` on the time-centred grid          (socp/solver_decorator.py:29-54)

Both take ``(n_time, geometry, **kwargs)`` and return ``(solution, run_history)``; they can be
passed as ``solver=`` to the reference's ``run_dot_surface`` (interface.py:106-134).
"""
import numpy as np

from .solver_socp import solver_socp

__all__ = ["solver_socp", "solver_raw", "solver"]


def _socp_to_dot(solution_socp, geom):
    """translate_solution_socp_to_dot (utils/type.py:48-65): densities -> masses / fluxes."""
    area_v = np.asarray(geom["area_vertices"], dtype=np.float64)
    area_t = np.asarray(geom["area_triangles"], dtype=np.float64)
    out = {
        "mu": solution_socp["mu"] * (area_v[np.newaxis, :] / 3.0),
        "E": solution_socp["E"] * area_t[np.newaxis, :, np.newaxis],
    }
    if solution_socp.get("checkpoints"):
        out["checkpoints"] = [
            {
                "mu": cp["mu"] * (area_v[np.newaxis, :] / 3.0),
                "E": cp["E"] * area_t[np.newaxis, :, np.newaxis],
                "iteration": cp["iteration"], "time": cp["time"], "kkt": cp["kkt"],
            }
            for cp in solution_socp["checkpoints"]
        ]
    return out


def _geometry_with_areas(geometry):
    if "area_vertices" in geometry and "area_triangles" in geometry:
        return geometry
    from ..meshes import triangle_areas, vertex_areas

    g = dict(geometry)
    g["area_triangles"] = triangle_areas(g["vertices"], g["triangles"])
    g["area_vertices"] = vertex_areas(np.asarray(g["vertices"]).shape[0], g["triangles"], g["area_triangles"])
    return g


def solver_raw(n_time, geometry, **kwargs):
    """Solve the DOT problem with the GPU SOCP solver; solution on the time-staggered grid."""
    solution_socp, run_history = solver_socp(n_time, geometry, **kwargs)
    return _socp_to_dot(solution_socp, _geometry_with_areas(geometry)), run_history


solver_raw.__name__ = "dot_solver_socp"


def _to_time_centered(solution_dot, mu0, mu1):
    mid = 0.5 * (solution_dot["mu"][:-1] + solution_dot["mu"][1:])
    solution_dot["mu"] = np.concatenate([mu0[None, :], mid, mu1[None, :]], axis=0)


def solver(n_time, geometry, **kwargs):
    """``solver_raw`` with the density moved to the time-centred grid and mu0 / mu1 as end points."""
    mu0 = np.asarray(geometry["mu0"], dtype=np.float64)
    mu1 = np.asarray(geometry["mu1"], dtype=np.float64)
    solution_dot, run_history = solver_raw(n_time, geometry, **kwargs)
    _to_time_centered(solution_dot, mu0, mu1)
    for cp in solution_dot.get("checkpoints") or []:
        _to_time_centered(cp, mu0, mu1)
    return solution_dot, run_history


solver.__name__ = "dot_solver_socp_center"
