"""Import shim for the reference's hot-path modules (fixture generation only).

TEST INFRASTRUCTURE.  This file never ships any reference source: it loads the
reference's own files from ``/root/reference`` at run time, inside THIS
container only, so that ``make_golden.py`` can record input/output vectors of
the reference implementation.  ``/root/reference`` does not exist on the GPU
box, so nothing in ``tests/`` imports this module at test time; only the
committed ``*.npz`` fixtures travel.

Why a shim is needed: the reference package requires
Python >= 3.12 (PEP-701 f-strings in ``utils/admm_tools.py:552,560``,
``typing.NotRequired`` in ``utils/type.py:3``) and ``numexpr`` / ``trimesh``,
none of which exist in this image (Python 3.10, no network).  The hot-path
modules themselves parse on 3.10.  The shim therefore

1. aliases ``typing.NotRequired`` to ``typing_extensions.NotRequired``;
2. registers a ``numexpr`` stand-in whose ``evaluate`` runs the same
   expression through numpy (every expression on the path is
   ``+ - * / **2 sqrt >=``, so the results are IEEE-identical);
3. pre-registers empty package modules so the reference's ``__init__`` files
   (which pull the py3.12-only ``interface.py``) never run;
4. loads ``utils/admm_tools.py`` from text with the two nested-quote f-strings
   rewritten (they only format log lines).
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import re
import sys
import types

REFERENCE_ROOT = os.environ.get("DOTS_REFERENCE_ROOT", "/root/reference")


def _install_numexpr_stub():
    import numpy as np

    mod = types.ModuleType("numexpr")
    _names = {"sqrt": np.sqrt, "exp": np.exp, "abs": np.abs, "where": np.where}

    def evaluate(expr, local_dict=None, global_dict=None, out=None, **_kw):
        if local_dict is None:
            frame = sys._getframe(1)
            scope = dict(frame.f_globals)
            scope.update(frame.f_locals)
        else:
            scope = dict(local_dict)
        scope.update(_names)
        res = eval(expr, {"__builtins__": {}}, scope)  # noqa: S307 - fixed expressions of the reference
        if out is not None:
            out[...] = res
            return out
        return np.asarray(res)

    mod.evaluate = evaluate
    mod.set_num_threads = lambda n: None
    mod.detect_number_of_cores = lambda: os.cpu_count() or 1
    sys.modules["numexpr"] = mod


def _register_namespace(name: str, path: str):
    mod = types.ModuleType(name)
    mod.__path__ = [path]
    sys.modules[name] = mod
    return mod


def _load_from_text(modname: str, path: str, patch=None):
    with open(path, "r", encoding="utf-8") as fh:
        src = fh.read()
    if patch is not None:
        src = patch(src)
    mod = types.ModuleType(modname)
    mod.__file__ = path
    sys.modules[modname] = mod
    exec(compile(src, path, "exec"), mod.__dict__)  # noqa: S102
    return mod


def _patch_admm_tools(src: str) -> str:
    # f"{self._separate_symbol("...")}" -> py3.10-compatible quoting (log text only)
    return re.sub(r'self\._separate_symbol\("([^"]*)"\)', r"self._separate_symbol('\1')", src)


def _patch_evaluate_solution(src: str) -> str:
    # f"... {error_transportation["l1"]:.2e}" -> single quotes inside the f-string (log text only)
    return re.sub(r'error_transportation\["(l1|l2|linf)"\]', r"error_transportation['\1']", src)


_loaded = None


def load_reference():
    """Return a namespace with the reference's hot-path modules."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")

    import typing
    import typing_extensions

    if not hasattr(typing, "NotRequired"):
        typing.NotRequired = typing_extensions.NotRequired  # type: ignore[attr-defined]

    import matplotlib

    matplotlib.use("Agg")
    _install_numexpr_stub()

    pkg = os.path.join(REFERENCE_ROOT, "dot_surface_socp")
    _register_namespace("dot_surface_socp", pkg)
    _register_namespace("dot_surface_socp.socp", os.path.join(pkg, "socp"))
    _register_namespace("dot_surface_socp.utils", os.path.join(pkg, "utils"))
    _register_namespace("dot_surface_socp.data", os.path.join(pkg, "data"))

    # config reads TOML with cwd-relative paths
    cwd = os.getcwd()
    os.chdir(REFERENCE_ROOT)
    try:
        importlib.import_module("dot_surface_socp.config")
        _load_from_text(
            "dot_surface_socp.utils.admm_tools",
            os.path.join(pkg, "utils", "admm_tools.py"),
            patch=_patch_admm_tools,
        )
        solver_mod = importlib.import_module("dot_surface_socp.socp.solver_socp")
        pre = importlib.import_module("dot_surface_socp.utils.surface_pre_computations_socp")
        lap = importlib.import_module("dot_surface_socp.utils.laplacian_inverse_socp")
        cv = importlib.import_module("dot_surface_socp.utils.condition_validator")
        cvw = importlib.import_module("dot_surface_socp.utils.condition_validator_wrapper")
        typ = importlib.import_module("dot_surface_socp.utils.type")
        # the steps either side of the path: decorators, solution checks, example settings
        decorator = importlib.import_module("dot_surface_socp.socp.solver_decorator")
        evaluate = _load_from_text(
            "dot_surface_socp.utils.evaluate_solution",
            os.path.join(pkg, "utils", "evaluate_solution.py"),
            patch=_patch_evaluate_solution,
        )
        _register_namespace("dot_surface_socp.data.settings", os.path.join(pkg, "data", "settings"))
        settings = {}
        for fn in sorted(os.listdir(os.path.join(pkg, "data", "settings"))):
            if fn.endswith(".py") and fn != "__init__.py":
                settings[fn[:-3]] = importlib.import_module("dot_surface_socp.data.settings." + fn[:-3])
    finally:
        os.chdir(cwd)

    def _file_module(name, relpath):
        spec = importlib.util.spec_from_file_location(name, os.path.join(pkg, relpath))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    data_util = importlib.import_module("dot_surface_socp.data.util")
    plane_mesh = _file_module("_ref_plane_mesh", "data/meshes/plane.py")
    plane_setting = _file_module("_ref_plane_setting", "data/settings/plane.py")

    ns = types.SimpleNamespace(
        solver_socp=solver_mod.solver_socp,
        solver_module=solver_mod,
        pre=pre,
        lap=lap,
        admm_tools=sys.modules["dot_surface_socp.utils.admm_tools"],
        condition_validator=cv,
        condition_validator_wrapper=cvw,
        type=typ,
        data_util=data_util,
        plane_mesh=plane_mesh,
        plane_setting=plane_setting,
        decorator=decorator,
        evaluate_solution=evaluate,
        settings=settings,
    )
    _loaded = ns
    return ns


if __name__ == "__main__":
    ref = load_reference()
    print("reference solver imported:", ref.solver_socp)
