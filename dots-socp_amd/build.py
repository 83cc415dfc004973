"""Build libdotsocp_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo).

    python -m dots_socp_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libdotsocp_hip.so")
SOURCES = ["dots_api.hip", "kernels_alm.hip", "kernels_cg.hip", "kernels_mg.hip", "kernels_front.hip", "kernels_factor.hip", "dissect.hip", "kernels_kkt.hip"]
HEADERS = [os.path.join(CSRC, "dots_dev.h"), os.path.join(PKG_DIR, "..", "include", "dots_socp_hip.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-ffp-contract=off"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES]
    cc = hipcc()

    def compile_one(pair):
        src, obj = pair
        if not force and not _newer(obj, [src] + HEADERS):
            return None
        cmd = [cc, "-c", src, "-o", obj] + FLAGS
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        notes = list(ex.map(compile_one, zip(srcs, objs)))
    if force or _newer(LIB_PATH, objs):
        r = subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        for n in notes:
            if n:
                sys.stderr.write(n)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
