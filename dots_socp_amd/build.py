"""Build libdotsocp_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo).

    python -m dots_socp_amd.build [--force]
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libdotsocp_hip.so")
SOURCES = ["dots_api.hip", "kernels_alm.hip", "kernels_cg.hip", "kernels_mg.hip", "kernels_front.hip", "kernels_factor.hip", "dissect.hip", "kernels_kkt.hip"]
EXPORTS = os.path.join(CSRC, "exports.map")      # linker version script: only dots_* is exported
HEADERS = [os.path.join(CSRC, "dots_dev.h"), os.path.join(PKG_DIR, "..", "include", "dots_socp_hip.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-ffp-contract=off", "-fvisibility=hidden"]      # (the header's declarations are the exports)
FLAGS += os.environ.get("DOTS_HIPCC_FLAGS", "").split()      # extra -D switches for A/B measurements of compile-time variants


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


STAMP = os.path.join(CSRC, ".build_stamp.json")     # {object or library: sha256 of everything it was built from}


def _toolchain() -> str:
    r = subprocess.run([hipcc(), "--version"], capture_output=True, text=True)
    return r.stdout.strip()


def _digest(paths, extra) -> str:
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    h.update(repr(extra).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Freshness is decided by CONTENT, not by mtime: an object is rebuilt when the hash of (its source, the headers,
    the flags, the hipcc version) differs from the one recorded when it was last built (the .o / .so are git-ignored but
    travel in-tree, so mtimes prove nothing after a checkout or a copy).  Prints what was rebuilt."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES]
    cc = hipcc()
    tool = _toolchain()
    try:
        with open(STAMP) as fh:
            stamp = json.load(fh)
    except (OSError, ValueError):
        stamp = {}
    want = {os.path.basename(o): _digest([s] + HEADERS, (FLAGS, tool)) for s, o in zip(srcs, objs)}
    want[os.path.basename(LIB_PATH)] = hashlib.sha256((repr(sorted(want.items())) + open(EXPORTS).read()).encode()).hexdigest()

    def stale(path):
        return force or not os.path.exists(path) or stamp.get(os.path.basename(path)) != want[os.path.basename(path)]

    def compile_one(pair):
        src, obj = pair
        if not stale(obj):
            return None
        cmd = [cc, "-c", src, "-o", obj] + FLAGS
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return os.path.basename(obj), r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        done = [d for d in ex.map(compile_one, zip(srcs, objs)) if d]
    relink = bool(done) or stale(LIB_PATH)
    if relink:
        r = subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", f"-Wl,--version-script={EXPORTS}", "-o", LIB_PATH] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if done or relink:
        with open(STAMP, "w") as fh:
            json.dump(want, fh, indent=0, sort_keys=True)
    print(f"[dots_socp_amd.build] compiled: {', '.join(n for n, _ in done) or 'nothing'}; "
          f"{'linked ' + os.path.basename(LIB_PATH) if relink else 'library up to date'} ({ARCH})", file=sys.stderr)
    if verbose:
        for _, note in done:
            if note:
                sys.stderr.write(note)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
