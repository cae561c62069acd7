"""Build librlsolver_hip.so (gfx950) in-tree with hipcc.  No torch headers, no hipify:
the kernels are plain HIP behind a C ABI (include/rlsolver_hip.h).

    python -m rlsolver_amd.build            # build if stale
    python -m rlsolver_amd.build --force
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_NAME = "librlsolver_hip.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)
ARCH = "gfx950"

SOURCES = ["rls_abi.hip", "rls_maxcut.hip", "rls_step.hip", "rls_mcpg.hip", "rls_tsp.hip", "rls_qubo.hip", "rls_spin.hip", "rls_localsearch.hip", "rls_isco.hip", "rls_track.hip"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _deps():
    deps = _sources()
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return deps


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library; returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    objs = []
    obj_dir = os.path.join(PKG_DIR, "csrc", "build")
    os.makedirs(obj_dir, exist_ok=True)
    common = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
              f"-I{INCLUDE}", f"-I{CSRC}", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]
    common += os.environ.get("RLS_EXTRA_CFLAGS", "").split()   # dev builds, e.g. -DRLS_PROF
    procs = []
    for src in _sources():
        obj = os.path.join(obj_dir, os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        if (not force and os.path.exists(obj)
                and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in _deps() if not d.endswith(".hip") or d == src)):
            continue
        cmd = common + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB_PATH


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print("built", path)
