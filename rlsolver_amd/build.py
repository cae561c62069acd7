"""Build librlsolver_hip.so (gfx950) in-tree with hipcc.  No torch headers, no hipify:
the kernels are plain HIP behind a C ABI (include/rlsolver_hip.h).

    python -m rlsolver_amd.build            # build if stale
    python -m rlsolver_amd.build --force
"""
from __future__ import annotations

import hashlib
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
OPS_NAME = "librlsolver_torch_ops.so"       # torch.ops.rlsolver_hip.*: host-only C++ over the C ABI (csrc/torch_ops.cpp)
OPS_PATH = os.path.join(PKG_DIR, OPS_NAME)
OPS_SRC = os.path.join(CSRC, "torch_ops.cpp")
ARCH = "gfx950"

SOURCES = ["rls_host.cpp", "rls_abi.hip", "rls_maxcut.hip", "rls_step.hip", "rls_mcpg.hip", "rls_tsp.hip", "rls_qubo.hip", "rls_spin.hip", "rls_localsearch.hip", "rls_isco.hip", "rls_track.hip"]


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


def _flags():
    return ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-pass-failed",
            f"--offload-arch={ARCH}"] + os.environ.get("RLS_EXTRA_CFLAGS", "").split()   # dev builds, e.g. -DRLS_PROF / -DRLS_DEV


def _digest(paths, extra=()) -> str:
    """Content hash of `paths` (by name relative to the repo + bytes) and of `extra` strings.  Staleness is decided by
    CONTENT, never by mtime: a transport that rewrites mtimes (a checkout beside shipped binaries, rsync without -t) neither
    triggers a 2.5-minute rebuild nor lets a binary older than its sources pass."""
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(os.path.relpath(p, ROOT).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    for e in extra:
        h.update(b"\1" + e.encode())
    return h.hexdigest()


def _stamp(path: str) -> str:
    return path + ".srchash"


def _stamp_matches(path: str, digest: str) -> bool:
    try:
        return os.path.exists(path) and open(_stamp(path)).read().strip() == digest
    except OSError:
        return False


def _write_stamp(path: str, digest: str) -> None:
    with open(_stamp(path), "w") as f:
        f.write(digest + "\n")


def lib_digest() -> str:
    return _digest(_deps(), _flags())


def ops_digest() -> str:
    import torch
    deps = [OPS_SRC] + [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return _digest(deps, [lib_digest(), torch.__version__])


def is_stale() -> bool:
    return not _stamp_matches(LIB_PATH, lib_digest())


def ops_is_stale() -> bool:
    return not _stamp_matches(OPS_PATH, ops_digest())


def build_torch_ops(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/torch_ops.cpp (no device code) against libtorch and link it to librlsolver_hip.so, in-tree."""
    if not force and not ops_is_stale():
        return OPS_PATH
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cxx = shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm}/include", f"-I{INCLUDE}"]
    cmd += [OPS_SRC, "-o", OPS_PATH, f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", f"-L{PKG_DIR}",
            "-lrlsolver_hip", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building {OPS_NAME} failed:\n{r.stdout}")
    _write_stamp(OPS_PATH, ops_digest())
    return OPS_PATH


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library; returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    objs = []
    obj_dir = os.path.join(PKG_DIR, "csrc", "build")
    os.makedirs(obj_dir, exist_ok=True)
    common = [_hipcc()] + _flags() + [f"-I{INCLUDE}", f"-I{CSRC}"]
    headers = [d for d in _deps() if d.endswith(".h")]
    procs = []
    for src in _sources():
        obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        dg = _digest(headers + [src], _flags())
        if not force and _stamp_matches(obj, dg):
            continue
        cmd = common + (["-x", "c++"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]     # rls_host.cpp: host-only C++
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, dg, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = None
    for src, obj, dg, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = failed or RuntimeError(f"hipcc failed on {src}:\n{out}")
            continue
        _write_stamp(obj, dg)
        if verbose and out.strip():
            print(out)
    if failed:
        raise failed
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    _write_stamp(LIB_PATH, lib_digest())
    return LIB_PATH


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print("built", path)
    print("built", build_torch_ops(force="--force" in sys.argv, verbose=True))
