"""CPU sanitizer job: the host-only half of the C ABI (csrc/rls_host.cpp: the schedule builders)
and the C oracle (oracle/oracle.c) compiled with -fsanitize=address,undefined and driven over random graphs -- hubs,
isolated nodes, N = 1 -- by tools/host_sanitize.cpp.  Sanitizers run on the CPU build only (the GPU pool has none)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("gcc") is None, reason="needs gcc / g++")
def test_host_builders_and_c_oracle_under_asan_ubsan(tmp_path):
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
    inc = [f"-I{ROOT}/include", f"-I{ROOT}/rlsolver_amd/csrc"]
    obj = str(tmp_path / "oracle.o")
    exe = str(tmp_path / "host_sanitize")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wno-unknown-pragmas", *san, "-c", f"{ROOT}/oracle/oracle.c", "-o", obj], check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", *san, *inc, f"{ROOT}/tools/host_sanitize.cpp", f"{ROOT}/rlsolver_amd/csrc/rls_host.cpp",
                    obj, "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    p = subprocess.run([exe, "150", "7"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-6000:])
    assert "clean" in p.stdout
