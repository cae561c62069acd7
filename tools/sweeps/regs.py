#!/usr/bin/env python3
"""VGPR / spill report of every kernel in a .hip file (compiles to ISA text with hipcc --cuda-device-only -S)."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for f in sys.argv[1:]:
    out = f"/tmp/{os.path.basename(f)}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{root}/include",
                    f"-I{root}/rlsolver_amd/csrc", "-Wno-pass-failed", "--cuda-device-only", "-S", f, "-o", out],
                   stderr=subprocess.DEVNULL, check=True)
    s = open(out).read()
    for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n){0,12}?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n){0,6}?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', s):
        d = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        d = re.sub(r'\(.*', '', d).replace('void rls::', '')
        print(f"{d[:90]:90s} vgpr {m.group(3):>4} sgpr-spill {m.group(2):>3} vgpr-spill {m.group(4):>3}")
