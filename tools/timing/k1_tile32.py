"""K1 on 64-env tiles vs half tiles (RLS_K1_TILE32 = 0 | 1: read once per process, run once per setting; unset = the launcher's
own choice) over node counts and batch sizes, G(N, 2N) graphs: us per call and the fraction of 8 TB/s.
`RLS_K1_TILE32=1 python tools/timing/k1_tile32.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import graph as G, ops
from rlsolver_amd.graph import build_csr
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (the library itself reads no environment)
dev = torch.device("cuda:0")


def t(f, reps=20):
    f(); f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


print("RLS_K1_TILE32 =", os.environ.get("RLS_K1_TILE32", "(launcher's choice)"))
for n, m in ((800, 4694), (2000, 19990), (3000, 6000), (5000, 10000), (7000, 14000), (10000, 9999), (14000, 28000), (16000, 32000),
             (20000, 40000), (24000, 48000), (40000, 80000)):
    dg = ops.DeviceGraph(build_csr(G.generate_gnm(n, m, 22), num_nodes=n), dev)
    row = []
    for B in (1 << 12, 1 << 14, 1 << 16, 1 << 17):
        if B * n > 3 << 30:
            continue
        xs = ops.rand_spins(B, n, 3, dev)
        us = t(lambda: ops.maxcut_obj(dg, xs))
        row.append(f"B=2^{B.bit_length() - 1}: {us:8.1f} us {B * n / us / 1e6 / 8:.3f}")
    print(f"N={n:6d} E={m:6d}  " + "   ".join(row), flush=True)
