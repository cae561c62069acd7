"""K5 (greedy sweep) and K6 (propose / accept, byte mask) on 64-env tiles vs half tiles (RLS_K5_TILE32 / RLS_K6_TILE32 = 1: read once
per process, run once per setting; unset = the launcher's choice: half tiles only past the 64-env tile): us per call.
`RLS_K5_TILE32=1 RLS_K6_TILE32=1 python tools/timing/k5_tile32.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import graph as G, ops
from rlsolver_amd.graph import build_csr
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (the library itself reads no environment)
dev = torch.device("cuda:0")


def t(f, reps=10):
    f(); f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


print("RLS_K5_TILE32 =", os.environ.get("RLS_K5_TILE32", "-"), " RLS_K6_TILE32 =", os.environ.get("RLS_K6_TILE32", "-"))
for tag, n, m, Bs in (("G14-sized", 800, 4694, (1 << 14, 1 << 16)), ("G22-sized", 2000, 19990, (1 << 12, 1 << 14, 1 << 16, 1 << 17)),
                      ("G70-sized", 10000, 9999, (1 << 14, 1 << 17)), ("N=20000", 20000, 40000, (1 << 14, 1 << 15)),
                      ("N=24000", 24000, 48000, (1 << 14, 1 << 15)), ("N=36000", 36000, 72000, (1 << 14,))):
    dg = ops.DeviceGraph(build_csr(G.generate_gnm(n, m, 22), num_nodes=n), dev)
    for B in Bs:
        xs = ops.rand_spins(B, n, 3, dev)
        vs = ops.maxcut_obj(dg, xs)
        x2, v2 = xs.clone(), vs.clone()
        k5 = t(lambda: ops.maxcut_greedy_sweep(dg, x2, v2))
        mask = torch.rand((B, n), device=dev) < 0.01
        k6 = t(lambda: ops.maxcut_propose_accept(dg, x2, mask, v2))
        print(f"{tag:10s} B=2^{B.bit_length() - 1}: K5 {k5:9.1f} us ({2 * B * n / k5 / 1e6 / 8:.3f})   K6 {k6:9.1f} us", flush=True)
