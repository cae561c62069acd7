"""The local-search weights pre-pass at small batches: time per call.  Forms: RLS_NODE_STATS_MIN_B=0 forces the bit-sliced kernel,
RLS_NODE_STATS_MIN_B=1000000000 the element-parallel one (the lane = env tile form this tool measured at 850 us per call is gone)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import ops
from rlsolver_amd.graph import build_csr, generate_ba, generate_gnm
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (the library itself reads no environment)
dev = torch.device("cuda:0")
for name, n, mg in (("G22", 2000, generate_gnm(2000, 19990, 22)), ("G14", 800, generate_gnm(800, 4694, 14)), ("BA-1e4", 10000, generate_ba(10000, 5, 5))):
    g = ops.DeviceGraph(build_csr(mg, n, False), dev)
    row = []
    for B in (1, 64, 256, 1024, 2048, 4096):
        xs = torch.rand(B, n, device=dev) < 0.5
        for _ in range(3): ops.maxcut_ls_weights(g, xs, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.maxcut_ls_weights(g, xs, 1)
        e1.record(); torch.cuda.synchronize()
        row.append(f"B={B}: {e0.elapsed_time(e1) * 100:.0f}")
    print("min_b=" + os.environ.get("RLS_NODE_STATS_MIN_B", "auto"), name, " ".join(row), "us")
