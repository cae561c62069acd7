"""K4 on rows past comfortable LDS staging (N = 44 000 .. 200 000, u8): the staged form (one wave per workgroup, the whole run in LDS)
against the unstaged register form (RLS_STEP_WPB = 4 forces it where 4 runs do not fit LDS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import _abi, graph, ops
dev = torch.device("cuda:0")
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for N, B in ((10000, 131072), (16000, 16384), (20000, 16384), (24000, 16384), (32000, 8192), (40000, 8192), (44000, 4096), (44000, 16384), (52000, 8192), (60000, 8192), (70000, 8192), (80000, 4096), (80000, 16384), (90000, 8192), (100000, 4096), (150000, 4096)):
    g = graph.generate_gnm(N, 2 * N, seed=1)
    dg = ops.DeviceGraph(graph.build_csr(g, num_nodes=N, if_bidirectional=False), dev)
    xa, xb = ops.rand_spins(B, N, 1, dev), torch.empty((B, N), dtype=torch.bool, device=dev)
    obj = ops.maxcut_obj(dg, xa).to(torch.int32)
    act = ops.rand_actions(B, N, seed=1, step=0, device=dev)
    rew = torch.empty(B, dtype=torch.float32, device=dev)
    row = []
    for label, knobs in (("auto", {}), ("unstaged", {"RLS_STEP_NOSTAGE": 1})):
        for k in ("RLS_STEP_WPB", "RLS_STEP_NOSTAGE"): _abi.tuning_unset(k)
        for k, v in knobs.items(): _abi.tuning_set(k, v)
        o = obj.clone()
        L = ops.maxcut_step_launcher(dg, xa, xb, act, o, rew)
        L(); ok = torch.equal(ops.maxcut_obj(dg, xb).to(torch.int32), o)
        us = t_us(L)
        row.append(f"{label}: {us:.0f} us {B * (2 * N + 20) / us / 8e6:.3f}{'' if ok else ' BROKEN'}")
    _abi.tuning_unset("RLS_STEP_WPB"); _abi.tuning_unset("RLS_STEP_NOSTAGE")
    print(f"N={N} B={B}: " + " | ".join(row), flush=True)
    del xa, xb
