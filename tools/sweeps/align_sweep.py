"""Rows of 16-byte multiples vs rows of 8-byte multiples (the Gset sizes 1000, 3000, 5000, 7000, 9000) at full batches: K4 emit, K1, K6,
K5, K3 -- us per call and bytes-per-us normalised by N."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import ops, _abi; _abi.tuning_from_env()
from rlsolver_amd.graph import build_csr, generate_gnm
dev = torch.device("cuda:0")


def t(f, K=6):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(K): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / K * 1e3


B = 1 << 16
for n in (3000, 3008, 5000, 5008, 7000, 7008, 9000, 9008):
    g = ops.DeviceGraph(build_csr(generate_gnm(n, 3 * n, n), num_nodes=n), dev)
    x = ops.rand_spins(B, n, 1, dev); y = torch.empty_like(x)
    obj = ops.maxcut_obj(g, x).to(torch.int32); rew = torch.empty(B, dtype=torch.float32, device=dev)
    act = ops.rand_actions(B, n, 7, 0, dev)
    k4 = t(lambda: ops.maxcut_step(g, x, y, act, obj, rew))
    k1 = t(lambda: ops.maxcut_obj(g, x))
    mask = torch.rand((B, n), device=dev) < 8.0 / n
    vs = ops.maxcut_obj(g, x)
    k6 = t(lambda: ops.maxcut_propose_accept(g, x, mask, vs))
    k5 = t(lambda: ops.maxcut_greedy_sweep(g, x, vs), 3)
    d = torch.empty((B, n), dtype=torch.int32, device=dev)
    k3 = t(lambda: ops.maxcut_delta_all(g, x, out=d), 3)
    print(f"N={n} (N % 16 = {n % 16}): K4 {k4:7.1f} [{2 * n * B / k4 / 1e6:.2f} TB/s]  K1 {k1:7.1f} [{n * B / k1 / 1e6:.2f}]  K6 {k6:7.1f}  K5 {k5:7.1f}  K3 {k3:7.1f}", flush=True)
