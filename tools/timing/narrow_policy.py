"""Which tile for which batch: K1 / K6 / K5 on the launcher's choice (64-env or half tiles) vs 16- and 8-env narrow tiles forced
(RLS_NARROW_TILE = 2 / 3), by graph size and batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import _abi, ops
from rlsolver_amd.graph import build_csr, generate_gnm
dev = torch.device("cuda:0")


def t(f, reps=20):
    f(); f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for n, m in ((800, 4694), (2000, 19990), (10000, 9999), (20000, 40000), (39936, 80000)):
    g = ops.DeviceGraph(build_csr(generate_gnm(n, m, 7), num_nodes=n), dev)
    for B in (tuple(int(b) for b in os.environ["NP_BATCHES"].split(",")) if "NP_BATCHES" in os.environ else (256, 1024, 4096, 16384, 65536)):
        if B * n > 3e9: continue
        xs = ops.rand_spins(B, n, 1, dev)
        vs = ops.maxcut_obj(g, xs)
        mask = torch.rand((B, n), device=dev) < 4.0 / n
        row = []
        for k in (1, 2, 3):
            _abi.tuning_set("RLS_NARROW_TILE", k)
            k1 = t(lambda: ops.maxcut_obj(g, xs))
            x6, v6 = xs.clone(), vs.clone()
            k6 = t(lambda: ops.maxcut_propose_accept(g, x6, mask, v6))
            x5, v5 = xs.clone(), vs.clone()
            k5 = t(lambda: ops.maxcut_greedy_sweep(g, x5, v5))
            row.append(f"{['', 'auto', '16-env', '8-env'][k]}: {k1:7.1f} {k6:7.1f} {k5:8.1f}")
        _abi.tuning_unset("RLS_NARROW_TILE")
        print(f"N={n:6d} B={B:6d}  K1 / K6 / K5 us   " + "  |  ".join(row), flush=True)
