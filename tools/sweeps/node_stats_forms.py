"""K2 / K3 at small batches and on weighted graphs: time per call under the dev knobs RLS_NODE_STATS_MIN_B / RLS_NODE_STATS_NO_TILE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import ops
from rlsolver_amd.graph import build_csr, generate_gnm
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (the library itself reads no environment)
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
for name, n, m, weighted in (("G22", 2000, 19990, False), ("G22 +-1", 2000, 19990, True), ("G70", 10000, 9999, False), ("G70 +-1", 10000, 9999, True)):
    mg = generate_gnm(n, m, 22)
    if weighted:
        mg = [(a, b, int(rng.choice([-1, 1]))) for a, b, _ in mg]
    g = ops.DeviceGraph(build_csr(mg, n, False), dev, use_weights=weighted)
    for kname, f in (("K3", lambda xs, out: ops.maxcut_delta_all(g, xs, out=out)), ("K2", None)):
        if f is None:
            continue
        row = []
        for B in (64, 256, 1024, 2048, 4096, 16384):
            xs = torch.rand(B, n, device=dev) < 0.5
            out = torch.empty((B, n), dtype=torch.int32, device=dev)
            for _ in range(3): f(xs, out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f(xs, out)
            e1.record(); torch.cuda.synchronize()
            row.append(f"B={B}: {e0.elapsed_time(e1) * 100:.0f}")
        print(f"min_b={os.environ.get('RLS_NODE_STATS_MIN_B', '2048')} no_tile={'RLS_NODE_STATS_NO_TILE' in os.environ}", name, kname, " ".join(row), "us")
