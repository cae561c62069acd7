"""K3 / K2: the 64-env kernel's row staging (RLS_NS_ROWS) against its per-group stores and the half tiles (RLS_NS_TILE32)."""
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
for name, N, E, B, gen in (("G22", 2000, 19990, 65536, "gnm"), ("G70", 10000, 9999, 131072, "gnm"), ("BA-1e4", 10000, 0, 65536, "ba"),
                           ("G14", 800, 4694, 65536, "gnm"), ("N3008", 3008, 30000, 32768, "gnm")):
    g = graph.generate_ba(N, 5, seed=5) if gen == "ba" else graph.generate_gnm(N, E, seed=1)
    dg = ops.DeviceGraph(graph.build_csr(g, num_nodes=N, if_bidirectional=False), dev)
    xs = ops.rand_spins(B, N, 1, dev)
    row, ref3, ref2 = [], None, None
    for label, t32, rows in (("auto", None, None), ("tile64 groups", 0, 0), ("tile64 rows", 0, 1), ("half groups", 1, 0), ("half rows", 1, 1)):
        for k, v in (("RLS_NS_TILE32", t32), ("RLS_NS_ROWS", rows)):
            _abi.tuning_unset(k) if v is None else _abi.tuning_set(k, v)
        try:
            d3 = ops.maxcut_delta_all(dg, xs)
            d2 = ops.maxcut_node_cutdeg(dg, xs)
        except Exception as e:
            row.append(f"{label}: {type(e).__name__}")
            continue
        if ref3 is None:
            ref3, ref2 = d3, d2
        ok = torch.equal(d3, ref3) and torch.equal(d2, ref2)
        k3 = t_us(lambda: ops.maxcut_delta_all(dg, xs))
        k2 = t_us(lambda: ops.maxcut_node_cutdeg(dg, xs), 10)
        row.append(f"{label}: K3 {k3:.0f} us {B * 5 * N / k3 / 8e6:.3f} | K2 {k2:.0f} us {B * 9 * N / k2 / 8e6:.3f}{'' if ok else ' MISMATCH'}")
        del d3, d2
    _abi.tuning_unset("RLS_NS_TILE32"); _abi.tuning_unset("RLS_NS_ROWS")
    print(f"{name} N={N} B={B}: " + " || ".join(row), flush=True)
    del xs, ref3, ref2
    torch.cuda.empty_cache()
