"""K1 and K6 (byte mask, 1 % proposals) at the judged shapes -- for A/B builds (e.g. RLS_EXTRA_CFLAGS=-DRLS_TILE_CONTIG).
    python tools/timing/k1k6_ab.py [tag]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import _abi, graph, ops

dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else ""


def t_us(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, N, E, B, gen in (("G22", 2000, 19990, 65536, "gnm"), ("G70", 10000, 9999, 131072, "gnm"), ("ER-2000", 2000, 9995, 65536, "gnm"),
                           ("BA-2000", 2000, 0, 65536, "ba"), ("G14", 800, 4694, 65536, "gnm"), ("N3008", 3008, 30000, 65536, "gnm")):
    g = graph.generate_ba(N, 4, seed=1) if gen == "ba" else graph.generate_gnm(N, E, seed=1)
    dg = ops.DeviceGraph(graph.build_csr(g, num_nodes=N, if_bidirectional=False), dev)
    xs = ops.rand_spins(B, N, 1, dev)
    obj = ops.maxcut_obj(dg, xs)
    k1 = t_us(lambda: ops.maxcut_obj(dg, xs))
    mask = (torch.rand((B, N), device=dev) < 0.004)
    o2 = obj.clone()
    k6 = t_us(lambda: ops.maxcut_propose_accept(dg, xs, mask, o2))
    ok = torch.equal(ops.maxcut_obj(dg, xs), o2)
    print(f"{tag} {name}: K1 {k1:.1f} us {B * (N + 8) / k1 / 8e6:.3f} | K6 byte {k6:.1f} us {B * (2 * N + 16) / k6 / 8e6:.3f} (2N+16){'' if ok else ' K6 PARITY BROKEN'}", flush=True)
