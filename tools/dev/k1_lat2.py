# K1 timing on a graph whose stored-edge order is LDS-bank-conflict free (upper bound for edge reordering)
import torch, sys
from rlsolver_amd import ops, graph
dev = torch.device('cuda:0')
N = 2000
kind = sys.argv[1] if len(sys.argv) > 1 else "band"
if kind == "band":
    g = [(i, (i + 1 + k) % N, 1) for k in range(10) for i in range(N)][:19990]
    g = [(min(a, b), max(a, b), w) for a, b, w in g]
else:
    g = graph.generate_gnm(2000, 19990, seed=1)
dg = ops.DeviceGraph(graph.build_csr(g, num_nodes=N, if_bidirectional=False), dev)
for B in (65536, 262144):
    xs = ops.rand_spins(B, N, 1, dev)
    for _ in range(3): ops.maxcut_obj(dg, xs)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.maxcut_obj(dg, xs)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(kind, B, f"{us:.1f} us", f"{B*N/us/1e6:.2f} TB/s")
