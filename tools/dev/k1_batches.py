import sys, torch
sys.path.insert(0, ".")
from rlsolver_amd import graph, ops
dev = torch.device("cuda:0")
n, m = 2000, 19990
g = ops.DeviceGraph(graph.build_csr(graph.generate_gnm(n, m, 22), num_nodes=n), dev)
for B in (98304, 131072, 262144):
    x = ops.rand_spins(B, n, 1, dev)
    out = torch.empty(B, dtype=torch.int64, device=dev)
    for _ in range(3): ops.maxcut_obj(g, x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20): ops.maxcut_obj(g, x, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("B=%d: %.1f us  frac %.3f" % (B, us, B * (n + 8) / us / 1e6 / 8))
