import sys, torch
from rlsolver_amd import graph, ops
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n, m = 2000, 19990
g = ops.DeviceGraph(graph.build_csr(graph.generate_gnm(n, m, 22), num_nodes=n), dev)
x = ops.rand_spins(B, n, 1, dev)
out = torch.empty((B, n), dtype=torch.int32, device=dev)
for _ in range(3):
    ops.maxcut_delta_all(g, x, out=out)
torch.cuda.synchronize()
