import sys, torch
from rlsolver_amd import ops, graph
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device('cuda:0')
g = graph.generate_gnm(2000, 19990, seed=1)
dg = ops.DeviceGraph(graph.build_csr(g, num_nodes=2000, if_bidirectional=False), dev)
xs = ops.rand_spins(B, 2000, 1, dev)
for _ in range(5): ops.maxcut_obj(dg, xs)
torch.cuda.synchronize()
