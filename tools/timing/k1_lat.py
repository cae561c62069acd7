import torch, time, numpy as np
from rlsolver_amd import ops, graph
g = graph.generate_gnm(2000, 19990, seed=1)
csr = graph.build_csr(g, num_nodes=2000, if_bidirectional=False)
dg = ops.DeviceGraph(csr, torch.device('cuda:0'))
for B in (2048, 4096, 16384, 65536, 262144):
    xs = ops.rand_spins(B, 2000, 1, torch.device('cuda:0'))
    for _ in range(3): ops.maxcut_obj(dg, xs)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.maxcut_obj(dg, xs)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(B, f"{us:.1f} us", f"{B*2000/us/1e6:.2f} TB/s")
