import sys, torch, numpy as np
from rlsolver_amd import graph
from rlsolver_amd.methods import MCPG as amcpg
from rlsolver_amd import ops_mcpg_tsp as mops
dev = torch.device('cuda:0')
n, m = 10000, 5
gb = np.asarray(graph.generate_ba(n, m, seed=5), dtype=np.int64)
ei = gb[:, :2].T.copy()
data = amcpg.make_data(n, ei[0], ei[1], dev)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
xs = (torch.rand((n, C), device=dev) < 0.5).float()
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for nls in (0, 1, 2, 8):
    ms = t(lambda: mops.mcpg_local_search_levels(data.graph, xs, data._lv_ptr, data._lv_data, nls, 1))
    print("levels kernel num_ls", nls, f"{ms:.3f} ms")
for nls in (0, 1, 8):
    ms = t(lambda: mops.mcpg_local_search(data.graph, xs, data._order_i32, nls, None, 1, visit_stream=data._visit_stream))
    print("stream kernel num_ls", nls, f"{ms:.3f} ms")
