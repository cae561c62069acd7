"""K7 (level-parallel sampler, bit-packed in place) and K5 (greedy sweep) at the BASELINE sizes: ms per launch.
`python tools/timing/k7_packed.py`."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import graph, ops, ops_mcpg_tsp as mops
from rlsolver_amd.graph import build_csr, generate_gnm
from rlsolver_amd.methods import MCPG as amcpg
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set
dev = torch.device('cuda:0')


def t(f, reps=8):
    f(); f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


n, m, C = 10000, 5, 1 << 18
gb = np.asarray(graph.generate_ba(n, m, seed=5), dtype=np.int64)
data = amcpg.make_data(n, gb[:, 0].copy(), gb[:, 1].copy(), dev)
pk = mops.PackedChains(torch.randint(-2 ** 62, 2 ** 62, (C // 64, n), dtype=torch.int64, device=dev), C)
for nls in (1, 8):
    ms = t(lambda: mops.mcpg_local_search_levels(data.graph, pk, data._lv_ptr, data._lv_data, nls, 1, out=pk))
    print(f"K7 packed in place, BA-1e4 2^18 chains, num_ls {nls}: {ms:.3f} ms")
ms = t(lambda: amcpg.sampler_func_packed(data, pk, 8, 2048, 128))
print(f"sampler_func_packed (K7 x 8 + K8 + pick): {ms:.3f} ms")
for tag, nn, mm, B in (("G22 2^16", 2000, 19990, 1 << 16), ("G70 2^17", 10000, 9999, 1 << 17), ("BA-1e4 2^16", 0, 0, 1 << 16)):
    g = data.graph if nn == 0 else ops.DeviceGraph(build_csr(generate_gnm(nn, mm, 22), num_nodes=nn), dev)
    xs = ops.rand_spins(B, g.num_nodes, 3, dev)
    vs = ops.maxcut_obj(g, xs)
    x2, v2 = xs.clone(), vs.clone()
    us = t(lambda: ops.maxcut_greedy_sweep(g, x2, v2), 20) * 1e3
    print(f"K5 greedy sweep {tag}: {us:.1f} us")
    us = t(lambda: ops.maxcut_delta_all(g, xs), 20) * 1e3
    print(f"K3 delta_all {tag}: {us:.1f} us")
