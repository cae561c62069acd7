"""The lane-group planner's cost constants (instructions per group / per block of 8 rounds / per cross-lane merge: rls_host.cpp,
plan_lane_groups) against the kernels as they are now: K7 at BASELINE config #3 and K5 at G22 / G70 size, the tables rebuilt
under each setting.  `python tools/timing/plan_ab.py "150,85,110" "110,50,130" ...`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import _abi, graph, ops, ops_mcpg_tsp as mops
from rlsolver_amd.graph import build_csr, generate_gnm
from rlsolver_amd.methods import MCPG as amcpg
dev = torch.device("cuda:0")


def t(f, reps):
    f(); f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


sets = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(150, 85, 110)]
gb = np.asarray(graph.generate_ba(10000, 5, seed=5), dtype=np.int64)
C = 1 << 18
pk0 = mops.PackedChains(torch.randint(-2 ** 62, 2 ** 62, (C // 64, 10000), dtype=torch.int64, device=dev), C)
for rep in range(2):
    for fx, bl, mg in sets:
        _abi.tuning_set("RLS_PLAN_FIXED", fx); _abi.tuning_set("RLS_PLAN_BLOCK", bl); _abi.tuning_set("RLS_PLAN_MERGE", mg)
        data = amcpg.make_data(10000, gb[:, 0].copy(), gb[:, 1].copy(), dev)
        pk = pk0.clone()
        k7 = t(lambda: mops.mcpg_local_search_levels(data.graph, pk, data._lv_ptr, data._lv_data, 8, 1, out=pk), 5)
        out = [f"K7 BA-1e4 2^18: {k7:.3f} ms (groups {data._lv_ptr.numel() - 1})"]
        for tag, nn, mm, B in (("G22 2^16", 2000, 19990, 1 << 16), ("G70 2^17", 10000, 9999, 1 << 17), ("BA-1e4 2^16", 0, 0, 1 << 16)):
            g = data.graph if nn == 0 else ops.DeviceGraph(build_csr(generate_gnm(nn, mm, 22), num_nodes=nn), dev)
            xs = ops.rand_spins(B, g.num_nodes, 3, dev)
            vs = ops.maxcut_obj(g, xs)
            out.append(f"K5 {tag}: {t(lambda: ops.maxcut_greedy_sweep(g, xs, vs), 20) * 1e3:.1f} us (groups {g.num_sweep_groups})")
        print(f"plan ({fx}, {bl}, {mg}):  " + ";  ".join(out), flush=True)
