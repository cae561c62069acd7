"""K5 and local_search_inplace across graph shapes under two settings of the lane-group planner's cost constants
(tools/timing/plan_ab.py): does a setting tuned on G22 hurt elsewhere?  `python tools/timing/plan_shapes.py "150,85,110" "150,50,60"`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import _abi, graph as G, ops
from rlsolver_amd.envs.env_L2A import EnvMaxcut
dev = torch.device("cuda:0")


def t(f, reps):
    f(); f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


sets = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
torus = [(r * 50 + c, r * 50 + (c + 1) % 50, 1) for r in range(40) for c in range(50)] + [(r * 50 + c, ((r + 1) % 40) * 50 + c, 1) for r in range(40) for c in range(50)]
shapes = [("G22 2^16", G.generate_gnm(2000, 19990, 22), 2000, 1 << 16), ("G14 2^14", G.generate_gnm(800, 4694, 14), 800, 1 << 14),
          ("G(2000,4000) 2^14", G.generate_gnm(2000, 4000, 3), 2000, 1 << 14), ("torus 40x50 2^14", torus, 2000, 1 << 14),
          ("BA-2000 m=20 2^14", G.generate_ba(2000, 20, 4), 2000, 1 << 14), ("G(2000,200000) 2^12", G.generate_gnm(2000, 200000, 5), 2000, 1 << 12),
          ("G(5000,20000) 2^15", G.generate_gnm(5000, 20000, 6), 5000, 1 << 15), ("G22 4096", G.generate_gnm(2000, 19990, 22), 2000, 4096)]
for name, mg, n, B in shapes:
    row = []
    for fx, bl, mgc in sets:
        _abi.tuning_set("RLS_PLAN_FIXED", fx); _abi.tuning_set("RLS_PLAN_BLOCK", bl); _abi.tuning_set("RLS_PLAN_MERGE", mgc)
        env = EnvMaxcut(mygraph=mg, device=dev, num_nodes=n)
        torch.manual_seed(0)
        xs = env.generate_xs_randomly(B)
        vs = env.calculate_obj_values(xs)
        k5 = min(t(lambda: ops.maxcut_greedy_sweep(env.graph, xs, vs), 20) for _ in range(2))
        ls = min(t(lambda: env.local_search_inplace(xs, vs), 8) for _ in range(2))
        row.append(f"({fx},{bl},{mgc}): K5 {k5:7.1f} us  LS {ls:8.1f} us  groups {env.graph.num_sweep_groups}")
    print(f"{name:22s} " + "   |   ".join(row), flush=True)
