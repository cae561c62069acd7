"""`local_search_inplace` by waves per tile (RLS_LS_WAVES = 4 | 8, read once per process: run once per setting) over batch sizes and
graphs: ms per call.  `RLS_LS_WAVES=8 python tools/timing/ls_waves.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import graph
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (the library itself reads no environment)
dev = torch.device('cuda:0')


def t(f, reps=6):
    f(); f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


print("RLS_LS_WAVES =", os.environ.get("RLS_LS_WAVES", "(auto)"))
for tag, g, n in (("G22-sized", graph.generate_gnm(2000, 19990, seed=1), 2000), ("G14-sized", graph.generate_gnm(800, 4694, seed=1), 800),
                  ("BA-3000 m=4", graph.generate_ba(3000, 4, seed=1), 3000), ("gnm-5000", graph.generate_gnm(5000, 20000, seed=1), 5000)):
    env = EnvMaxcut(mygraph=g, device=dev, num_nodes=n)
    for B in (1024, 4096, 1 << 14, 1 << 15, 1 << 16, 3 << 15):
        xs = env.generate_xs_randomly(B)
        vs = env.calculate_obj_values(xs)
        ms = t(lambda: env.local_search_inplace(xs, vs))
        print(f"{tag:12s} B={B:6d}: {ms:.3f} ms per local_search_inplace")
