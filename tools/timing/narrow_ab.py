"""K1 / K5 / K6 / local_search_inplace past the half tile (N > ~40 000): narrow tiles (16 / 8 envs per workgroup) vs one env per wave
on a byte row (RLS_NARROW_TILE=0), 4096 envs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import _abi, ops
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.graph import generate_gnm
dev = torch.device("cuda:0")


def t(f, reps):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


B = 4096
for n, m in ((39936, 80000), (44000, 88000), (80000, 160000), (100000, 200000), (160000, 320000)):
    env = EnvMaxcut(mygraph=generate_gnm(n, m, 7), device=dev, num_nodes=n)
    torch.manual_seed(0)
    xs = env.generate_xs_randomly(B)
    vs = env.calculate_obj_values(xs)
    mask = torch.rand((B, n), device=dev) < 4.0 / n
    row = []
    for narrow in (1, 0):
        _abi.tuning_set("RLS_NARROW_TILE", narrow)
        reps = 5 if narrow else 1
        k1 = t(lambda: ops.maxcut_obj(env.graph, xs), reps)
        x6, v6 = xs.clone(), vs.clone()
        k6 = t(lambda: ops.maxcut_propose_accept(env.graph, x6, mask, v6), reps)
        x5, v5 = xs.clone(), vs.clone()
        k5 = t(lambda: ops.maxcut_greedy_sweep(env.graph, x5, v5), reps)
        x7, v7 = xs.clone(), vs.clone()
        ls = t(lambda: env.local_search_inplace(x7, v7), 2 if narrow else 1)
        row.append(f"{'narrow' if narrow else 'rows  '}: K1 {k1:8.3f}  K6 {k6:8.3f}  K5 {k5:8.3f}  local_search_inplace {ls:9.3f} ms")
    _abi.tuning_unset("RLS_NARROW_TILE")
    print(f"N={n} B={B}:  " + "   |   ".join(row), flush=True)
