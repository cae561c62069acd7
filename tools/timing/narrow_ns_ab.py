"""K2 / K3 / the local-search weights on narrow tiles (16 / 8 envs per workgroup, bit-sliced lane = node) vs what they took before:
past the half tile (N > 40 960) the element-parallel kernels; below it, the half / 64-env tiles (forced narrow: RLS_NARROW_TILE=2 | 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import _abi, ops
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.graph import generate_gnm
dev = torch.device("cuda:0")


def t(f, reps=5):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for n, m, Bs in ((44000, 88000, (256, 4096)), (100000, 200000, (256, 4096)), (160000, 320000, (4096,)), (10000, 9999, (256, 1024, 4096)),
                 (20000, 40000, (256, 1024, 4096)), (2000, 19990, (256, 1024, 2048, 4096, 8192))):
    env = EnvMaxcut(mygraph=generate_gnm(n, m, 7), device=dev, num_nodes=n)
    g = env.graph
    for B in Bs:
        torch.manual_seed(0)
        xs = env.generate_xs_randomly(B)
        d = torch.empty((B, n), dtype=torch.int32, device=dev)
        row = []
        for name, nk in (("auto", 1), ("off", 0), ("n16", 2), ("n8", 3)):
            _abi.tuning_set("RLS_NARROW_TILE", nk)
            if nk >= 2: _abi.tuning_set("RLS_NODE_STATS_MIN_B", 0)
            try:
                form = ops.node_stats_form(g, B, True)
                k3 = t(lambda: ops.maxcut_delta_all(g, xs, out=d))
                k2 = t(lambda: ops.maxcut_node_cutdeg(g, xs))
                kw = t(lambda: ops.maxcut_ls_weights(g, xs, 4))
            finally:
                _abi.tuning_unset("RLS_NARROW_TILE"); _abi.tuning_unset("RLS_NODE_STATS_MIN_B")
            row.append(f"{name}[{form}] K3 {k3:8.1f} K2 {k2:8.1f} ws {kw:8.1f}")
        print(f"N={n} B={B}: " + " | ".join(row), flush=True)
