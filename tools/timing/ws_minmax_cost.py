"""The local-search weights pre-pass with and without its batch min / max fold (G22 / 2^16, BA-1e4 / 2^15)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import graph, ops, _abi
from rlsolver_amd.ops import _t
from rlsolver_amd.envs.env_L2A import EnvMaxcut
dev = torch.device('cuda:0')


def t(f, K=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(K): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / K * 1e3


for name, g, n, B in (("G22", graph.generate_gnm(2000, 19990, seed=22), 2000, 1 << 16), ("BA-1e4", graph.generate_ba(10000, 5, seed=5), 10000, 1 << 15),
                      ("G22", graph.generate_gnm(2000, 19990, seed=22), 2000, 4096), ("G22", graph.generate_gnm(2000, 19990, seed=22), 2000, 16384),
                      ("BA-1e4", graph.generate_ba(10000, 5, seed=5), 10000, 4096)):
    env = EnvMaxcut(mygraph=g, device=dev, num_nodes=n)
    xs = env.generate_xs_randomly(B)
    dt = ops.ls_weight_dtype(env.graph, 1)
    P = (n + 15) // 16 * 16
    ws = torch.empty((B, P), dtype=dt, device=dev)
    mm = torch.empty((2, n), dtype=torch.int32, device=dev)
    for seeds in (0, 16):
        _abi.tuning_set("RLS_NS_PARK", seeds)
        print(f"  seeds={seeds}: {t(lambda: _t.maxcut_ls_weights(env.graph.handle, xs, 1, ws, mm)):.1f} us", flush=True)
    _abi.tuning_unset("RLS_NS_PARK")
    a = t(lambda: _t.maxcut_ls_weights(env.graph.handle, xs, 1, ws, mm))
    b = t(lambda: _t.maxcut_ls_weights(env.graph.handle, xs, 1, ws, None))
    k2 = t(lambda: ops.maxcut_node_cutdeg(env.graph, xs))
    print(f"{name} B={B}: weights with min/max {a:.1f} us, without {b:.1f} us; K2 (int64 out) {k2:.1f} us", flush=True)
