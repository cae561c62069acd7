#!/usr/bin/env python3
"""A/B the step kernel's development knobs in one process (interleaved rounds; guide rule 24)."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlsolver_amd import _abi, ops
from rlsolver_amd.graph import build_csr, generate_gnm

n, m, B, S = (int(os.environ.get(k, d)) for k, d in (("SW_N", 2000), ("SW_M", 19990), ("SW_B", 65536), ("SW_S", 8)))
F32 = os.environ.get("SW_DT", "u8") == "f32"
dev = torch.device("cuda:0")
g = ops.DeviceGraph(build_csr(generate_gnm(n, m, 22), num_nodes=n), dev)
x = ops.rand_spins(B, n, 1, dev)
if F32:
    x = x.float()
slots = [torch.empty_like(x) for _ in range(S)]
slots[0].copy_(x)
obj = ops.maxcut_obj(g, x).to(torch.int32)
reward = torch.empty(B, dtype=torch.float32, device=dev)
acts = [ops.rand_actions(B, n, 7, s, dev) for s in range(16)]


def timeit(iters=int(os.environ.get('SW_ITERS', 200))):
    for i in range(5):
        ops.maxcut_step(g, slots[i % S], slots[(i + 1) % S], acts[i % 16], obj, reward)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        ops.maxcut_step(g, slots[i % S], slots[(i + 1) % S], acts[i % 16], obj, reward)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


configs = [dict(RLS_STEP_NTS=str(md), RLS_STEP_EPW=str(e), RLS_STEP_WPB=str(w), RLS_STEP_PERSIST=str(ps), RLS_STEP_CHASE=str(ch),
                RLS_STEP_ALIGN=str(al))   # nontemporal stores 0 | 1 (-1 = the launcher's rule); ALIGN: instruction boundaries on cache lines
           for md, e, w, ps, ch, al in itertools.product(os.environ.get("SW_NTS", "0,1").split(","),
                                                     os.environ.get("SW_EPW", "1,2,4,8").split(","),
                                                     os.environ.get("SW_WPB", "1,2,4").split(","),
                                                     os.environ.get("SW_PERSIST", "0").split(","),
                                                     os.environ.get("SW_CHASE", "0").split(","),
                                                     os.environ.get("SW_ALIGN", "1").split(","))]
res = {i: [] for i in range(len(configs))}
for rep in range(3):
    for i, c in enumerate(configs):
        for k_, v_ in c.items():
            _abi.tuning_set(k_, int(v_))
        try:
            res[i].append(timeit())
        except Exception as ex:
            res[i].append(float("nan"))
by = B * (2 * n * (4 if F32 else 1) + 20)
for i, c in enumerate(configs):
    t = min(res[i])
    print(f"{c}: best {t*1e6:7.1f} us  med {sorted(res[i])[1]*1e6:7.1f} us  {by/t/8e12*100:5.1f}% of 8 TB/s")
