"""metro_sampling_packed at BASELINE config #3 with the walk's draw windows in LDS (RLS_METRO_QG=0: one workgroup per CU at N = 10^4)
vs in global scratch (two per CU), interleaved; same chains either way."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import _abi
from rlsolver_amd.methods import MCPG as amcpg
from rlsolver_amd.ops_mcpg_tsp import PackedChains
dev = torch.device("cuda:0")
for n, C, p in ((10000, 1 << 18, 0.5), (10000, 1 << 18, 0.3), (8000, 1 << 17, 0.5), (2000, 1 << 16, 0.5)):
    M, T = C // 128, n // 10
    torch.manual_seed(0)
    probs = torch.full((n,), p, device=dev)
    kept = PackedChains.pack((torch.rand((n, M), device=dev) < 0.5).float())
    res, outs = {0: [], 1: []}, {}
    for rep in range(3):
        for v in (0, 1):
            _abi.tuning_set("RLS_METRO_QG", v)
            out = PackedChains.empty(n, C, dev)
            amcpg.metro_sampling_packed(probs, kept, T, num_chains=C, out=out, seed=5)
            outs[v] = out.words.clone()
            torch.cuda.synchronize()
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): amcpg.metro_sampling_packed(probs, kept, T, num_chains=C, out=out, seed=5)
            e.record(); torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) / 5)
    assert torch.equal(outs[0], outs[1])
    print(f"metro_sampling_packed N={n} C={C} p={p}: windows in LDS {min(res[0]):.3f} ms, in global scratch {min(res[1]):.3f} ms")
