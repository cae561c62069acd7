"""K7 (level-parallel sampler, bit-packed in place) at BASELINE config #3, A/B over tuning knobs in ONE process, interleaved
(guide rule: never compare across boxes).  `python tools/timing/k7_ab.py RLS_K7_EARLY=0,1 [RLS_K7_WAVES=8,16]`."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import _abi, graph, ops_mcpg_tsp as mops
from rlsolver_amd.methods import MCPG as amcpg
dev = torch.device('cuda:0')
axes = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(",")]) for a in sys.argv[1:] if "=" in a]
n, m, C = int(os.environ.get("K7_N", 10000)), 5, int(os.environ.get("K7_C", 1 << 18))
gb = np.asarray(graph.generate_ba(n, m, seed=5), dtype=np.int64)
data = amcpg.make_data(n, gb[:, 0].copy(), gb[:, 1].copy(), dev)
pk0 = mops.PackedChains(torch.randint(-2 ** 62, 2 ** 62, (C // 64, n), dtype=torch.int64, device=dev), C)


def run(reps=6):
    pk = pk0.clone()
    f = lambda: mops.mcpg_local_search_levels(data.graph, pk, data._lv_ptr, data._lv_data, 8, 1, out=pk)
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, pk


configs = [dict(zip([a[0] for a in axes], vals)) for vals in itertools.product(*[a[1] for a in axes])] or [{}]
res = {i: [] for i in range(len(configs))}
ref = None
for rep in range(3):
    for i, c in enumerate(configs):
        _abi.tuning_unset()
        for k, v in c.items():
            _abi.tuning_set(k, v)
        ms, pk = run()
        res[i].append(ms)
        if rep == 0:      # every variant computes the same chains (one seed)
            if ref is None: ref = pk.words.clone()
            else: assert torch.equal(ref, pk.words), f"variant {c} changed the result"
for i, c in enumerate(configs):
    print(f"K7 BA-{n} m=5, {C} chains, num_ls 8  {c}:  min {min(res[i]):.3f} ms  all {[round(v, 3) for v in res[i]]}")
