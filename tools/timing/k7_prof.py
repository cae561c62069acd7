"""Where a K7 wave's cycles go (dev build: RLS_EXTRA_CFLAGS=-DRLS_K7_PROF python -m rlsolver_amd.build): per wave, cycles at the level
barriers / waiting for a group's header / inside groups, at BASELINE config #3.  `python tools/timing/k7_prof.py [RLS_K7_EARLY=0]`."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import _abi, graph, ops_mcpg_tsp as mops
from rlsolver_amd.methods import MCPG as amcpg
for a in sys.argv[1:]:
    if "=" in a:
        _abi.tuning_set(a.split("=")[0], int(a.split("=")[1]))
dev = torch.device('cuda:0')
n, C = 10000, 1 << 18
gb = np.asarray(graph.generate_ba(n, 5, seed=5), dtype=np.int64)
data = amcpg.make_data(n, gb[:, 0].copy(), gb[:, 1].copy(), dev)
pk = mops.PackedChains(torch.randint(-2 ** 62, 2 ** 62, (C // 64, n), dtype=torch.int64, device=dev), C)
f = lambda: mops.mcpg_local_search_levels(data.graph, pk, data._lv_ptr, data._lv_data, 8, 1, out=pk)
f(); f(); torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record(); f(); e.record(); torch.cuda.synchronize()
out = np.zeros(2048 * 16 * 6, dtype=np.uint64)
lib = _abi.lib()
lib.rls_dev_k7_prof.argtypes = [ctypes.c_void_p]
assert lib.rls_dev_k7_prof(out.ctypes.data_as(ctypes.c_void_p)) == 0
t = out.reshape(2048, 16, 6)[:, :8, :].astype(np.float64)
print(f"launch {s.elapsed_time(e):.3f} ms (instrumented); knobs {sys.argv[1:]}")
tot = t[:, :, 0].mean()
print(f"per wave, mean over {t.shape[0]} workgroups x 8 waves: total {tot:.0f} cycles; barrier {t[:,:,1].mean()/tot:.3f}, header wait "
      f"{t[:,:,2].mean()/tot:.3f}, in groups {t[:,:,3].mean()/tot:.3f} (hub groups {t[:,:,5].mean()/tot:.3f}); groups per wave {t[:,:,4].mean():.0f}; "
      f"cycles per group {t[:,:,3].sum()/t[:,:,4].sum():.0f}, header wait per group {t[:,:,2].sum()/t[:,:,4].sum():.0f}")
for w in range(8):
    print(f"  wave {w}: barrier {t[:,w,1].mean()/tot:.3f} header {t[:,w,2].mean()/tot:.3f} groups {t[:,w,3].mean()/tot:.3f} n {t[:,w,4].mean():.0f}")
