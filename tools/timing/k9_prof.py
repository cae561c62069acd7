"""Where the packed metro walk's cycles go (dev build -DRLS_K7_PROF): the walker wave's loop vs the window barriers, the producers'
draw windows; BASELINE config #3, one launch of T = 1000 rounds."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import _abi, ops_mcpg_tsp as mops
dev = torch.device('cuda:0')
n, C, T = 10000, 1 << 18, 1000
probs = torch.full((n,), 0.5, device=dev)
pk = mops.PackedChains(torch.randint(-2 ** 62, 2 ** 62, (C // 64, n), dtype=torch.int64, device=dev), C)
acc = torch.zeros((64, T), dtype=torch.int64, device=dev)
f = lambda: mops.mcpg_metro_rounds(pk, probs, T, None, None, 1, None, True, acc)
f(); torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record(); f(); e.record(); torch.cuda.synchronize()
out = np.zeros(2048 * 16 * 6, dtype=np.uint64)
lib = _abi.lib()
lib.rls_dev_k7_prof.argtypes = [ctypes.c_void_p]
assert lib.rls_dev_k7_prof(out.ctypes.data_as(ctypes.c_void_p)) == 0
t = out.reshape(2048, 16, 6)[:, :8, :].astype(np.float64)
print(f"launch {s.elapsed_time(e):.3f} ms for {T} rounds (instrumented)")
tot = t[:, 0, 0].mean()
print(f"walker wave: total {tot:.0f} cycles = {tot / T:.0f} per round; in its loop {t[:,0,3].mean()/tot:.3f}, at the window barrier {t[:,0,1].mean()/tot:.3f}; windows {t[:,0,4].mean():.0f}")
for w in (1, 4, 7):
    print(f"producer wave {w}: producing {t[:,w,3].mean()/t[:,w,0].mean():.3f}, at the window barrier {t[:,w,1].mean()/t[:,w,0].mean():.3f}")
