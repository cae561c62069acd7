# needs RLS_EXTRA_CFLAGS=-DRLS_PROF build
import ctypes, sys, torch
import rlsolver_amd.build as b
from rlsolver_amd import graph
from rlsolver_amd.envs.env_L2A import EnvMaxcut
lib = ctypes.CDLL(b.LIB_PATH)
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = graph.generate_gnm(2000, 19990, seed=1)
env = EnvMaxcut(mygraph=g, device=dev, num_nodes=2000)
xs = env.generate_xs_randomly(B)
vs = env.calculate_obj_values(xs)
for _ in range(2): env.local_search_inplace(xs.clone(), vs.clone())
torch.cuda.synchronize()
lib.rls_dev_prof_ls(None, 1)
x2, v2 = xs.clone(), vs.clone()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record(); env.local_search_inplace(x2, v2); e.record(); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
lib.rls_dev_prof_ls(out, 0)
n = max(out[7], 1)
print("call us", s.elapsed_time(e) * 1e3, "tiles", n)
for name, i in (("phase0 load+rp", 0), ("phase1 threshold", 1), ("phase2 proposals", 2), ("phase3 sweep", 3), ("phase4 store", 4)):
    print(f"{name:20s} {out[i] / n / 100.0:9.2f} us per tile")
print(f"  of phase 2: mask build {out[5] / n / 100.0:9.2f} us, cut count {out[6] / n / 100.0:9.2f} us per tile (8 rounds)")
