# needs a build with RLS_EXTRA_CFLAGS=-DRLS_PROF
import ctypes, sys, torch
from rlsolver_amd import ops, graph, _abi
lib = ctypes.CDLL(_abi.LIB_PATH) if hasattr(_abi, "LIB_PATH") else None
import rlsolver_amd.build as b
lib = ctypes.CDLL(b.LIB_PATH)
dev = torch.device('cuda:0')
g = graph.generate_gnm(2000, 19990, seed=1)
dg = ops.DeviceGraph(graph.build_csr(g, num_nodes=2000, if_bidirectional=False), dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
xs = ops.rand_spins(B, 2000, 1, dev)
for _ in range(3): ops.maxcut_obj(dg, xs)
torch.cuda.synchronize()
lib.rls_dev_prof(None, 1)
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record(); ops.maxcut_obj(dg, xs); e.record(); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
lib.rls_dev_prof(out, 0)
it = out[4]
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 8
print("launch us", s.elapsed_time(e) * 1e3, "iters", it)
# wall_clock64 ticks at 100 MHz
for name, i, n in (("producer load", 0, NW), ("consumer count", 1, NW), ("producer barrier wait", 2, NW), ("consumer barrier wait", 3, NW)):
    print(f"{name:20s} {out[i] / max(it,1) / n / 100.0:8.2f} us per wave per tile")
print(f"edge loop  {out[5] / max(it,1) / 4 / 100.0:8.2f} us per wave per tile")
print(f"reduction  {out[6] / max(it,1) / 4 / 100.0:8.2f} us per wave per tile")
