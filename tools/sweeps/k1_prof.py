# K1 phase breakdown; needs a build with RLS_EXTRA_CFLAGS=-DRLS_PROF (python -m rlsolver_amd.build --force)
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from rlsolver_amd import ops, graph
import rlsolver_amd.build as b
lib = ctypes.CDLL(b.LIB_PATH)
dev = torch.device('cuda:0')
for n, m, B, W in ((2000, 19990, 65536, 4), (10000, 9999, 131072, 8)):
    dg = ops.DeviceGraph(graph.build_csr(graph.generate_gnm(n, m, seed=1), num_nodes=n, if_bidirectional=False), dev)
    xs = ops.rand_spins(B, n, 1, dev)
    for _ in range(3):
        ops.maxcut_obj(dg, xs)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); ops.maxcut_obj(dg, xs); e.record(); torch.cuda.synchronize()
    nw = min(B // 64 * W, 65536)
    out = np.zeros(nw * 5, dtype=np.uint64)
    lib.rls_dev_prof_waves(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(nw))
    t = out.reshape(nw, 5).astype(np.float64) * 0.01          # us (wall_clock64: 100 MHz)
    t -= t[:, 0].min()
    print(f"N={n} B={B}: launch {s.elapsed_time(e) * 1e3:.1f} us; waves {nw}; last end {t[:, 4].max():.1f} us")
    print(f"   start      : min {t[:,0].min():6.1f}  median {np.median(t[:,0]):6.1f}  max {t[:,0].max():6.1f}")
    print(f"   load  done : min {t[:,1].min():6.1f}  median {np.median(t[:,1]):6.1f}  max {t[:,1].max():6.1f}   (duration median {np.median(t[:,1]-t[:,0]):.1f})")
    print(f"   barrier    : duration median {np.median(t[:,2]-t[:,1]):.1f} max {(t[:,2]-t[:,1]).max():.1f}")
    print(f"   count done : min {t[:,3].min():6.1f}  median {np.median(t[:,3]):6.1f}  max {t[:,3].max():6.1f}   (duration median {np.median(t[:,3]-t[:,2]):.1f})")
    print(f"   end        : median {np.median(t[:,4]):6.1f}  max {t[:,4].max():6.1f}   (reduce+store median {np.median(t[:,4]-t[:,3]):.1f})")
