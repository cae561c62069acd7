import sys, os, torch
from rlsolver_amd import ops
from rlsolver_amd.graph import build_csr, generate_gnm
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (the library itself reads no environment)
dev = torch.device("cuda:0")
for tag, n, m, B in (("G22 2^16", 2000, 19990, 1 << 16), ("G70 2^17", 10000, 9999, 1 << 17)):
    g = ops.DeviceGraph(build_csr(generate_gnm(n, m, 22), num_nodes=n), dev)
    xs = [ops.rand_spins(B, n, s, dev) for s in range(4)]
    out = torch.empty(B, dtype=torch.int64, device=dev)
    for i in range(5): ops.maxcut_obj(g, xs[i % 4], out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for i in range(40): ops.maxcut_obj(g, xs[i % 4], out=out)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("RLS_K1_LDS_KB", "-"), tag, "K1 %.1f us" % (e0.elapsed_time(e1) / 40 * 1e3))
