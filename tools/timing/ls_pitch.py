"""local_search_inplace at BASELINE config #2 by the row pitch of the padded weights (ops.LS_PITCH_BYTES: 16 vs 128), interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import ops
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.graph import generate_gnm
dev = torch.device("cuda:0")
for n, m, B in ((2000, 19990, 1 << 16), (10000, 9999, 1 << 16), (2000, 19990, 4096)):
    env = EnvMaxcut(mygraph=generate_gnm(n, m, 22), device=dev, num_nodes=n)
    torch.manual_seed(0)
    xs = env.generate_xs_randomly(B)
    vs = env.calculate_obj_values(xs)
    res = {16: [], 128: []}
    for rep in range(3):
        for pb in (16, 128):
            ops.LS_PITCH_BYTES = pb
            for _ in range(2):
                env.local_search_inplace(xs, vs)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                env.local_search_inplace(xs, vs)
            e1.record(); torch.cuda.synchronize()
            res[pb].append(e0.elapsed_time(e1) / 10)
    print(f"N={n} B={B}: local_search_inplace pitch 16 B: {min(res[16]):.3f} ms, pitch 128 B: {min(res[128]):.3f} ms")
