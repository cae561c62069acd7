"""local_search_inplace through the round kernels at G22 / 2^16 (force_ls_rounds): run under `rocprofv3 --kernel-trace --stats` to see
the mask kernel (all 8 rounds' draws in one launch, lane = node) beside the fused kernel's per-round slope."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import graph
from rlsolver_amd.envs.env_L2A import EnvMaxcut
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = EnvMaxcut(mygraph=graph.generate_gnm(2000, 19990, seed=22), device=dev, num_nodes=2000)
env.force_ls_rounds = True
xs = env.generate_xs_randomly(B); vs = env.calculate_obj_values(xs)
for _ in range(6):
    env.local_search_inplace(xs, vs)
torch.cuda.synchronize()
