"""local_search_inplace at the reference's batch (4096 envs, G22-sized): run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import graph
from rlsolver_amd.envs.env_L2A import EnvMaxcut
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = EnvMaxcut(mygraph=graph.generate_gnm(2000, 19990, seed=22), device=dev, num_nodes=2000)
xs = env.generate_xs_randomly(B); vs = env.calculate_obj_values(xs)
for _ in range(20):
    env.local_search_inplace(xs, vs)
torch.cuda.synchronize()
