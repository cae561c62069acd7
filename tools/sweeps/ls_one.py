import sys, torch
from rlsolver_amd import graph
from rlsolver_amd.envs.env_L2A import EnvMaxcut
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = graph.generate_gnm(2000, 19990, seed=1)
env = EnvMaxcut(mygraph=g, device=dev, num_nodes=2000)
xs = env.generate_xs_randomly(B)
vs = env.calculate_obj_values(xs)
for _ in range(3): env.local_search_inplace(xs.clone(), vs.clone())
torch.cuda.synchronize()
