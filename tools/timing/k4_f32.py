import torch
from rlsolver_amd import ops, graph
dev = torch.device('cuda:0')
n, m, B = 2000, 19990, 65536
g = ops.DeviceGraph(graph.build_csr(graph.generate_gnm(n, m, 22), num_nodes=n), dev)
S = 6
slots = [ops.rand_spins(B, n, s, dev).float() for s in range(S)]
obj = ops.maxcut_obj(g, slots[0]).to(torch.int32)
rew = torch.empty(B, dtype=torch.float32, device=dev)
acts = [ops.rand_actions(B, n, 7, s, dev) for s in range(8)]
def run(K):
    for i in range(K):
        ops.maxcut_step(g, slots[i % S], slots[(i + 1) % S], acts[i % 8], obj, rew)
run(5); torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record(); run(60); e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 60 * 1e3
print(f"K4 emit f32 surface: {us:.1f} us per launch, {B/us*1e6:.3g} env-steps/s, {B*(8*n+20)/us/1e6:.2f} TB/s algorithmic (2*4N+20 B per env-step)")
