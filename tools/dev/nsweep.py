# K4 / K1 / K6 bandwidth for row lengths that are not multiples of 16 bytes (same total bytes as the headline)
import torch
from rlsolver_amd import ops, graph
dev = torch.device('cuda:0')
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for N, E in ((2000, 19990), (1000, 5000), (3000, 6000), (5000, 12498), (7000, 17148), (1004, 5000)):
    B = 65536 * 2000 // N
    g = graph.generate_gnm(N, E, seed=1)
    dg = ops.DeviceGraph(graph.build_csr(g, num_nodes=N, if_bidirectional=False), dev)
    xs = [ops.rand_spins(B, N, i, dev) for i in range(4)]
    ys = [torch.empty_like(xs[0]) for _ in range(4)]
    obj = ops.maxcut_obj(dg, xs[0]).to(torch.int32)
    rew = torch.empty(B, dtype=torch.float32, device=dev)
    act = ops.rand_actions(B, N, 7, 0, dev)
    k = [0]
    def step():
        i = k[0] % 4; k[0] += 1
        ops.maxcut_step(dg, xs[i], ys[i], act, obj, rew)
    us4 = t(step)
    us1 = t(lambda: ops.maxcut_obj(dg, xs[0]))
    mask = ops.rand_spins(B, N, 99, dev)
    o64 = ops.maxcut_obj(dg, xs[1])
    us6 = t(lambda: ops.maxcut_propose_accept(dg, xs[1], mask, o64))
    by = B * N
    print(f"N={N:5d} B={B:6d}  K4 {us4:7.1f} us {2*by/us4/1e6:5.2f} TB/s | K1 {us1:7.1f} us {by/us1/1e6:5.2f} TB/s | K6 {us6:7.1f} us {2.5*by/us6/1e6:5.2f} TB/s(2.5N)")
