"""Every batched entry point at batch sizes from 1 to 16 384: time per call.  Looks for forms that are slow at SMALL batches
(a per-tile cost that does not shrink with the batch), the way the lane = env weights kernel was.
`python tools/sweeps/batch_sweep.py [g22|g14|ba1e4]`."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import ops, ops_mcpg_tsp as mops
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.envs.env_PPO import EnvMaxcut as Gym
from rlsolver_amd.envs.spinsystem import SpinSystem
from rlsolver_amd.graph import build_csr, generate_ba, generate_gnm, generate_tsp_coords, tsp_tables
from rlsolver_amd.methods import MCPG as amcpg

dev = torch.device("cuda:0")
BS = (1, 64, 256, 1024, 4096, 16384)


def t_us(f, n=8):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def row(name, make):
    out = []
    for B in BS:
        try:
            out.append(f"{t_us(make(B)):8.0f}")
        except Exception as e:   # noqa
            out.append(f"{'ERR':>8}")
    flag = ""
    v = [float(x) for x in out if x.strip() != "ERR"]
    if len(v) == len(BS) and (v[0] > 2.5 * v[4] or v[2] > 2.5 * v[4] or v[3] > 2.0 * v[4]):
        flag = "   <-- small batches slower than 4096"
    print(f"{name:44s}" + "".join(out) + flag, flush=True)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "g22"
    n, mg = {"g22": (2000, generate_gnm(2000, 19990, 22)), "g14": (800, generate_gnm(800, 4694, 14)),
             "ba1e4": (10000, generate_ba(10000, 5, 5))}[which]
    env = EnvMaxcut(mygraph=mg, device=dev, num_nodes=n)
    g = env.graph
    print(f"{which}: N={n} E={len(mg)}; us per call at B =" + "".join(f"{b:8d}" for b in BS))
    xs_of = lambda B: (torch.rand(B, n, device=dev) < 0.5)
    def mk_obj(B):
        x = xs_of(B); return lambda: ops.maxcut_obj(g, x)
    def mk_cutdeg(B):
        x = xs_of(B); return lambda: ops.maxcut_node_cutdeg(g, x)
    def mk_delta(B):
        x = xs_of(B); o = torch.empty((B, n), dtype=torch.int32, device=dev); return lambda: ops.maxcut_delta_all(g, x, out=o)
    def mk_sweep(B):
        x = xs_of(B); v = ops.maxcut_obj(g, x); return lambda: ops.maxcut_greedy_sweep(g, x, v)
    def mk_prop(B):
        x = xs_of(B); v = ops.maxcut_obj(g, x); m = torch.rand(B, n, device=dev) < 0.004; return lambda: ops.maxcut_propose_accept(g, x, m, v)
    def mk_ls(B):
        x = xs_of(B); v = ops.maxcut_obj(g, x); return lambda: env.local_search_inplace(x, v, num_iters=8, num_spin=8)
    def mk_lsw(B):
        x = xs_of(B); return lambda: ops.maxcut_ls_weights(g, x, 1)
    def mk_sel(B):
        a, b = xs_of(B), xs_of(B); va, vb = ops.maxcut_obj(g, a), ops.maxcut_obj(g, b); return lambda: ops.select_better_rows(a, va, b, vb)
    def mk_rand(B):
        return lambda: ops.rand_spins(B, n, 5, dev)
    def mk_gym(B, out):
        e = Gym(types.SimpleNamespace(num_nodes=n, num_envs=B, num_steps=10 ** 9), mygraph=mg, device=dev)
        e.reset(); a = torch.randint(0, n, (B,), device=dev)
        slot = torch.empty((B, n), dtype=torch.float32, device=dev) if out else None
        return (lambda: e.step(a, out=slot)) if out else (lambda: e.step(a))
    def mk_spin(B, what):
        e = SpinSystem(mg, n, B, max_steps=10 ** 6, device=dev, include_adjacency=False)
        a = torch.randint(0, n, (B,), device=dev)
        return {"step": lambda: e.step(a), "reset": lambda: e.reset()}[what]
    row("K1 maxcut_obj", mk_obj); row("K2 node_cutdeg", mk_cutdeg); row("K3 delta_all", mk_delta)
    row("K5 greedy_sweep", mk_sweep); row("K6 propose_accept", mk_prop); row("ls_weights", mk_lsw)
    row("local_search_inplace", mk_ls); row("K10 select_better_rows", mk_sel); row("K14 rand_spins", mk_rand)
    row("gym step in place", lambda B: mk_gym(B, False)); row("gym step(out=slot)", lambda B: mk_gym(B, True))
    if n <= 2000:
        row("spin step + observation (rows only)", lambda B: mk_spin(B, "step")); row("spin reset", lambda B: mk_spin(B, "reset"))
    arr = np.asarray(mg, dtype=np.int64)
    data = amcpg.make_data(n, arr[:, 0], arr[:, 1], dev)
    probs = torch.full((n,), 0.5, device=dev)
    def mk_sampler(C):
        C = max(C, 128) // 128 * 128
        x = (torch.rand((n, C), device=dev) < 0.5).float(); return lambda: amcpg.sampler_func(data, x, 2, C // 128, 128, dev)
    def mk_metro(C):
        x = (torch.rand((n, C), device=dev) < 0.5).float(); return lambda: amcpg.metro_sampling(probs, x, max(1, n // 100), dev)
    row("MCPG sampler_func (num_ls=2; C >= 128)", mk_sampler); row("MCPG metro_sampling (T = N / 100)", mk_metro)
    if which == "g22":
        N = 100
        dist, near, rnd = tsp_tables(generate_tsp_coords(N, seed=1), K=20)
        D = torch.from_numpy(dist).to(dev)
        def mk_tour(B):
            t = mops.rand_perms(B, N, 3, dev); return lambda: mops.tsp_tour_length(D, t)
        row("K12 tsp_tour_length (N=100)", mk_tour)
        Q = torch.randn(1000, 1000, device=dev).round()
        Q = Q + Q.T
        from rlsolver_amd.methods import MCPG_qubo as mq
        def mk_qubo(C):
            x = (torch.rand((1000, C), device=dev) < 0.5).float(); return lambda: mq.qubo_local_search_value(Q, x, 1, False)
        row("K11 qubo_local_search_value (n=1000)", mk_qubo)


if __name__ == "__main__":
    main()
