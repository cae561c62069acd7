"""Differential fuzz of local_search_inplace with recorded noise: the fused kernel against the decomposed K2 / K6 / K5 path on
random graphs (incl. hubs, where the two take different kernels), and both against the reference-shaped numpy oracle on small
ones; random batch sizes around the tile, num_iters 0-6, num_spin 1-8, both adjacency forms.  Where the rows are 16-byte
multiples, additionally the threshold / proposal-round kernels against the fused kernel with in-kernel draws (same seed), at a
random env_offset, and -- the decomposed path now draws the kernels' own normals -- the decomposed path too; every few
configurations the batch is also run as two shards (env_offset, env_offset + B / 2) that must equal the whole batch.
`python tools/fuzz/fuzz_ls.py [seconds] [seed]`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_np as onp
from rlsolver_amd import graph as G
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (forced forms)
from rlsolver_amd.envs.env_L2A import EnvMaxcut

DEV = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = nrounds = 0
while time.time() < t_end:
    kind = rng.choice(["gnm", "ba", "hub", "tiny"])
    n = int(rng.randint(10, 64)) if kind == "tiny" else int(rng.choice([rng.randint(64, 400), rng.randint(400, 2500)]))
    if kind != "tiny" and rng.rand() < 0.5:
        n = max(64, n // 16 * 16)
    if kind in ("gnm", "tiny"):
        garr = np.asarray(G.generate_gnm(n, int(rng.randint(n, min(n * (n - 1) // 2, n * 8))), int(rng.randint(1 << 30))), dtype=np.int64)
    elif kind == "ba":
        garr = np.asarray(G.generate_ba(n, int(rng.randint(1, 7)), int(rng.randint(1 << 30))), dtype=np.int64)
    else:
        e = {(0, j) for j in rng.choice(np.arange(1, n), min(n - 1, int(rng.choice([100, 300, 700]))), replace=False).tolist()}
        for _ in range(2 * n):
            a, b = rng.randint(0, n, 2)
            if a != b:
                e.add((min(a, b), max(a, b)))
        garr = np.asarray([(a, b, 1) for a, b in sorted(e)], dtype=np.int64)
    bidir = bool(rng.rand() < 0.4)
    B = int(rng.choice([1, 9, 64, 65, 130, 300]))
    num_iters, num_spin = int(rng.randint(0, 7)), int(rng.randint(1, 9))
    num_spin = min(num_spin, n - 1)
    tag = f"it={it} kind={kind} n={n} E={len(garr)} bidir={bidir} B={B} iters={num_iters} spin={num_spin}"
    if "-v" in sys.argv:
        print(tag, flush=True)
    env = EnvMaxcut(mygraph=[tuple(int(v) for v in r) for r in garr], device=DEV, if_bidirectional=bidir, num_nodes=n)
    xs0 = torch.from_numpy(rng.randint(0, 2, size=(B, n)).astype(bool)).to(DEV)
    noise = torch.from_numpy(rng.randn(num_iters + 2, B, n).astype(np.float32)).to(DEV)
    outs = []
    for fused in (True, False):
        env.fused_local_search = fused
        xs = xs0.clone()
        gx, gv = env.local_search_inplace(xs, torch.empty(()), num_iters=num_iters, num_spin=num_spin, noise_std=0.3, noise=noise)
        outs.append((gx.clone(), gv.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "fused vs decomposed " + tag
    assert np.array_equal(outs[0][1].cpu().numpy(), onp.maxcut_obj(outs[0][0].cpu().numpy(), garr, bidir)), "objective " + tag
    if n <= 64 and B <= 65:
        wx, wv = onp.local_search_inplace(xs0.cpu().numpy(), garr, n, bidir, noise.cpu().numpy(), num_iters=num_iters, num_spin=num_spin)
        assert np.array_equal(outs[0][0].cpu().numpy(), wx) and np.array_equal(outs[0][1].cpu().numpy(), wv), "vs oracle " + tag
    from rlsolver_amd import ops
    env.fused_local_search = True
    if n % 16 == 0 and ops.local_search_fusable(env.graph, num_spin, B) and ops.ls_rounds_supported(env.graph, num_spin):
        first = bool(rng.rand() < 0.5)
        if num_iters > 0 or not first:
            off = int(rng.choice([0, 64, 1000003, 1 << 33]))
            env.set_shard(off)

            def run(e, x_in, form):
                e.fused_local_search = form != "decomposed"
                e.force_ls_rounds, e.force_ls_fused = form == "rounds", form == "fused"
                torch.manual_seed(1000 + it)
                xs, vs = x_in.clone(), e.calculate_obj_values(x_in)
                e.local_search_pipeline(xs, vs, weight_mult=2 if first else 1, num_iters=num_iters, num_spin=num_spin, noise_std=0.3,
                                        noise=None, first_draw_proposes=first)
                return xs, vs
            res = [run(env, xs0, form) for form in ("fused", "rounds", "decomposed")]
            for k in (1, 2):
                assert torch.equal(res[0][0], res[k][0]) and torch.equal(res[0][1], res[k][1]), ("rounds", "decomposed")[k - 1] + " vs fused " + tag
            if B >= 2 and it % 3 == 0:       # two shards under the whole batch's weight range = the whole batch
                form = ("fused", "rounds", "decomposed")[it // 3 % 3]
                mm = ops.maxcut_ls_weights(env.graph, xs0, 2 if first else 1, return_minmax=True)[1]
                h = B // 2
                for lo, hi in ((0, h), (h, B)):
                    part = EnvMaxcut(mygraph=[tuple(int(v) for v in r) for r in garr], device=DEV, if_bidirectional=bidir, num_nodes=n,
                                     env_offset=off + lo)
                    part.stat_hook = lambda kind, t: mm.clone()
                    px, pv = run(part, xs0[lo:hi].contiguous(), form)
                    assert torch.equal(px, res[0][0][lo:hi]) and torch.equal(pv, res[0][1][lo:hi]), f"shard [{lo}, {hi}) {form} " + tag
            env.force_ls_rounds = env.force_ls_fused = False
            env.fused_local_search = True
            env.set_shard(0)
            nrounds += 1
    it += 1
print(f"fuzz_ls: {it} random configurations ({nrounds} also round kernels vs fused), no mismatch")
