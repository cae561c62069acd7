"""Differential fuzz of the MCPG kernels against the numpy restatement of the reference's sampler_func / metro_sampling with
recorded draws: random graphs (G(n, m), Barabasi-Albert, stars and multi-hub graphs up to degree ~900, paths), chain counts on
both sides of the tile size, 1-3 passes.  `python tools/fuzz/fuzz_mcpg.py [seconds] [seed]`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_np as onp
from rlsolver_amd import graph as G
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (forced forms)
from rlsolver_amd.methods import MCPG as amcpg
from rlsolver_amd import ops_mcpg_tsp as mops
from rlsolver_amd.ops_mcpg_tsp import PackedChains

DEV = torch.device("cuda:0")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    kind = rng.choice(["gnm", "ba", "hubs", "path"])
    n = int(rng.choice([rng.randint(12, 80), rng.randint(80, 400), rng.randint(400, 1200)]))
    if kind == "gnm":
        m = int(rng.randint(n // 2, min(n * (n - 1) // 2, n * rng.randint(2, 9))))
        graph = np.asarray(G.generate_gnm(n, m, int(rng.randint(1 << 30))), dtype=np.int64)
    elif kind == "ba":
        graph = np.asarray(G.generate_ba(n, int(rng.randint(1, min(8, n - 1))), int(rng.randint(1 << 30))), dtype=np.int64)
    elif kind == "path":
        graph = np.asarray([(i, i + 1, 1) for i in range(n - 1)] + [(i, i + 3, 1) for i in range(0, n - 3, 2)], dtype=np.int64)
    else:
        e = set()
        for h in rng.choice(n, int(rng.randint(1, 4)), replace=False).tolist():
            for j in rng.choice(n, min(n - 1, int(rng.choice([70, 130, 300, 900]))), replace=False).tolist():
                if j != h:
                    e.add((min(h, j), max(h, j)))
        for _ in range(int(rng.randint(0, 2 * n))):
            a, b = rng.randint(0, n, 2)
            if a != b:
                e.add((min(a, b), max(a, b)))
        graph = np.asarray([(a, b, 1) for a, b in sorted(e)], dtype=np.int64)
    ei = graph[:, :2].T.copy()
    deg = np.bincount(ei.reshape(-1), minlength=n)
    order = np.argsort(-deg, kind="stable")
    M = int(rng.choice([1, 3, 8, 32, 65]))
    R = int(rng.choice([1, 2, 4]))
    C = M * R
    num_ls = int(rng.randint(1, 4))
    tag = f"it={it} kind={kind} n={n} E={len(graph)} maxdeg={deg.max()} C={C} (M={M}, R={R}) num_ls={num_ls}"
    if "-v" in sys.argv:
        print(tag, flush=True)
    data = amcpg.make_data(n, ei[0], ei[1], DEV, sorted_degree_nodes=order)
    probs = (rng.rand(n) * 0.6 + 0.2).astype(np.float32)
    start = rng.randint(0, 2, size=(n, C)).astype(np.float32)
    T = max(1, n // 10)
    index = rng.randint(0, n, size=(5 * T, C)).astype(np.int64)
    u = rng.rand(5 * T, C).astype(np.float32)
    want, _ = onp.metro_sampling(probs, start, T, index, u)
    got = amcpg.metro_sampling(dev(probs), dev(start), T, device=DEV, index=dev(index), u=dev(u))
    if "-v" in sys.argv:
        torch.cuda.synchronize(); print("  metro ok", flush=True)
    assert np.array_equal(got.cpu().numpy(), want), "metro_sampling " + tag
    uni = rng.rand(num_ls, n, C).astype(np.float32)
    if rng.rand() < 0.3:                                                        # crowd the draws around one half
        uni = (np.float32(0.5) + rng.randint(-40, 41, size=uni.shape).astype(np.float32) * np.float32(2.0 ** -25)).astype(np.float32)
    vs_w, xs_w, val_w, x_all, exp_w = onp.sampler_func(ei, n, order, want, num_ls, M, R, uni)
    vs_g, xs_g, val_g = amcpg.sampler_func(data, got, num_ls, M, R, DEV, uniforms=dev(uni))
    assert np.array_equal(vs_g.cpu().numpy(), vs_w) and np.array_equal(xs_g.cpu().numpy(), xs_w), "sampler_func " + tag
    assert np.allclose(val_g.cpu().numpy(), val_w, atol=1e-3), "sampler_func value " + tag
    # best-merge of the outer loop (MCPG.py:376-391) on bit-packed kept chains: incumbents partly better, partly worse, ties
    now_res = (vs_w + rng.randint(-2, 3, size=M)).astype(np.float32)
    now_info = (rng.rand(n, M) < 0.5).astype(np.float32)
    w_res, w_info, w_temp, w_max, w_idx = onp.mcpg_merge_best(vs_w, xs_w, now_res, now_info)
    d_res, d_info, xg_p = dev(now_res), PackedChains.pack(dev(now_info)), PackedChains.pack(xs_g.contiguous())
    bv, bi = mops.mcpg_merge_best(vs_g.contiguous(), xg_p, d_res, d_info)
    assert np.array_equal(d_res.cpu().numpy(), w_res) and np.array_equal(d_info.unpack().cpu().numpy(), w_info), "merge_best incumbents " + tag
    ok_t, ok_v, ok_i = np.array_equal(xg_p.unpack().cpu().numpy(), w_temp), float(bv) == float(w_max), int(bi) == w_idx
    assert ok_t and ok_v and ok_i, f"merge_best temp_info={ok_t} max={ok_v} ({float(bv)} vs {float(w_max)}) index={ok_i} ({int(bi)} vs {w_idx}) now_res={now_res} vs={vs_w} " + tag
    it += 1
print(f"fuzz_mcpg: {it} random configurations, no mismatch")
