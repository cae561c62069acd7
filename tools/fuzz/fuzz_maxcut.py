"""Differential fuzz of the MaxCut kernels (K1, K2, K3, K4, K5, K6) against the C oracle on random graph shapes: G(n, m),
stars / hubs of any degree, paths, near-complete graphs, ragged and full tiles, both adjacency forms, batches on both sides
of the kernels' dispatch thresholds.  `python tools/fuzz/fuzz_maxcut.py [seconds] [seed]` -- prints the first mismatch."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_c as oc
from rlsolver_amd import ops
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (forced forms)
from rlsolver_amd.graph import build_csr

DEV = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def random_graph():
    kind = rng.choice(["gnm", "hub", "path", "dense", "tiny", "multi_hub"])
    if kind == "tiny":
        n = int(rng.randint(2, 70))
    elif kind == "dense":
        n = int(rng.randint(20, 300))
    else:
        n = int(rng.choice([rng.randint(64, 400), rng.randint(400, 3000), rng.randint(3000, 7000)]))
    e = set()
    if kind in ("gnm", "tiny"):
        m = int(rng.randint(1, max(2, min(n * (n - 1) // 2, n * rng.randint(1, 12)))))
        while len(e) < m:
            a, b = rng.randint(0, n, 2)
            if a != b:
                e.add((min(a, b), max(a, b)))
    elif kind == "dense":
        p = rng.uniform(0.3, 0.95)
        iu = np.triu_indices(n, 1)
        keep = rng.rand(len(iu[0])) < p
        e = set(zip(iu[0][keep].tolist(), iu[1][keep].tolist()))
    elif kind == "path":
        e = {(i, i + 1) for i in range(n - 1)} | {(i, i + 2) for i in range(0, n - 2, int(rng.randint(1, 5)))}
    else:
        hubs = [int(rng.randint(n))] if kind == "hub" else [int(h) for h in rng.choice(n, int(rng.randint(2, 6)), replace=False)]
        for h in hubs:
            d = int(rng.choice([rng.randint(60, 260), rng.randint(250, 1100), min(n - 1, rng.randint(1000, 4500))]))
            d = min(d, n - 1)
            for j in rng.choice(n, d, replace=False).tolist():
                if j != h:
                    e.add((min(h, j), max(h, j)))
        for _ in range(int(rng.randint(0, 3 * n))):
            a, b = rng.randint(0, n, 2)
            if a != b:
                e.add((min(a, b), max(a, b)))
    if not e:
        e = {(0, 1)}
    el = sorted(e)
    if rng.rand() < 0.5:                                   # stored orientation: either way round
        el = [(b, a) if rng.rand() < 0.5 else (a, b) for a, b in el]
    return kind, n, np.asarray([(a, b, 1) for a, b in el], dtype=np.int64)


t_end = time.time() + budget
it = 0
while time.time() < t_end:
    kind, n, graph = random_graph()
    bidir = int(rng.rand() < 0.4)
    B = int(rng.choice([1, 7, 64, 65, 130, 2048, 2048 + 37, 4096]))
    if n > 3000 and B > 2100:
        B = 2048
    csr = build_csr((graph[:, 0].copy(), graph[:, 1].copy(), graph[:, 2].copy()), num_nodes=n, if_bidirectional=bool(bidir))
    g = ops.DeviceGraph(csr, DEV)
    eu, ev = csr.eu, csr.ev
    xs0 = rng.randint(0, 2, size=(B, n)).astype(np.uint8)
    xs = torch.from_numpy(xs0).to(DEV).view(torch.bool)
    tag = f"it={it} kind={kind} n={n} E={len(graph)} maxdeg={csr.max_degree} B={B} bidir={bidir}"
    want = oc.maxcut_obj(xs0, eu, ev, bidir)
    got = ops.maxcut_obj(g, xs).cpu().numpy()
    assert np.array_equal(got, want), "K1 " + tag
    cd = ops.maxcut_node_cutdeg(g, xs).cpu().numpy()
    erowptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(eu, minlength=n), out=erowptr[1:])
    assert np.array_equal(cd, oc.node_cutdeg(xs0, erowptr, ev)), "K2 " + tag
    # K3 = obj(flip_i) - obj, from the symmetric per-node counts: delta_i = deg_i - 2 * #differing (symmetric adjacency)
    sym = oc.node_cutdeg(xs0, csr.rowptr, csr.col)
    deg = np.diff(csr.rowptr)
    assert np.array_equal(ops.maxcut_delta_all(g, xs).cpu().numpy(), (deg[None, :] - 2 * sym).astype(np.int32)), "K3 " + tag
    # K4: three steps
    x1 = xs.clone().view(torch.uint8)
    y1 = torch.empty_like(x1)
    obj = torch.from_numpy(want.astype(np.int32)).to(DEV)
    rew = torch.empty(B, dtype=torch.float32, device=DEV)
    xo = xs0.copy()
    last = want.astype(np.int64).copy()
    for s in range(3):
        a = rng.randint(0, n, B).astype(np.int64)
        ops.maxcut_step(g, x1.view(torch.bool), y1.view(torch.bool), torch.from_numpy(a).to(DEV), obj, rew)
        x1, y1 = y1, x1
        r = oc.step_u8(xo, a, eu, ev, bidir, last)
        assert np.array_equal(x1.cpu().numpy(), xo) and np.array_equal(obj.cpu().numpy(), last.astype(np.int32)), f"K4 step {s} " + tag
        assert np.array_equal(rew.cpu().numpy(), r.astype(np.float32)), f"K4 reward {s} " + tag
    # K4 on the f32 gym surface, in place (env_PPO keeps float spins and flips them in place)
    xf = xs.float().contiguous()
    objf = torch.from_numpy(want.astype(np.int32)).to(DEV)
    xo = xs0.copy()
    last = want.astype(np.int64).copy()
    for s in range(2):
        a = rng.randint(0, n, B).astype(np.int64)
        ops.maxcut_step(g, xf, xf, torch.from_numpy(a).to(DEV), objf, rew)
        r = oc.step_u8(xo, a, eu, ev, bidir, last)
        assert np.array_equal(xf.cpu().numpy(), xo.astype(np.float32)) and np.array_equal(objf.cpu().numpy(), last.astype(np.int32)), f"K4 f32 in place {s} " + tag
        assert np.array_equal(rew.cpu().numpy(), r.astype(np.float32)), f"K4 f32 reward {s} " + tag
    # K6 + K5
    mask = torch.from_numpy((rng.rand(B, n) < 4.0 / n)).to(DEV)
    xk = xs.clone()
    vk = torch.from_numpy(want).to(DEV)
    ops.maxcut_propose_accept(g, xk, mask, vk)
    prop = xs0 ^ mask.cpu().numpy().astype(np.uint8)
    pv = oc.maxcut_obj(prop, eu, ev, bidir)
    acc = pv >= want
    assert np.array_equal(vk.cpu().numpy(), np.where(acc, pv, want)) and np.array_equal(xk.cpu().numpy().astype(np.uint8), np.where(acc[:, None], prop, xs0)), "K6 " + tag
    if n * 8 + 8192 <= 160 * 1024:          # the same proposal with the mask as a bit tile (uint64 [ceil(B / 64), N])
        from rlsolver_amd.ops_mcpg_tsp import PackedChains
        xk2, vk2 = xs.clone(), torch.from_numpy(want).to(DEV)
        ops.maxcut_propose_accept(g, xk2, PackedChains.pack(mask.t().contiguous()).words, vk2)
        assert torch.equal(xk2, xk) and torch.equal(vk2, vk), "K6 bit-packed mask " + tag
    if B <= 130 or n <= 1500:
        xsw = xs.clone()
        vsw = torch.from_numpy(want).to(DEV)
        wx, wv = oc.greedy_sweep(xs0.copy(), want.copy(), eu, ev, bidir)
        ops.maxcut_greedy_sweep(g, xsw, vsw)
        assert np.array_equal(vsw.cpu().numpy(), wv) and np.array_equal(xsw.cpu().numpy().astype(np.uint8), wx), "K5 " + tag
    it += 1
print(f"fuzz_maxcut: {it} random configurations, no mismatch")
