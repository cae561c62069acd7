"""Invariant fuzz of the on-device MCPG round with PRODUCTION draws (no oracle: the kernels' own generators): random graphs
(G(n, m), BA, hubs), kept-chain counts M and repeats R around the tile sizes, a few rounds each.  Invariants: incumbents never
get worse, every incumbent value is the cut of its kept chain, the best value / index are the arg-max of the incumbents, the
worst incumbent has been replaced by the best, get_return is finite with a finite gradient.  Every other configuration with
two or more tiles of kept chains is ALSO run as shards of the kept chains (1 : rest and half : half) that replay the whole run's
exchange (the stop rule's accept counts, the mean, the best / worst incumbent, get_return's sums): a shard's chains, expected
cuts and incumbents must be the whole batch's, bit for bit.
`python tools/fuzz/fuzz_mcpg_round.py [seconds] [seed]`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_np as onp
from rlsolver_amd import graph as G
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (forced forms)
from rlsolver_amd.methods import MCPG as amcpg
from rlsolver_amd.ops_mcpg_tsp import PackedChains

DEV = torch.device("cuda:0")


class Tape:
    """The whole-batch statistics of a run: recorded when the batch is whole (local = global), replayed to its shards."""

    def __init__(self):
        self.items = []

    def recorder(self):
        def hook(kind, arg):
            if kind == "best":
                vs, row_of, off, maximize = arg
                li = (vs == vs.max()).nonzero()[0, 0]
                v = vs[li] if maximize else -vs[li]
                out = (v.clone(), (li + off).clone(), None if row_of is None else row_of(li).clone())
            else:
                out = arg.clone()
            self.items.append(out)
            return out if kind == "best" else arg
        return hook

    def replayer(self):
        pos = [0]

        def hook(kind, arg):
            out = self.items[pos[0]]
            pos[0] += 1
            return tuple(None if o is None else o.clone() for o in out) if kind == "best" else out.clone()
        return hook


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    kind = rng.choice(["gnm", "ba", "hub"])
    n = int(rng.choice([rng.randint(12, 100), rng.randint(100, 600), rng.randint(600, 2500)]))
    if kind == "gnm":
        graph = np.asarray(G.generate_gnm(n, int(rng.randint(n, min(n * 6, n * (n - 1) // 2))), int(rng.randint(1 << 30))), dtype=np.int64)
    elif kind == "ba":
        graph = np.asarray(G.generate_ba(n, int(rng.randint(1, min(7, n - 1))), int(rng.randint(1 << 30))), dtype=np.int64)
    else:
        e = {(0, j) for j in rng.choice(np.arange(1, n), min(n - 1, int(rng.choice([70, 300, 900]))), replace=False).tolist()}
        for _ in range(2 * n):
            a, b = rng.randint(0, n, 2)
            if a != b:
                e.add((min(a, b), max(a, b)))
        graph = np.asarray([(a, b, 1) for a, b in sorted(e)], dtype=np.int64)
    M, R = int(rng.choice([64, 128, 320])), int(rng.choice([1, 3, 8]))
    num_ls = int(rng.randint(1, 4))
    tag = f"it={it} kind={kind} n={n} E={len(graph)} M={M} R={R} num_ls={num_ls}"
    if "-v" in sys.argv:
        print(tag, flush=True)
    torch.manual_seed(int(rng.randint(1 << 30)))
    ei = graph[:, :2].T.copy()
    data = amcpg.make_data(n, ei[0], ei[1], DEV)
    kept = (torch.rand((n, M), device=DEV) < 0.5).float()
    cut = lambda cols: torch.from_numpy(onp.maxcut_obj(cols.t().cpu().numpy().astype(np.uint8), graph, False)).float().to(DEV)
    rnd = amcpg.MCPGRound(data, PackedChains.pack(kept), cut(kept), M, R, num_ls)
    probs = torch.rand(n, device=DEV) * 0.6 + 0.2
    prev = rnd.now_max_res.clone()
    shard_check = M >= 128 and it % 2 == 0
    seed0 = int(rng.randint(1 << 30))
    tape, whole_log = Tape(), []
    if shard_check:                       # the whole batch through the sharded code path, recording its exchange
        rnd.sharded, rnd.stat_hook = True, tape.recorder()
    torch.manual_seed(seed0)
    for r in range(3):
        rnd.step(probs)
        if shard_check:
            whole_log.append((rnd.samples.words.clone(), rnd.expected.clone(), rnd.now_max_res.clone(), rnd.now_max_info.words.clone()))
        assert bool((rnd.now_max_res >= prev).all()), "incumbents got worse " + tag
        info = rnd.now_max_info.unpack()
        assert torch.equal(cut(info), rnd.now_max_res), "incumbent value is not the cut of its chain " + tag
        assert float(rnd.best_value) == float(rnd.now_max_res.max()) and float(rnd.now_max_res.min()) >= float(prev.min()), "best / worst " + tag
        assert float(rnd.now_max_res[int(rnd.best_index)]) == float(rnd.best_value), "best index " + tag
        prev = rnd.now_max_res.clone()
    pr = probs.clone().requires_grad_(True)
    obj = rnd.get_return(pr)
    obj.backward()
    assert bool(torch.isfinite(obj)) and bool(torch.isfinite(pr.grad).all()), "get_return " + tag
    if shard_check:
        cuts = [64, M] if it % 4 == 0 else [M // 128 * 64, M]
        m0 = 0
        for m1 in cuts:
            ml = m1 - m0
            part = amcpg.MCPGRound(data, PackedChains.pack(kept[:, m0:m1].contiguous()), cut(kept[:, m0:m1]), ml, R, num_ls,
                                   kept_offset=m0, total_kept=M)
            part.stat_hook = tape.replayer()
            torch.manual_seed(seed0)
            gt = torch.tensor([q * (M // 64) + m0 // 64 + t for q in range(R) for t in range(ml // 64)], device=DEV)
            gc = torch.tensor([q * M + m0 + j for q in range(R) for j in range(ml)], device=DEV)
            for r in range(3):
                part.step(probs)
                ws, we, wr, wi = whole_log[r]
                assert torch.equal(part.samples.words, ws[gt]), f"shard [{m0}, {m1}) round {r}: chains " + tag
                assert torch.equal(part.expected, we[gc]), f"shard [{m0}, {m1}) round {r}: expected " + tag
                assert torch.equal(part.now_max_res, wr[m0:m1]) and torch.equal(part.now_max_info.words, wi[m0 // 64:m1 // 64]), \
                    f"shard [{m0}, {m1}) round {r}: incumbents " + tag
            part.get_return(probs.clone().requires_grad_(True))          # (consumes its exchange like the whole run did)
            m0 = m1
    it += 1
print(f"fuzz_mcpg_round: {it} random configurations, no violation")
