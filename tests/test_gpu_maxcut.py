"""Parity of the HIP MaxCut kernels (through the C ABI) against the oracle and the golden
vectors captured from the reference.  Integer results must be bit-exact."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as onp
from rlsolver_amd import ops
from tests.gpu_util import DEV, device_graph, gnm_arr, to_dev_bool

pytestmark = pytest.mark.gpu

GRAPH_NAMES = ["BA_5_ID0", "BA_5_ID1", "PL_20_ID0", "BA_100_ID0", "ER_100_ID0", "PL_100_ID0", "gset_14_stub"]


@pytest.mark.parametrize("gname", GRAPH_NAMES)
@pytest.mark.parametrize("bidir", [0, 1])
def test_obj_cutdeg_golden(golden, gname, bidir):
    z = golden("maxcut_obj")
    graph = z[f"{gname}/graph"]
    n = int(z[f"{gname}/bidir{bidir}/num_nodes"])
    g = device_graph(graph, n, bidir)
    for seed in (0, 1, 2):
        t = f"{gname}/bidir{bidir}/seed{seed}"
        xs = to_dev_bool(z[f"{t}/xs"])
        obj = ops.maxcut_obj(g, xs)
        assert obj.dtype == torch.int64
        assert np.array_equal(obj.cpu().numpy(), z[f"{t}/obj"])
        # float32 spins (env_PPO surface) give the same objective
        assert np.array_equal(ops.maxcut_obj(g, xs.float()).cpu().numpy(), z[f"{t}/obj"])
        cd = ops.maxcut_node_cutdeg(g, xs).cpu().numpy()
        want = z[f"{t}/cutdeg"]
        assert np.array_equal(cd, (want * 2 if bidir else want).astype(np.int64))
        if f"{t}/edge_mask" in z.files and not bidir:
            assert np.array_equal(ops.maxcut_edge_cut_mask(g, xs).cpu().numpy().astype(np.uint8), z[f"{t}/edge_mask"])


@pytest.mark.parametrize("n,m,B", [(2000, 19990, 200), (800, 4694, 256), (64, 300, 1), (1000, 5000, 65), (3004, 9000, 129),
                                   (333, 2000, 130), (10000, 9999, 70)])
@pytest.mark.parametrize("bidir", [0, 1])
def test_obj_random_vs_oracle(n, m, B, bidir):
    graph = gnm_arr(n, m, seed=n + m)
    g = device_graph(graph, n, bidir)
    rng = np.random.RandomState(B)
    xs = rng.randint(0, 2, size=(B, n)).astype(np.uint8)
    got = ops.maxcut_obj(g, to_dev_bool(xs)).cpu().numpy()
    assert np.array_equal(got, onp.maxcut_obj(xs, graph, bool(bidir)))
    # invariant under global spin flip
    assert np.array_equal(ops.maxcut_obj(g, to_dev_bool(1 - xs)).cpu().numpy(), got)


@pytest.mark.parametrize("n,m,B,bidir", [(64, 300, 98304 + 37, 0), (2000, 19990, 98304 + 1, 0), (800, 4694, 131072, 1),
                                         (5008, 12000, 98304 + 64, 0)])
def test_obj_large_batch(n, m, B, bidir):
    """Batches of 6 - 8 tiles per CU, ragged last tile included (the sizes at which round 1's persistent producer /
    consumer variant used to take over; it was removed when the plain tile kernel overtook it)."""
    graph = gnm_arr(n, m, seed=n + m)
    g = device_graph(graph, n, bidir)
    xs = ops.rand_spins(B, n, 12345, DEV)
    got = ops.maxcut_obj(g, xs)
    # every row against an independent torch formulation on the device ...
    eu, ev = g.eu.long(), g.ev.long()
    for lo in range(0, B, 8192):
        blk = xs[lo:lo + 8192]
        want = (blk[:, eu] ^ blk[:, ev]).sum(dim=1)
        if bidir:
            want = want // 2
        assert torch.equal(got[lo:lo + 8192], want)
    # ... and a sample of rows (first / last tile, a stride in between) against the oracle
    rows = np.unique(np.concatenate([np.arange(130), np.arange(B - 130, B), np.arange(0, B, 997)]))
    xs_h = xs[torch.from_numpy(rows).to(DEV)].cpu().numpy().astype(np.uint8)
    assert np.array_equal(got.cpu().numpy()[rows], onp.maxcut_obj(xs_h, graph, bool(bidir)))


def test_obj_edge_cases():
    graph = gnm_arr(50, 100, 1)
    g = device_graph(graph, 50, 0)
    out = ops.maxcut_obj(g, torch.zeros((0, 50), dtype=torch.bool, device=DEV))
    assert out.shape == (0,)
    z = ops.maxcut_obj(g, torch.zeros((3, 50), dtype=torch.bool, device=DEV))
    assert (z == 0).all()
    with pytest.raises(TypeError):
        ops.maxcut_obj(g, torch.zeros((3, 50), dtype=torch.bool))  # CPU tensor: no fallback
    with pytest.raises(ValueError):
        ops.maxcut_obj(g, torch.zeros((3, 49), dtype=torch.bool, device=DEV))
    # non-contiguous view
    with pytest.raises(ValueError):
        ops.maxcut_obj(g, torch.zeros((50, 3), dtype=torch.bool, device=DEV).t())
    # multigraph + self loop: duplicates count twice, loops never
    mg = np.array([[0, 1, 1], [0, 1, 1], [2, 2, 1], [1, 2, 1]], dtype=np.int64)
    g2 = device_graph(mg, 3, 0)
    xs = np.array([[0, 1, 0], [0, 0, 1], [1, 1, 1]], dtype=np.uint8)
    assert ops.maxcut_obj(g2, to_dev_bool(xs)).cpu().tolist() == [3, 1, 0]
    d = ops.maxcut_delta_all(g2, to_dev_bool(xs)).cpu().numpy()
    assert np.array_equal(d, onp.maxcut_delta_all(xs, mg, 3))


@pytest.mark.parametrize("n,m,B", [(100, 384, 33), (2000, 19990, 70), (37, 120, 64)])
def test_delta_all_is_flip_gain(n, m, B):
    graph = gnm_arr(n, m, seed=7)
    g = device_graph(graph, n, 0)
    xs = np.random.RandomState(1).randint(0, 2, size=(B, n)).astype(np.uint8)
    d = ops.maxcut_delta_all(g, to_dev_bool(xs)).cpu().numpy()
    assert d.dtype == np.int32
    if n <= 100:
        assert np.array_equal(d, onp.maxcut_delta_all(xs, graph, n))
    else:  # spot check 20 nodes by literal flip + re-evaluation on the GPU
        base = ops.maxcut_obj(g, to_dev_bool(xs)).cpu().numpy()
        for i in np.random.RandomState(2).choice(n, 20, replace=False):
            x1 = xs.copy()
            x1[:, i] ^= 1
            assert np.array_equal(ops.maxcut_obj(g, to_dev_bool(x1)).cpu().numpy() - base, d[:, i])


def test_weighted_delta_golden(golden):
    z = golden("weighted_gain")
    g = device_graph(z["graph"], 100, 0, use_weights=True)
    d = ops.maxcut_delta_all(g, to_dev_bool(z["xs"])).cpu().numpy()
    assert np.array_equal(d, z["gain"])


@pytest.mark.parametrize("gname", ["BA_100_ID0", "gset_14_stub"])
@pytest.mark.parametrize("bidir", [0, 1])
@pytest.mark.parametrize("mode", ["emit_u8", "inplace_u8", "emit_f32", "inplace_f32"])
def test_step_golden(golden, gname, bidir, mode):
    z = golden("env_ppo")
    graph = z[f"{gname}/graph"]
    tag = f"{gname}/bidir{bidir}"
    n = int(graph[:, :2].max()) + 1
    g = device_graph(graph, n, bidir)
    x = to_dev_bool(z[f"{tag}/xs0"])
    if mode.endswith("f32"):
        x = x.float()
    obj = ops.maxcut_obj(g, x).to(torch.int32)
    assert np.array_equal(obj.cpu().numpy().astype(np.float32), z[f"{tag}/cut0"])
    B = x.shape[0]
    reward = torch.empty(B, dtype=torch.float32, device=DEV)
    cur = torch.empty(B, dtype=torch.float32, device=DEV)
    done = torch.empty(B, dtype=torch.float32, device=DEV)
    other = torch.empty_like(x)
    for t in range(50):
        a = torch.from_numpy(z[f"{tag}/actions"][t]).to(DEV)
        dv = float(z[f"{tag}/dones"][t][0])
        if mode.startswith("emit"):
            ops.maxcut_step(g, x, other, a, obj, reward, cur, done, dv)
            x, other = other, x
        else:
            ops.maxcut_step(g, x, x, a, obj, reward, cur, done, dv)
        assert np.array_equal(reward.cpu().numpy(), z[f"{tag}/rewards"][t])
        assert np.array_equal(cur.cpu().numpy(), z[f"{tag}/curs"][t])
        assert np.array_equal(done.cpu().numpy(), z[f"{tag}/dones"][t])
    assert np.array_equal((x > 0).cpu().numpy().astype(np.uint8), z[f"{tag}/xs_final"])


@pytest.mark.parametrize("n,m,B", [(2000, 19990, 203), (96, 400, 5), (10000, 49975, 66),
                                   # rows that are not a multiple of 16 bytes: runs of 8 / 2 / 4 rows are (vector path with
                                   # an element-wise tail on the short last run); 333 and 7003 are not (element-wise kernel)
                                   (1000, 5000, 203), (7000, 20000, 67), (100, 384, 13), (3004, 9000, 41), (333, 2000, 70),
                                   (7003, 9000, 9)])
def test_step_random_vs_oracle(n, m, B):
    graph = gnm_arr(n, m, seed=3)
    g = device_graph(graph, n, 0)
    rng = np.random.RandomState(4)
    xs = rng.randint(0, 2, size=(B, n)).astype(np.uint8)
    env = onp.PPOEnvOracle(graph, n, 10 ** 9, False)
    env.reset_to(xs)
    x = to_dev_bool(xs)
    y = torch.empty_like(x)
    obj = ops.maxcut_obj(g, x).to(torch.int32)
    reward = torch.empty(B, dtype=torch.float32, device=DEV)
    for t in range(12):
        a = rng.randint(0, n, size=B)
        if t == 3:
            a[:] = a[0]           # every env flips the same node
        if t == 4:
            a[: B // 2] = 0       # first / last node
            a[B // 2:] = n - 1
        _, r, _, c = env.step(a)
        ops.maxcut_step(g, x, y, torch.from_numpy(a).to(DEV), obj, reward)
        x, y = y, x
        assert np.array_equal(reward.cpu().numpy(), r)
        assert np.array_equal(obj.cpu().numpy().astype(np.float32), c)
    assert np.array_equal(x.cpu().numpy().astype(np.float32), env.xs)
    # obj stays consistent with a from-scratch evaluation
    assert np.array_equal(ops.maxcut_obj(g, x).cpu().numpy(), obj.cpu().numpy().astype(np.int64))


@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0", "ER_100_ID0", "gset_14_stub"])
@pytest.mark.parametrize("bidir", [0, 1])
def test_greedy_sweep_golden(golden, gname, bidir):
    z = golden("maxcut_local_search")
    graph = z[f"{gname}/graph"]
    n = onp.num_nodes_distinct(graph)
    g = device_graph(graph, n, bidir)
    tag = f"{gname}/bidir{bidir}"
    xs = to_dev_bool(z[f"{tag}/sweep/xs_in"]).clone()
    vs = ops.maxcut_obj(g, xs)
    ops.maxcut_greedy_sweep(g, xs, vs)
    assert np.array_equal(xs.cpu().numpy().astype(np.uint8), z[f"{tag}/sweep/xs_out"])
    assert np.array_equal(vs.cpu().numpy(), z[f"{tag}/sweep/vs_out"])


@pytest.mark.parametrize("n,m,B", [(800, 4694, 70), (2000, 19990, 64), (128, 1000, 129), (1000, 5000, 70), (3004, 9000, 65),
                                   (100, 384, 200)])
def test_greedy_sweep_properties(n, m, B):
    graph = gnm_arr(n, m, seed=11)
    g = device_graph(graph, n, 0)
    xs0 = np.random.RandomState(5).randint(0, 2, size=(B, n)).astype(np.uint8)
    xs = to_dev_bool(xs0).clone()
    vs = ops.maxcut_obj(g, xs)
    v0 = vs.clone()
    ops.maxcut_greedy_sweep(g, xs, vs)
    assert (vs >= v0).all()
    assert torch.equal(ops.maxcut_obj(g, xs), vs)          # incremental value == recomputed
    if n <= 128:
        x_ref, v_ref = onp.greedy_sweep(xs0.astype(bool), onp.maxcut_obj(xs0, graph, False), graph, False)
        assert np.array_equal(xs.cpu().numpy(), x_ref)
        assert np.array_equal(vs.cpu().numpy(), v_ref)
    # repeated sweeps never lose value and stay consistent with a from-scratch evaluation
    # (zero-gain flips are accepted, so the state itself may keep moving)
    for _ in range(5):
        prev = vs.clone()
        ops.maxcut_greedy_sweep(g, xs, vs)
        assert (vs >= prev).all()
    assert torch.equal(ops.maxcut_obj(g, xs), vs)


def _chain_fan_graph(chain=500, fan=1000, seed=0):
    """A path 0..chain-1 (one node per dependency level) whose last 8 nodes fan out to `fan` later nodes, plus
    random edges among those: hundreds of one-node batches followed by batches of 64 -- waves that sat idle
    through the chain must re-synchronise their view of the LDS ring (rls_ring.h: ring_advance)."""
    rng = np.random.RandomState(seed)
    e = [(i, i + 1) for i in range(chain - 1)]
    e += [(chain - 1 - (k % 8), chain + k) for k in range(fan)]
    extra = rng.randint(chain, chain + fan, size=(3 * fan, 2))
    e += [(min(a, b), max(a, b)) for a, b in extra.tolist() if a != b]
    e = sorted(set(e))
    return np.array([(a, b, 1) for a, b in e], dtype=np.int64), chain + fan


@pytest.mark.parametrize("case", ["chain_fan", "g22", "fan_first", "ba_hubs", "star_mid", "two_hubs_1100"])
def test_greedy_sweep_level_schedule_vs_sequential_oracle(case):
    """The level-scheduled multi-wave sweep against the C oracle's strictly sequential flip / re-evaluate /
    keep-if-not-worse loop (env_L2A.py:109-116) at sizes the numpy oracle cannot reach."""
    from oracle import oracle_c as oc
    if case == "chain_fan":
        graph, n = _chain_fan_graph()
    elif case == "fan_first":       # the same shape relabelled so that the wide levels come first
        graph, n = _chain_fan_graph(chain=300, fan=1200, seed=1)
        relabel = np.concatenate([np.arange(n - 300, n), np.arange(0, n - 300)])   # old id -> new id
        graph = np.stack([relabel[graph[:, 0]], relabel[graph[:, 1]], graph[:, 2]], axis=1)
    elif case == "ba_hubs":         # hubs of degree 256 ... 373: groups of their own, lane = neighbour
        from rlsolver_amd.graph import generate_ba
        n = 5000
        graph = np.asarray(generate_ba(n, 10, 4), dtype=np.int64)
    elif case == "star_mid":        # one hub of degree 699 in the MIDDLE of the node order: decided after half its neighbours
        n = 700
        graph = np.asarray([(min(350, j), max(350, j), 1) for j in range(n) if j != 350] + [(j, j + 1, 1) for j in range(0, 349, 3)], dtype=np.int64)
    elif case == "two_hubs_1100":
        n = 1500
        rng = np.random.RandomState(12)
        e = {(0, j) for j in range(1, 1101)} | {(700, j) for j in range(701, 1400)} | {tuple(sorted(p)) for p in rng.randint(0, n, (3000, 2)).tolist() if p[0] != p[1]}
        graph = np.asarray([(a, b, 1) for a, b in sorted(e)], dtype=np.int64)
    else:
        n = 2000
        graph = gnm_arr(2000, 19990, seed=22)
    B = 70
    g = device_graph(graph, n, 0)
    if case in ("ba_hubs", "star_mid", "two_hubs_1100"):
        assert g.sweep_lv_ptr is not None and np.bincount(graph[:, :2].ravel()).max() >= 256
    xs0 = np.random.RandomState(9).randint(0, 2, size=(B, n)).astype(np.uint8)
    xs = to_dev_bool(xs0).clone()
    vs = ops.maxcut_obj(g, xs)
    want_x, want_v = oc.greedy_sweep(xs0.copy(), vs.cpu().numpy().copy(), graph[:, 0], graph[:, 1], False)
    ops.maxcut_greedy_sweep(g, xs, vs)
    assert np.array_equal(vs.cpu().numpy(), want_v)
    assert np.array_equal(xs.cpu().numpy().astype(np.uint8), want_x)


@pytest.mark.parametrize("n,m,B,bidir", [(100, 384, 40, 0), (2000, 19990, 65, 1), (50, 200, 64, 0), (1000, 5000, 130, 0),
                                         (3004, 9000, 67, 1)])
def test_propose_accept(n, m, B, bidir):
    graph = gnm_arr(n, m, seed=13)
    g = device_graph(graph, n, bidir)
    rng = np.random.RandomState(6)
    xs0 = rng.randint(0, 2, size=(B, n)).astype(bool)
    mask = rng.rand(B, n) < 0.05
    mask[0] = False               # empty proposal: tie -> accepted, unchanged
    vs0 = onp.maxcut_obj(xs0, graph, bool(bidir))
    x1 = xs0 ^ mask
    v1 = onp.maxcut_obj(x1, graph, bool(bidir))
    want_x, want_v = xs0.copy(), vs0.copy()
    onp.update_xs_by_vs(want_x, want_v, x1, v1, True)
    xs = to_dev_bool(xs0).clone()
    vs = torch.from_numpy(vs0).to(DEV)
    ops.maxcut_propose_accept(g, xs, to_dev_bool(mask), vs)
    assert np.array_equal(xs.cpu().numpy(), want_x)
    assert np.array_equal(vs.cpu().numpy(), want_v)
    # the same proposal with the mask as the bit tile it becomes (uint64 [ceil(B / 64), N]: an eighth of the bytes)
    from rlsolver_amd.ops_mcpg_tsp import PackedChains
    words = PackedChains.pack(to_dev_bool(mask).t().contiguous()).words
    assert words.shape == ((B + 63) // 64, n)
    xs2 = to_dev_bool(xs0).clone()
    vs2 = torch.from_numpy(vs0).to(DEV)
    ops.maxcut_propose_accept(g, xs2, words, vs2)
    assert np.array_equal(xs2.cpu().numpy(), want_x) and np.array_equal(vs2.cpu().numpy(), want_v)
    with pytest.raises((ValueError, RuntimeError)):
        ops.maxcut_propose_accept(g, xs2, words[:, :-1].contiguous(), vs2)


def test_select_ops_golden(golden):
    z = golden("select_ops")
    for mx in (1, 0):
        a = to_dev_bool(z["update/xs0"]).clone()
        b = torch.from_numpy(z["update/vs0"]).to(DEV).clone()
        ops.select_better_rows(a, b, to_dev_bool(z["update/xs1"]), torch.from_numpy(z["update/vs1"]).to(DEV), bool(mx))
        assert np.array_equal(a.cpu().numpy().astype(np.uint8), z[f"update/max{mx}/xs"])
        assert np.array_equal(b.cpu().numpy(), z[f"update/max{mx}/vs"])
        gx, gv = ops.pick_best_of_repeats(to_dev_bool(z["update/xs0"]), torch.from_numpy(z["update/vs0"]).to(DEV),
                                          int(z["pick/R"]), bool(mx))
        assert np.array_equal(gx.cpu().numpy().astype(np.uint8), z[f"pick/max{mx}/xs"])
        assert np.array_equal(gv.cpu().numpy(), z[f"pick/max{mx}/vs"])


@pytest.mark.parametrize("B,N", [(7, 300), (64, 2000), (3, 5), (130, 129)])
def test_rand_spins_and_actions(B, N):
    xs = ops.rand_spins(B, N, seed=0x1234567890, device=DEV, env_offset=11)
    assert np.array_equal(xs.cpu().numpy().astype(np.uint8), onp.rand_spins(B, N, 0x1234567890, 11))
    a = ops.rand_actions(B, N, seed=99, step=5, device=DEV, env_offset=2)
    assert np.array_equal(a.cpu().numpy(), onp.rand_actions(B, N, 99, 5, 2))
    assert int(a.min()) >= 0 and int(a.max()) < N


def test_rand_spins_is_one_sequence_whatever_kernel_writes_it():
    """spin(b, n) = bit (n & 127) of Philox(seed; (b, n >> 7)) with node 0 := 0: independent of N, of the env offset
    (rank invariance) and of the kernel that writes it (one Philox call per 128 spins for rows of >= 512 nodes, one per
    16-byte piece below that, byte stores for rows that are not 16-byte multiples)."""
    B = 300
    big = ops.rand_spins(B, 2000, 99, DEV)                      # rows kernel
    assert torch.equal(big[:, :496], ops.rand_spins(B, 496, 99, DEV))      # per-piece kernel
    assert torch.equal(big[:, :1001], ops.rand_spins(B, 1001, 99, DEV))    # byte-store kernel
    assert torch.equal(ops.rand_spins(B - 70, 2000, 99, DEV, env_offset=70), big[70:])
    assert not bool(big[:, 0].any()) and 0.45 < float(big[:, 1:].float().mean()) < 0.55
    wide = ops.rand_spins(5, 10000, 99, DEV)                    # two trips of 64 blocks per row
    assert torch.equal(wide[:, :2000], big[:5])
    assert torch.equal(big[:, :1000], ops.rand_spins(B, 1000, 99, DEV))    # rows kernel, 8-byte stores (N % 16 == 8)
    assert torch.equal(big[:, :1996], ops.rand_spins(B, 1996, 99, DEV))    # rows kernel, 4-byte stores (N % 8 == 4)
    assert torch.equal(wide[:, :9000], ops.rand_spins(5, 9000, 99, DEV)) and torch.equal(wide[:, :8204], ops.rand_spins(5, 8204, 99, DEV))


@pytest.mark.parametrize("case", ["ba_hubs", "star", "two_hubs_1100"])
def test_node_stats_on_hub_graphs(case):
    """K2 / K3 / local-search weights on graphs with hubs (max degree >= 256: the bit-sliced kernel's 16-plane form, 16-bit
    fields for the hub groups; >= 1024 for the widest extraction) against the dense expression s * (A s), full and ragged
    tiles, both adjacency forms."""
    from rlsolver_amd.graph import generate_ba
    rng = np.random.RandomState(12)
    if case == "ba_hubs":
        n = 5000
        graph = np.asarray(generate_ba(n, 10, 4), dtype=np.int64)
    elif case == "star":
        n = 700
        graph = np.asarray([(0, j, 1) for j in range(1, n)] + [(j, j + 1, 1) for j in range(1, n - 1, 3)], dtype=np.int64)
    else:
        n = 1500
        e = {(0, j) for j in range(1, 1101)} | {(700, j) for j in range(701, 1400)} | {tuple(sorted(p)) for p in rng.randint(0, n, (3000, 2)) if p[0] != p[1]}
        graph = np.asarray([(a, b, 1) for a, b in sorted(e)], dtype=np.int64)
    deg = np.bincount(graph[:, :2].ravel(), minlength=n)
    assert deg.max() >= 256
    for bidir in (0, 1):
        g = device_graph(graph, n, bidir)
        A = torch.zeros((n, n), device=DEV)
        A[graph[:, 0], graph[:, 1]] = 1
        A = A + A.t()
        for B in (2048, 2048 + 37):
            xs = ops.rand_spins(B, n, 3 + B, DEV)
            s = 2 * xs.float() - 1
            want_delta = (s * (s @ A)).to(torch.int32)                           # sum_j (x_i == x_j ? 1 : -1)
            assert torch.equal(ops.maxcut_delta_all(g, xs), want_delta)
            st = A
            if not bidir:                                                        # stored adjacency = each edge once, at its first node
                st = torch.zeros((n, n), device=DEV)
                st[graph[:, 0], graph[:, 1]] = 1
            sdeg = st.sum(1)
            cut = ((sdeg[None] - s * (s @ st.t())) / 2)
            assert torch.equal(ops.maxcut_node_cutdeg(g, xs).float(), cut)
            assert torch.equal(ops.maxcut_ls_weights(g, xs, 2)[0].float(), sdeg[None] - 2 * cut)
            rows = [0, 1, B - 1]
            assert np.array_equal(ops.maxcut_node_cutdeg(g, xs)[rows].cpu().numpy(),
                                  onp.maxcut_node_cutdeg(xs[rows].cpu().numpy(), graph, n, bool(bidir)))


@pytest.mark.parametrize("dtype,B", [(torch.bool, 27001), (torch.float32, 7001), (torch.float32, 300)])
def test_step_long_rows_staged_beyond_64k_and_nontemporal_stores(dtype, B):
    """K4 on rows of 10^4 nodes: f32 rows (40 KB) are staged with two waves per workgroup (80 KB of LDS, opted in past
    the 64 KB default); a batch larger than the 256 MB Infinity Cache (27 001 x 10^4 bytes; 7 001 x 40 KB) leaves through
    nontemporal stores.  Checked by what does not need a CPU oracle at this size: the next state is the state with
    exactly the action's spin flipped, reward = the change of the recomputed cut, obj stays the recomputed cut;
    ragged last workgroup; an out-of-range action leaves its env alone (reward NaN)."""
    n = 10000
    graph = gnm_arr(n, 9999, seed=70)
    g = device_graph(graph, n, 0)
    x = ops.rand_spins(B, n, 5, DEV)
    x = x.float() if dtype == torch.float32 else x
    obj0 = ops.maxcut_obj(g, x)
    obj = obj0.to(torch.int32)
    act = ops.rand_actions(B, n, 3, 0, DEV)
    act[B // 2] = n + 5
    y = torch.empty_like(x)
    rew = torch.empty(B, dtype=torch.float32, device=DEV)
    ops.maxcut_step(g, x, y, act, obj, rew)
    ok = torch.ones(B, dtype=torch.bool, device=DEV)
    ok[B // 2] = False
    diff = (x != y)
    assert torch.equal(diff.sum(dim=1), ok.long())
    rows = torch.nonzero(ok).flatten()
    assert bool(diff[rows, act[rows]].all())
    obj1 = ops.maxcut_obj(g, y)
    assert torch.equal(obj.long(), obj1)
    assert torch.equal(rew[rows], (obj1 - obj0)[rows].float()) and bool(torch.isnan(rew[B // 2]))


def test_step_refuses_weighted_graphs(golden):
    """The gym env counts cut EDGES (env_PPO.py:108-121): a weighted graph handle is refused, not silently counted."""
    z = golden("weighted_gain")
    g = device_graph(z["graph"], 100, 0, use_weights=True)
    x = to_dev_bool(z["xs"])
    B = x.shape[0]
    with pytest.raises(RuntimeError, match="counts cut edges"):
        ops.maxcut_step(g, x, torch.empty_like(x), torch.zeros(B, dtype=torch.int64, device=DEV),
                        torch.zeros(B, dtype=torch.int32, device=DEV), torch.empty(B, device=DEV))


@pytest.mark.parametrize("n,B,lead", [(1999, 130, 0), (1999, 65, 3), (337, 9, 1), (81, 64, 2), (2001, 200, 5), (333, 70, 16), (2004, 67, 6)])
def test_rows_that_are_not_dword_aligned_with_garbage_around(n, B, lead):
    """Spin rows that are not 4-byte aligned (odd N, or a view that starts at an odd byte) go through the funnel-shift form of the
    row-piece stage.  The array sits inside a buffer of 0xFF bytes: nothing outside it may leak into a result (the bytes after the
    last row share a dword with it) and nothing outside it may be written."""
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    garr = gnm_arr(n, 5 * n, seed=n + lead)
    g = device_graph(garr, n, 0)
    rng = np.random.RandomState(n + B)
    xs = rng.randint(0, 2, size=(B, n)).astype(np.uint8)
    buf = torch.full((lead + B * n + 37,), 255, dtype=torch.uint8, device=DEV)
    x = buf[lead:lead + B * n].view(B, n)
    x.copy_(torch.from_numpy(xs).to(DEV))
    xb = x.view(torch.bool)
    assert xb.data_ptr() == buf.data_ptr() + lead
    want = onp.maxcut_obj(xs, garr, False)
    assert np.array_equal(ops.maxcut_obj(g, xb).cpu().numpy(), want)                                     # K1
    sub = [0, B // 2, B - 1]
    assert np.array_equal(ops.maxcut_node_cutdeg(g, xb)[sub].cpu().numpy(), onp.maxcut_node_cutdeg(xs[sub], garr, n, False))   # K2
    assert np.array_equal(ops.maxcut_delta_all(g, xb)[sub].cpu().numpy(), onp.maxcut_delta_all(xs[sub], garr, n, None))        # K3
    # K6: a proposal that flips the last nodes of every row (the pieces that straddle the row ends)
    mask = torch.zeros((B, n), dtype=torch.bool, device=DEV)
    mask[:, -3:] = True
    mask[:, 0] = True
    vs = torch.from_numpy(want).to(DEV)
    ops.maxcut_propose_accept(g, xb, mask, vs)
    after = x.cpu().numpy()
    prop = xs ^ mask.cpu().numpy().astype(np.uint8)
    pv = onp.maxcut_obj(prop, garr, False)
    acc = pv >= want
    assert np.array_equal(after, np.where(acc[:, None], prop, xs)) and np.array_equal(vs.cpu().numpy(), np.where(acc, pv, want))
    # K5 and the fused local search write the tile back: values stay 0 | 1, the cut is consistent, the buffer around is untouched
    ops.maxcut_greedy_sweep(g, xb, vs)
    wx, wv = onp.greedy_sweep(after.astype(bool), vs.cpu().numpy() * 0 + onp.maxcut_obj(after, garr, False), garr, False)
    assert np.array_equal(x.cpu().numpy(), wx.astype(np.uint8)) and np.array_equal(vs.cpu().numpy(), wv)
    env = EnvMaxcut(mygraph=[tuple(int(v) for v in r) for r in garr], device=DEV, num_nodes=n)
    v0 = vs.clone()
    env.local_search_inplace(xb, vs, num_iters=3, num_spin=3)
    assert bool((vs >= v0).all()) and int(x.max()) <= 1
    assert np.array_equal(vs.cpu().numpy(), onp.maxcut_obj(x.cpu().numpy(), garr, False))
    assert bool((buf[:lead] == 255).all()) and bool((buf[lead + B * n:] == 255).all())


@pytest.mark.parametrize("dt", ["u8", "f32"])
@pytest.mark.parametrize("n,m,B,off_in,off_out", [(2000, 19990, 131, 16, 16), (2000, 19990, 131, 48, 112), (10000, 9999, 70, 0, 64),
                                                   (12000, 24000, 37, 80, 80), (1000, 5000, 203, 32, 96)])
def test_step_runs_that_start_anywhere_in_a_cache_line(dt, n, m, B, off_in, off_out):
    """K4's staged emit puts its load / store instructions on the cache lines of the global side: a run that starts h x 16 bytes into
    a line lives h slots further into the LDS stage (csrc/rls_step.hip).  Input and output as views at every 16-byte offset inside
    buffers of 0xFF bytes, the same and different offsets on the two sides (f32: the chase form takes the shift only when they
    agree): results against the oracle, nothing written outside the views."""
    graph = gnm_arr(n, m, seed=3)
    g = device_graph(graph, n, 0)
    rng = np.random.RandomState(n + off_in)
    xs = rng.randint(0, 2, size=(B, n)).astype(np.uint8)
    env = onp.PPOEnvOracle(graph, n, 10 ** 9, False)
    env.reset_to(xs)
    esz = 1 if dt == "u8" else 4
    tdt = torch.bool if dt == "u8" else torch.float32
    nbytes = B * n * esz

    def view(off):
        buf = torch.full((nbytes + 512,), 255, dtype=torch.uint8, device=DEV)
        return buf, buf[off:off + nbytes].view(tdt).view(B, n)
    bx, x = view(off_in)
    by, y = view(off_out)
    x.copy_(to_dev_bool(xs) if dt == "u8" else to_dev_bool(xs).float())
    obj = ops.maxcut_obj(g, x).to(torch.int32)
    reward = torch.empty(B, dtype=torch.float32, device=DEV)
    for t in range(6):
        a = rng.randint(0, n, size=B)
        _, r, _, c = env.step(a)
        ops.maxcut_step(g, x, y, torch.from_numpy(a).to(DEV), obj, reward)
        x, y = y, x
        bx, by = by, bx
        off_in, off_out = off_out, off_in
        assert np.array_equal(reward.cpu().numpy(), r) and np.array_equal(obj.cpu().numpy().astype(np.float32), c), t
        assert np.array_equal(x.float().cpu().numpy(), env.xs), t
        for buf, off in ((bx, off_in), (by, off_out)):            # the bytes around the views are still 0xFF
            assert bool((buf[:off] == 255).all()) and bool((buf[off + nbytes:] == 255).all()), t


@pytest.mark.parametrize("dt", ["u8", "f32"])
def test_step_on_the_last_isolated_node_with_col_at_the_end_of_its_buffer(dt):
    """MODE 3 (the f32 emit step) issued its hand-written neighbour-id load for EVERY valid action, also for a node
    without neighbours -- for an isolated node at the end of the CSR that address is col + nnz, one entry past the array.  The
    graph here leaves its last three nodes isolated and its `col` array ends exactly where a 2 MB device buffer ends (the
    DeviceGraph's own tensor is swapped for a view at the tail of a block the allocator hands out whole), so the over-read left the
    allocation; every env acts on an isolated node in two of the steps."""
    n, m, B = 2000, 6000, 131
    rng = np.random.RandomState(7)
    graph = gnm_arr(n - 3, m, seed=11)                       # nodes n-3 .. n-1 have no edge
    g = device_graph(graph, n, 0)
    assert int(g.csr.rowptr[n - 3]) == g.nnz == int(g.csr.rowptr[n])
    block = torch.zeros((2 << 20) // 4, dtype=torch.int32, device=DEV)          # one whole 2 MB allocator block
    tail = block[block.numel() - g.nnz:]
    tail.copy_(g.col)
    g.col = tail
    g.struct.col = tail.data_ptr()
    xs = rng.randint(0, 2, size=(B, n)).astype(np.uint8)
    env = onp.PPOEnvOracle(graph, n, 10 ** 9, False)
    env.reset_to(xs)
    x = to_dev_bool(xs) if dt == "u8" else to_dev_bool(xs).float()
    y = torch.empty_like(x)
    obj = ops.maxcut_obj(g, x).to(torch.int32)
    reward = torch.empty(B, dtype=torch.float32, device=DEV)
    for t in range(6):
        a = rng.randint(0, n, size=B)
        if t in (1, 4):
            a[:] = n - 1 - (np.arange(B) % 3)                # isolated nodes only: gain 0, the spin still flips
        _, r, _, c = env.step(a)
        ops.maxcut_step(g, x, y, torch.from_numpy(a).to(DEV), obj, reward)
        x, y = y, x
        assert np.array_equal(reward.cpu().numpy(), r) and np.array_equal(obj.cpu().numpy().astype(np.float32), c), t
        assert np.array_equal(x.float().cpu().numpy(), env.xs), t


def test_rand_spins_repeats_is_the_per_repeat_launches_in_one():
    """rls_rand_spins_repeats (ABI v11): row r * S + s = rls_rand_spins(seed = seeds[r], env_offset) of env s, for both
    kernels (rows of >= 512 spins take the wave-per-row form) and at a shard offset."""
    for S, N, off in ((37, 2000, 0), (5, 100, 1 << 33), (64, 1000, 12345), (3, 20, 7)):
        seeds = [(0x9E3779B97F4A7C15 * (k + 1)) & ((1 << 62) - 1) for k in range(6)]
        got = ops.rand_spins_repeats(seeds, S, N, DEV, env_offset=off)
        assert got.shape == (6 * S, N)
        for r, sd in enumerate(seeds):
            assert torch.equal(got[r * S:(r + 1) * S], ops.rand_spins(S, N, sd, DEV, env_offset=off)), (S, N, r)
    assert ops.rand_spins_repeats([], 4, 64, DEV).shape == (0, 64)
