"""The drop-in env classes against golden vectors captured from the reference classes."""
import types

import numpy as np
import pytest
import torch

from oracle import oracle_np as onp
from tests.gpu_util import DEV, to_dev_bool

pytestmark = pytest.mark.gpu


def mygraph_of(arr):
    return [tuple(int(v) for v in r) for r in arr]


@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0", "ER_100_ID0", "gset_14_stub"])
@pytest.mark.parametrize("bidir", [False, True])
def test_env_l2a_local_search_inplace_golden(golden, gname, bidir):
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    z = golden("maxcut_local_search")
    env = EnvMaxcut(mygraph=mygraph_of(z[f"{gname}/graph"]), device=DEV, if_bidirectional=bidir)
    tag = f"{gname}/bidir{int(bidir)}"
    assert env.num_nodes == onp.num_nodes_distinct(z[f"{gname}/graph"]) and env.if_maximize
    xs = to_dev_bool(z[f"{tag}/ls/xs_in"]).clone()
    noise = torch.from_numpy(z[f"{tag}/ls/noise"]).to(DEV)
    gx, gv = env.local_search_inplace(xs, torch.empty(()), num_iters=8, num_spin=int(z[f"{tag}/ls/num_spin"]),
                                      noise_std=0.3, noise=noise)
    assert gx.data_ptr() == xs.data_ptr()                      # in place, like the reference
    assert gv.dtype == torch.int64 and gx.dtype == torch.bool
    assert np.array_equal(gx.cpu().numpy().astype(np.uint8), z[f"{tag}/ls/xs_out"])
    assert np.array_equal(gv.cpu().numpy(), z[f"{tag}/ls/vs_out"])


@pytest.mark.parametrize("gname", ["BA_5_ID0", "PL_20_ID0", "ER_100_ID0"])
@pytest.mark.parametrize("bidir", [False, True])
def test_env_l2a_objectives_golden(golden, gname, bidir):
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    z = golden("maxcut_obj")
    env = EnvMaxcut(mygraph=mygraph_of(z[f"{gname}/graph"]), device=DEV, if_bidirectional=bidir)
    tag = f"{gname}/bidir{int(bidir)}"
    assert np.array_equal(env.n0_num_n1.cpu().numpy(), z[f"{tag}/n0_num_n1"])
    assert env.n0_ids.shape == (1, len(z[f"{gname}/graph"]) * (2 if bidir else 1))
    for seed in (0, 1, 2):
        t = f"{tag}/seed{seed}"
        xs = to_dev_bool(z[f"{t}/xs"])
        v = env.calculate_obj_values(xs)
        assert str(v.dtype) == str(z[f"{t}/obj_dtype"]) and np.array_equal(v.cpu().numpy(), z[f"{t}/obj"])
        raw = env.calculate_obj_values_for_loop(xs, if_sum=False)
        assert str(raw.dtype) == str(z[f"{t}/cutdeg_dtype"]) and np.array_equal(raw.cpu().numpy(), z[f"{t}/cutdeg"])
        lp = env.calculate_obj_values_for_loop(xs, if_sum=True)
        assert str(lp.dtype) == str(z[f"{t}/obj_loop_dtype"]) and np.array_equal(lp.cpu().numpy(), z[f"{t}/obj_loop"])
        if f"{t}/edge_mask" in z.files and not bidir:
            assert np.array_equal(env.calculate_obj_values(xs, if_sum=False).cpu().numpy().astype(np.uint8), z[f"{t}/edge_mask"])
    torch.manual_seed(3)
    a = env.generate_xs_randomly(50)
    torch.manual_seed(3)
    b = env.generate_xs_randomly(50)
    assert a.dtype == torch.bool and a.shape == (50, env.num_nodes) and torch.equal(a, b) and not a[:, 0].any()
    adj = env.adjacency_bool.cpu().numpy()
    assert np.array_equal(adj, adj.T) and len(env.adjacency_indies) == env.num_nodes


@pytest.mark.parametrize("gname", ["BA_100_ID0", "PL_20_ID0"])
def test_local_search_class_golden(golden, gname):
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    from rlsolver_amd.methods.LocalSearch import LocalSearch
    z = golden("local_search_class")
    env = EnvMaxcut(mygraph=mygraph_of(z[f"{gname}/graph"]), device=DEV, if_bidirectional=False)
    tag = f"{gname}/bidir0"
    ls = LocalSearch(simulator=env, num_nodes=env.num_nodes)
    vs = ls.reset(to_dev_bool(z[f"{tag}/xs_in"]).clone())
    assert np.array_equal(vs.cpu().numpy(), z[f"{tag}/vs_reset"])
    for r in range(2):
        gx, gv, nupd = ls.random_search(num_iters=4, num_spin=4, noise_std=0.3,
                                        noise=torch.from_numpy(z[f"{tag}/round{r}/noise"]).to(DEV))
        assert np.array_equal(gx.cpu().numpy().astype(np.uint8), z[f"{tag}/round{r}/xs"])
        assert np.array_equal(gv.cpu().numpy(), z[f"{tag}/round{r}/vs"])
        assert nupd == int(z[f"{tag}/round{r}/num_update"])
    env_b = EnvMaxcut(mygraph=mygraph_of(z[f"{gname}/graph"]), device=DEV, if_bidirectional=True)
    lsb = LocalSearch(simulator=env_b, num_nodes=env_b.num_nodes)
    lsb.reset(to_dev_bool(z[f"{tag}/xs_in"]).clone())
    with pytest.raises(RuntimeError):          # the reference raises here as well
        lsb.random_search(num_iters=1, num_spin=4)
    xs = ls.reset_search(4)
    assert xs.shape == (4, env.num_nodes) and xs.dtype == torch.bool
    # chunked repeats (memory bound): a budget of one repeat at a time still returns num_sims incumbents, each at least
    # as good as a lone random row would be on average (best of 16 draws)
    ls.RESET_SEARCH_MAX_BYTES = 1
    torch.manual_seed(0)
    xs16 = ls.reset_search(16)
    assert xs16.shape == (16, env.num_nodes)
    assert float(env.calculate_obj_values(xs16).float().mean()) > float(env.calculate_obj_values(env.generate_xs_randomly(256)).float().mean())


@pytest.mark.parametrize("gname", ["BA_100_ID0", "gset_14_stub"])
@pytest.mark.parametrize("bidir", [False, True])
def test_env_ppo_class_golden(golden, gname, bidir):
    from rlsolver_amd.envs.env_PPO import EnvMaxcut
    z = golden("env_ppo")
    graph = z[f"{gname}/graph"]
    tag = f"{gname}/bidir{int(bidir)}"
    n = int(graph[:, :2].max()) + 1
    args = types.SimpleNamespace(num_nodes=n, num_envs=16, num_steps=20)
    env = EnvMaxcut(args, mygraph=mygraph_of(graph), device=DEV, if_bidirectional=bidir)
    obs = env.reset()
    assert obs.dtype == torch.float32 and obs.shape == (16, n) and not obs[:, 0].any()
    # load the golden initial state into the env (reset() draws from our own generator)
    env.xs.copy_(torch.from_numpy(z[f"{tag}/xs0"]).to(DEV).float())
    env._obj = env.calculate_obj_values().to(torch.int32)
    env.last_reward = env._obj.float()
    assert np.array_equal(env.last_reward.cpu().numpy(), z[f"{tag}/cut0"])
    for t in range(50):
        xs, r, d, c = env.step(torch.from_numpy(z[f"{tag}/actions"][t]).to(DEV))
        assert xs.data_ptr() == env.xs.data_ptr()
        assert [str(v.dtype) for v in (xs, r, d, c)] == list(z[f"{tag}/ret_dtypes"])
        assert np.array_equal(r.cpu().numpy(), z[f"{tag}/rewards"][t])
        assert np.array_equal(d.cpu().numpy(), z[f"{tag}/dones"][t])
        assert np.array_equal(c.cpu().numpy(), z[f"{tag}/curs"][t])
    assert np.array_equal((env.xs > 0).cpu().numpy().astype(np.uint8), z[f"{tag}/xs_final"])
    # emitting variant: next state lands in a caller buffer (rollout slot)
    slot = torch.empty_like(env.xs)
    prev = env.xs.clone()
    xs, r, d, c = env.step(torch.zeros(16, dtype=torch.int64, device=DEV), out=slot)
    assert xs.data_ptr() == slot.data_ptr() and torch.equal(slot[:, 1:], prev[:, 1:]) and torch.equal(slot[:, 0], 1 - prev[:, 0])


@pytest.mark.parametrize("bidir", [False, True])
def test_env_ppo_state_edited_behind_the_env(golden, bidir):
    """The reference recomputes the cut from self.xs every step (env_PPO.py:96-98): a caller that edits env.xs between steps
    gets reward = new cut - last_reward there.  Here the objective is incremental: an in-place edit needs resync(), an
    assignment to env.xs is seen by itself -- rewards / curs then equal the reference-shaped oracle's, edit included."""
    from oracle.oracle_torch import PPOEnvRefShaped
    from rlsolver_amd.envs.env_PPO import EnvMaxcut
    z = golden("env_ppo")
    graph = z["BA_100_ID0/graph"]
    tag = f"BA_100_ID0/bidir{int(bidir)}"
    n = int(graph[:, :2].max()) + 1
    args = types.SimpleNamespace(num_nodes=n, num_envs=16, num_steps=20)
    env = EnvMaxcut(args, mygraph=mygraph_of(graph), device=DEV, if_bidirectional=bidir)
    ref = PPOEnvRefShaped(graph, n, 16, 20, bidir)
    env.reset()
    env.xs = torch.from_numpy(z[f"{tag}/xs0"]).to(DEV).float()               # assignment: seen, no call needed
    ref.reset_to(z[f"{tag}/xs0"] > 0)
    env.last_reward = ref.last_reward.to(DEV)
    rng = np.random.RandomState(3)
    for t in range(12):
        if t in (3, 7):                                                      # an in-place edit of a block of spins, both sides
            rows, cols = rng.randint(0, 16, 5), rng.randint(1, n, 5)
            env.xs[torch.from_numpy(rows).to(DEV), torch.from_numpy(cols).to(DEV)] = 1.0
            ref.xs[torch.from_numpy(rows), torch.from_numpy(cols)] = 1.0
            env.resync()
        if t == 9:                                                           # a whole new state by assignment
            new = torch.from_numpy(rng.randint(0, 2, (16, n))).float()
            env.xs = new.to(DEV)
            ref.xs = new.clone()
        a = torch.from_numpy(z[f"{tag}/actions"][t])
        xs, r, d, c = env.step(a.to(DEV))
        rx, rr, rd, rc = ref.step(a)
        assert torch.equal(xs.cpu(), rx) and torch.equal(r.cpu(), rr) and torch.equal(c.cpu(), rc) and torch.equal(d.cpu(), rd), t


def test_select_wrappers_golden(golden):
    from rlsolver_amd.methods.util_read_data import evolutionary_replacement, pick_xs_by_vs, update_xs_by_vs
    z = golden("select_ops")
    a = to_dev_bool(z["update/xs0"]).clone()
    b = torch.from_numpy(z["update/vs0"]).to(DEV).clone()
    ret = update_xs_by_vs(a, b, to_dev_bool(z["update/xs1"]), torch.from_numpy(z["update/vs1"]).to(DEV), True)
    assert ret == int(z["update/max1/ret"])
    assert np.array_equal(a.cpu().numpy().astype(np.uint8), z["update/max1/xs"])
    gx, gv = pick_xs_by_vs(to_dev_bool(z["update/xs0"]), torch.from_numpy(z["update/vs0"]).to(DEV), int(z["pick/R"]), True)
    assert np.array_equal(gx.cpu().numpy().astype(np.uint8), z["pick/max1/xs"])
    xs = to_dev_bool(z["update/xs0"]).clone()
    vs = torch.from_numpy(z["evo/vs_in"]).to(DEV).clone()
    torch.manual_seed(0)
    evolutionary_replacement(xs, vs, 5, True)
    best5 = np.sort(z["evo/vs_in"])[-5:]
    assert np.sum(np.isin(vs.cpu().numpy(), best5)) == 10       # the 5 best now appear twice


def test_evolutionary_replacement_golden(golden, monkeypatch):
    """methods/util.py:87-94 with the reference's recorded randperm: rows and values bit for bit."""
    from rlsolver_amd.methods import util_read_data as urd
    z = golden("select_ops")
    perm = torch.from_numpy(z["evo/max1/perm"]).to(DEV)
    monkeypatch.setattr(urd.th, "randperm", lambda n, device=None: perm[:n] if perm.numel() == n else (_ for _ in ()).throw(AssertionError(n)))
    xs = to_dev_bool(z["update/xs0"]).clone()
    vs = torch.from_numpy(z["evo/vs_in"]).to(DEV).clone()
    urd.evolutionary_replacement(xs, vs, 5, True)
    assert np.array_equal(xs.cpu().numpy().astype(np.uint8), z["evo/max1/xs"])
    assert np.array_equal(vs.cpu().numpy(), z["evo/max1/vs"])
    monkeypatch.undo()
    with pytest.raises(IndexError):                             # the reference's minimise branch fails the same way
        urd.evolutionary_replacement(xs, vs, 5, False)
    # rows that are a multiple of 16 bytes (vector path) and float values (torch moves them)
    x2 = (torch.rand((64, 160), device=DEV) < 0.5)
    v2 = torch.randperm(64, device=DEV).float()
    keep = x2.clone()
    urd.evolutionary_replacement(x2, v2, 8, True)
    vals, counts = torch.unique(v2, return_counts=True)
    assert int((counts == 2).sum()) == 8 and bool((vals[counts == 2] >= 56).all())
    for val in vals[counts == 2]:
        rows = torch.nonzero(v2 == val).flatten()
        assert torch.equal(x2[rows[0]], x2[rows[1]])
    assert int((x2 != keep).any(dim=1).sum()) <= 8


def test_evaluator_tracks_best_on_device(tmp_path):
    """Evaluator.record2 (util_evaluator.py:90-107) as a device-side tracker: same decisions as the reference's
    host logic on a random stream of batches (first argmax, strict improvement), int64 and float values, both
    directions, single-solution form; the value log and best_x_str read back what the reference would hold."""
    from rlsolver_amd.methods.util_evaluator import EncoderBase64, Evaluator
    rng = np.random.RandomState(8)
    N = 77
    for maximize in (True, False):
        for dtype in (torch.int64, torch.float32):
            x0 = torch.from_numpy(rng.randint(0, 2, N).astype(bool)).to(DEV)
            ev = Evaluator(str(tmp_path / f"{maximize}{dtype}"), N, x0, 50, maximize)
            best_v, best_x, log = 50.0, x0.cpu().numpy(), [50.0]
            for it in range(1, 40):
                B = int(rng.randint(1, 3000))
                xs = torch.from_numpy(rng.randint(0, 2, (B, N)).astype(bool)).to(DEV)
                vs_np = rng.randint(0, 100, B)
                flag = ev.record2(it, torch.from_numpy(vs_np).to(DEV).to(dtype), xs)
                gi = int(vs_np.argmax() if maximize else vs_np.argmin())
                gv = float(vs_np[gi])
                upd = gv > best_v if maximize else gv < best_v
                if upd:
                    best_v, best_x = gv, xs[gi].cpu().numpy()
                log.append(gv)
                assert bool(flag) is bool(upd), (maximize, dtype, it)
            assert ev.best_v == best_v and np.array_equal(ev.best_x.cpu().numpy(), best_x)
            assert [r[1] for r in ev.recorder2] == log and [r[0] for r in ev.recorder2] == list(range(40))
            assert ev.first_v == 50.0 and ev.best_x_str == EncoderBase64(N).bool_to_str(best_x).replace("\n", "")
            # single solution + python float value (TNCO-style callers pass good_v.item())
            one = torch.from_numpy(rng.randint(0, 2, N).astype(bool)).to(DEV)
            better = best_v + 1 if maximize else best_v - 1
            assert bool(ev.record2(99, better, one)) and ev.best_v == better and torch.equal(ev.best_x, one)
            assert "best" in ev.logging_print(show_str="x", if_show_x=flag)
            ev.save_record_draw_plot()
            assert (tmp_path / f"{maximize}{dtype}" / "recorder2.npy").exists()


@pytest.mark.parametrize("maximize", [True, False])
@pytest.mark.parametrize("vdt", ["int64", "float32"])
def test_evaluator_golden(golden, tmp_path, maximize, vdt):
    """evaluator.npz: the reference's Evaluator (util_evaluator.py:66-107) on a seeded stream of batches and single
    solutions.  The device tracker (rls_best_update) returns the same if_update, holds the same incumbent after every
    call, and its logs read back as the reference's recorder1 / recorder2."""
    from rlsolver_amd.methods.util_evaluator import Evaluator
    from tests.test_oracle_golden import _evaluator_stream
    z = golden("evaluator")
    tag = f"max{int(maximize)}/{vdt}"
    n = int(z[f"{tag}/num_bits"])
    ev = Evaluator(str(tmp_path / "ev"), n, to_dev_bool(z[f"{tag}/x0"]), float(z[f"{tag}/v0"]), maximize)
    for it, xs, vs, single, upd, best_v, best_x in _evaluator_stream(z, tag):
        ev.record1(i=it, v=float(vs.max()))
        dx, dv = torch.from_numpy(xs).to(DEV), torch.from_numpy(vs).to(DEV)
        got = ev.record2(it, dv[0] if single else dv, dx[0] if single else dx)
        assert bool(got) is upd, (tag, it)
        assert ev.best_v == best_v and np.array_equal(ev.best_x.cpu().numpy(), best_x), (tag, it)
    assert np.array_equal(np.asarray([(r[0], r[1]) for r in ev.recorder2], dtype=np.float64), z[f"{tag}/recorder2_i_v"])
    assert np.array_equal(np.asarray(ev.recorder1, dtype=np.float64), z[f"{tag}/recorder1"])
    assert ev.first_v == float(z[f"{tag}/first_v"]) and ev.best_x_str == str(z[f"{tag}/best_x_str"])


def test_winner_message_round_trip_and_key_unpack():
    """C2's device half (rls_winner_message / rls_winner_unpack) and rls_key_unpack: the rank whose low code is in the reduced
    key writes its global index + bit-packed row (the layout of dist.pack_bits), every other rank zeros; a SUM of the messages
    unpacks to the winner's row and index."""
    from rlsolver_amd import dist as rdist
    from rlsolver_amd.torch_ops import ops as R          # (the recorder of tests/test_gpu_zz_op_coverage.py when it is on)
    rng = np.random.RandomState(9)
    for N, B in ((203, 40), (8, 3), (64, 1), (1001, 7)):
        xs = torch.from_numpy(rng.randint(0, 2, size=(B, N)).astype(np.uint8)).to(DEV).bool()
        li = int(rng.randint(B))
        index = torch.tensor([li], dtype=torch.int64, device=DEV)
        key = torch.tensor([(12345 << 20) | 5], dtype=torch.int64, device=DEV)
        nb = 8 + (N + 7) // 8
        mine, other = torch.empty(nb, dtype=torch.uint8, device=DEV), torch.full((nb,), 7, dtype=torch.uint8, device=DEV)
        R.winner_message(xs, index, key, 20, 5, 1 << 33, N, mine)            # my code is the key's: I am the winner
        R.winner_message(xs, index, key, 20, 4, 1 << 33, N, other)           # another rank: zeros
        R.winner_message(None, None, key, 20, 5, 0, N, other[:nb])           # ... and a rank without envs: zeros too
        assert int(other.sum()) == 0
        assert torch.equal(mine[8:], rdist.pack_bits(xs[li]))
        assert int.from_bytes(bytes(mine[:8].cpu().tolist()), "little") == (1 << 33) + li
        x_out, g_out = torch.empty(N, dtype=torch.bool, device=DEV), torch.empty(1, dtype=torch.int64, device=DEV)
        R.winner_unpack(mine + other, N, x_out, g_out)
        assert torch.equal(x_out, xs[li]) and int(g_out) == (1 << 33) + li
        R.winner_message(xs[li].contiguous(), index, key, 20, 5, 0, N, mine)  # the single-row form: index only supplies the env id
        R.winner_unpack(mine, N, x_out, g_out)
        assert torch.equal(x_out, xs[li]) and int(g_out) == li
    obj, owner = torch.empty(1, dtype=torch.int64, device=DEV), torch.empty(1, dtype=torch.int64, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    R.key_unpack(torch.tensor([(-77 << 20) | 3], dtype=torch.int64, device=DEV), 20, 8, obj, owner, -(1 << 63), flag)
    assert int(obj) == -77 and int(owner) == 8 - 1 - 3 and int(flag) == 0
    objf = torch.empty(1, dtype=torch.float64, device=DEV)
    R.key_unpack(torch.tensor([(15 << 20) | 0], dtype=torch.int64, device=DEV), 20, 1, objf, None, -(1 << 63), flag)
    assert float(objf) == 7.5
    R.key_unpack(torch.tensor([-(1 << 63)], dtype=torch.int64, device=DEV), 20, 4, obj, owner, -(1 << 63), flag)
    assert int(flag) == 2                                                      # no rank had an env


def test_best_key_and_single_process_global_best():
    """rls_best_key: first argmax + the packed MAXLOC key of the episode-boundary exchange in one launch (all value types,
    ties, negative values, the float surface carried doubled, the range / half-integer flag), and dist.global_best on device
    tensors without a process group."""
    from rlsolver_amd import dist as rdist
    R = torch.ops.rlsolver_hip
    rng = np.random.RandomState(4)
    key = torch.zeros(1, dtype=torch.int64, device=DEV)
    idx = torch.zeros(1, dtype=torch.int64, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    for B in (1, 63, 1024, 70001):
        v = rng.randint(-500, 500, size=B).astype(np.int64)
        v[rng.randint(B)] = v.max()                                       # (often a tie: the first position wins)
        for dt, scale in ((torch.int64, 1), (torch.int32, 1), (torch.float32, 2), (torch.float64, 2)):
            t = torch.from_numpy(v).to(DEV).to(dt)
            if scale == 2:
                t = t / 2                                                 # half-integers, as count / 2 of a bidirectional env
            R.best_key(t, 20, 5, 1 << 42, key, idx, flag)
            assert int(key) == (int(v.max()) << 20) | 5 and int(idx) == int(v.argmax()) and int(flag) == 0, (B, dt)
            best, owner, bx = rdist.global_best(t, torch.zeros((B, 3), dtype=torch.bool, device=DEV), want_solution=True)
            assert float(best) == v.max() / scale and int(owner) == 0 and bx.shape == (3,)
    R.best_key(torch.tensor([0.25, 0.125], device=DEV), 20, 0, 1 << 42, key, None, flag)    # the maximum is not a half-integer
    assert int(flag) == 1
    flag.zero_()
    R.best_key(torch.tensor([1 << 50], device=DEV), 20, 0, 1 << 42, key, None, flag)         # outside the key's range
    assert int(flag) == 1
