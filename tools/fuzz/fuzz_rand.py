"""Differential fuzz of the counter-based generators against their numpy restatements at random (often odd) sizes: rand_spins,
rand_actions, rand_perms, rand_couplings (ER / BA, every edge type, shard offsets), written into tensors with a canary row
behind them.  `python tools/fuzz/fuzz_rand.py [seconds] [seed]`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_np as onp
from rlsolver_amd import ops, _abi
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (forced forms)
from rlsolver_amd import ops_mcpg_tsp as mops
from rlsolver_amd.ops import _ptr, _stream

DEV = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    B = int(rng.choice([1, 3, 63, 64, 65, 200, 1000]))
    N = int(rng.choice([rng.randint(1, 20), rng.randint(20, 200), rng.randint(200, 1300), 512, 1024, 2000]))
    seed = int(rng.randint(1 << 62))
    off = int(rng.choice([0, 5, 1 << 33]))
    tag = f"it={it} B={B} N={N} seed={seed} off={off}"
    if "-v" in sys.argv:
        print(tag, flush=True)
    # rand_spins into the head of a larger buffer: the tail must stay untouched
    buf = torch.full((B * N + 64,), 7, dtype=torch.uint8, device=DEV)
    x = buf[:B * N].view(B, N).view(torch.bool)
    ops.rand_spins(B, N, seed, DEV, env_offset=off, out=x)
    want = onp.rand_spins(B, N, seed, off)
    assert np.array_equal(x.view(torch.uint8).cpu().numpy(), want), "rand_spins " + tag
    assert bool((buf[B * N:] == 7).all()), "rand_spins wrote past its rows " + tag
    a = ops.rand_actions(B, N, seed, 3, DEV)
    assert np.array_equal(a.cpu().numpy(), onp.rand_actions(B, N, seed, 3)), "rand_actions " + tag
    if N >= 2:
        p = mops.rand_perms(B, N, seed, DEV)
        assert np.array_equal(p.cpu().numpy(), onp.rand_perms(B, N, seed)), "rand_perms " + tag
    if N >= 3 and N <= 300 and B <= 200:
        et = int(rng.randint(1, 4))
        Bc = min(B, 40)
        for st, dt in ((4, torch.float32), (8, torch.float64)):
            big = torch.full((Bc * N * N + 32,), 9, dtype=dt, device=DEV)
            m = big[:Bc * N * N].view(Bc, N, N)
            _abi.call("rls_rand_couplings", _ptr(m), st, Bc, N, 0, 0.25, 0, et, seed, off, _stream(DEV))
            assert np.array_equal(m.cpu().numpy(), onp.rand_couplings_er(Bc, N, 0.25, et, seed, off).astype(m.cpu().numpy().dtype)), "ER " + tag
            assert bool((big[Bc * N * N:] == 9).all()), "ER wrote past its matrices " + tag
        mi = int(rng.randint(1, min(6, N - 1) + 1))
        if N <= 120:
            big = torch.full((Bc * N * N + 32,), 9, dtype=torch.float32, device=DEV)
            m = big[:Bc * N * N].view(Bc, N, N)
            _abi.call("rls_rand_couplings", _ptr(m), 4, Bc, N, 1, 0.0, mi, et, seed, off, _stream(DEV))
            assert np.array_equal(m.cpu().numpy(), onp.rand_couplings_ba(Bc, N, mi, et, seed, off).astype(np.float32)), f"BA m={mi} " + tag
            assert bool((big[Bc * N * N:] == 9).all()), "BA wrote past its matrices " + tag
    it += 1
print(f"fuzz_rand: {it} random configurations, no mismatch")
