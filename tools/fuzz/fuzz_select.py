"""Differential fuzz of the row-selection ops against the numpy restatements: update_xs_by_vs and pick_xs_by_vs
at random shapes incl. ties, both directions of optimisation.
`python tools/fuzz/fuzz_select.py [seconds] [seed]`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_np as onp
from rlsolver_amd.methods import util_read_data as U

DEV = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
while time.time() < t_end:
    n = int(rng.choice([rng.randint(1, 70), rng.randint(70, 700), rng.randint(700, 2500)]))
    B = int(rng.choice([1, 2, 7, 64, 65, 300]))
    mx = bool(rng.rand() < 0.6)
    tag = f"it={it} n={n} B={B} maximize={mx}"
    xs0, xs1 = rng.randint(0, 2, (B, n)).astype(bool), rng.randint(0, 2, (B, n)).astype(bool)
    vs0, vs1 = rng.randint(0, 6, B).astype(np.int64), rng.randint(0, 6, B).astype(np.int64)          # plenty of ties
    d0, dv0 = dev(xs0), dev(vs0)
    U.update_xs_by_vs(d0, dv0, dev(xs1), dev(vs1), if_maximize=mx)
    a, b = xs0.copy(), vs0.copy()
    onp.update_xs_by_vs(a, b, xs1, vs1, mx)
    assert np.array_equal(d0.cpu().numpy(), a) and np.array_equal(dv0.cpu().numpy(), b), "update_xs_by_vs " + tag
    R = int(rng.choice([1, 2, 5]))
    S = B
    xr, vr = rng.randint(0, 2, (R * S, n)).astype(bool), rng.randint(0, 5, R * S).astype(np.int64)
    gx, gv = U.pick_xs_by_vs(dev(xr), dev(vr), num_repeats=R, if_maximize=mx)
    wx, wv = onp.pick_xs_by_vs(xr, vr, R, mx)
    assert np.array_equal(gx.cpu().numpy(), wx) and np.array_equal(gv.cpu().numpy(), wv), "pick_xs_by_vs " + tag
    it += 1
print(f"fuzz_select: {it} random configurations, no mismatch")
