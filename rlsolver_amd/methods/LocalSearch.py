"""LocalSearch -- drop-in for rlsolver/methods/LocalSearch.py:27-86 on a HIP EnvMaxcut.

The class only holds the incumbent batch (``good_xs`` / ``good_vs``); every search step is the env's kernels:
``reset_search`` is a K1 launch + a best-of-repeats pick per memory-bounded chunk of repeats (the reference loops
num_sims times), ``random_search`` is the env's local-search pipeline (fused kernel, or K2 + K6 + K5) with this
class's weight factor and threshold rule, followed by the row-wise keep-better kernel.
"""
from __future__ import annotations

from typing import Optional

import torch as th

from .. import ops
from .util_read_data import update_xs_by_vs

TEN = th.Tensor


class LocalSearch:
    def __init__(self, simulator, num_nodes: int):
        self.simulator, self.num_nodes = simulator, num_nodes
        self.num_sims = 0
        self.good_xs = th.tensor([])
        self.good_vs = th.tensor([])

    def reset(self, xs: TEN):
        """LocalSearch.py:36-42: adopt xs as the incumbents; returns their objective values."""
        self.good_xs, self.good_vs = xs, self.simulator.calculate_obj_values(xs=xs)
        self.num_sims = xs.shape[0]
        return self.good_vs

    RESET_SEARCH_MAX_BYTES = 1 << 30     # candidate rows held at once by reset_search

    def reset_search(self, num_sims, num_repeats: Optional[int] = None):
        """LocalSearch.py:44-50: num_sims incumbents, each the best of num_sims random rows (first best on ties).
        The reference holds num_sims rows per iteration; here the num_sims repeats are taken in chunks of as many as fit
        RESET_SEARCH_MAX_BYTES (row r * num_sims + s of a chunk = its candidate r of slot s), each chunk reduced by
        one K1 launch + one best-of-repeats pick and merged into the running best by the keep-better kernel -- memory is
        O(chunk * num_sims * N), never num_sims^2 * N.
        Sharded (the simulator's ``env_offset``, rlsolver_amd/seeding.py): ``num_sims`` is this rank's share of the incumbents;
        the number of repeats is the batch's GLOBAL size (``num_repeats``, default num_sims: the single-process meaning), and
        repeat r of incumbent s is keyed by (one seed per call mixed with r, global id of s) -- the same row whatever the rank
        count or the chunking."""
        return self._reset_search(num_sims, num_sims if num_repeats is None else num_repeats)

    def _reset_search(self, num_sims: int, num_repeats: int):
        sim = self.simulator
        per_repeat = max(1, num_sims * (sim.num_nodes + 8))
        chunk = max(1, min(num_repeats, self.RESET_SEARCH_MAX_BYTES // per_repeat))
        best = best_v = None
        base_seed = sim._next_seed()
        if num_sims == 0:            # an empty shard: nothing to draw (the seed above is still consumed, as on every rank)
            return th.empty((0, sim.num_nodes), dtype=th.bool, device=sim.device)
        for r0 in range(0, num_repeats, chunk):
            reps = min(chunk, num_repeats - r0)
            # ONE keyed launch per chunk: row r * num_sims + s = repeat r0 + r of incumbent s, drawn under that repeat's seed
            cand = ops.rand_spins_repeats([sim._seeds.derive(base_seed, r0 + r) for r in range(reps)], num_sims, sim.num_nodes,
                                          sim.device, env_offset=sim.env_offset)
            cx, cv = ops.pick_best_of_repeats(cand, sim.calculate_obj_values(cand), reps, if_maximize=True)
            if best is not None:     # earlier repeats win ties: the running best replaces the chunk's row when it is >=
                ops.select_better_rows(cx, cv, best, best_v, if_maximize=True)
            best, best_v = cx, cv
        return best

    # ---- checkpoint: the incumbents
    def state_dict(self):
        return {"good_xs": self.good_xs.clone(), "good_vs": self.good_vs.clone(), "num_sims": self.num_sims}

    def load_state_dict(self, d):
        self.good_xs, self.good_vs, self.num_sims = d["good_xs"].clone(), d["good_vs"].clone(), int(d["num_sims"])

    def random_search(self, num_iters: int = 8, num_spin: int = 8, noise_std: float = 0.3,
                      noise: Optional[TEN] = None):
        """LocalSearch.py:53-86: a copy of the incumbents goes through ``num_iters`` noisy top-``num_spin`` multi-flip
        proposals (threshold fixed by the first draw, which is also the first proposal, :66-69; weights
        n0_num_n1 - 2 * cut-degree, :64) and one greedy sweep; rows that did not get worse replace the incumbents.
        ``noise`` f32 [num_iters, B, N] replaces randn_like (test hook)."""
        sim = self.simulator
        if sim.if_bidirectional:
            # the reference fails here: float per-node values against int64 incumbents (LocalSearch.py:60-75)
            raise RuntimeError("Index put requires the source and destination dtypes match, "
                               "got Float for the destination and Long for the source.")
        trial_xs = self.good_xs.clone()
        trial_vs = sim.calculate_obj_values(trial_xs)
        sim.local_search_pipeline(trial_xs, trial_vs, weight_mult=2, num_iters=num_iters, num_spin=num_spin,
                                  noise_std=noise_std, noise=noise, first_draw_proposes=True)
        num_update = update_xs_by_vs(self.good_xs, self.good_vs, trial_xs, trial_vs)
        return self.good_xs, self.good_vs, num_update
