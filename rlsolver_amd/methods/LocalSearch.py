"""LocalSearch -- drop-in for rlsolver/methods/LocalSearch.py:27-86 on a HIP EnvMaxcut."""
from __future__ import annotations

from typing import Optional

import torch as th

from .. import ops
from .util_read_data import update_xs_by_vs

TEN = th.Tensor


class LocalSearch:
    def __init__(self, simulator, num_nodes: int):
        self.simulator = simulator
        self.num_nodes = num_nodes
        self.num_sims = 0
        self.good_xs = th.tensor([])
        self.good_vs = th.tensor([])

    def reset(self, xs: TEN):
        vs = self.simulator.calculate_obj_values(xs=xs)
        self.good_xs = xs
        self.good_vs = vs
        self.num_sims = xs.shape[0]
        return vs

    def reset_search(self, num_sims):
        """LocalSearch.py:44-50: best of num_sims random rows, num_sims times."""
        sim = self.simulator
        xs = th.empty((num_sims, self.num_nodes), dtype=th.bool, device=sim.device)
        for sim_id in range(num_sims):
            _xs = sim.generate_xs_randomly(num_sims=num_sims)
            _vs = sim.calculate_obj_values(_xs)
            xs[sim_id] = _xs[_vs.argmax()]
        return xs

    def random_search(self, num_iters: int = 8, num_spin: int = 8, noise_std: float = 0.3,
                      noise: Optional[TEN] = None):
        """LocalSearch.py:53-86.  ``noise`` f32 [num_iters, B, N] replaces randn_like (test hook)."""
        sim = self.simulator
        kth = self.num_nodes - num_spin

        prev_xs = self.good_xs.clone()
        prev_vs_raw = sim.calculate_obj_values_for_loop(prev_xs, if_sum=False)
        prev_vs = prev_vs_raw.sum(dim=1)
        if prev_vs.dtype != th.int64:
            # the reference fails here too (float prev_vs vs int64 vs, LocalSearch.py:75)
            raise RuntimeError("Index put requires the source and destination dtypes match, "
                               "got Float for the destination and Long for the source.")

        if getattr(sim, "fused_local_search", False) and ops.local_search_fusable(sim.graph, num_spin, prev_xs.shape[0]) and num_iters > 0:
            # pre-pass + one kernel; here the first draw both fixes the threshold and is the first proposal (:66-69)
            ws32, ws_std = ops.maxcut_ls_weights(sim.graph, prev_xs, 2)     # n0_num_n1 - 2 * prev_vs_raw (:64)
            rd_std = ws_std.float() * noise_std
            if noise is not None:
                noise = noise.to(device=sim.device, dtype=th.float32).contiguous()
            seed = 0 if noise is not None else int(th.randint(0, 2 ** 62, (1,), dtype=th.int64).item())
            ops.maxcut_local_search(sim.graph, prev_xs, ws32, rd_std.contiguous(), prev_vs, num_iters, num_spin,
                                    noise=noise, seed=seed, first_draw_proposes=True)
        else:
            ws = sim.n0_num_n1 - (4 if sim.if_bidirectional else 2) * prev_vs_raw
            ws_std = ws.max(dim=0, keepdim=True)[0] - ws.min(dim=0, keepdim=True)[0]
            rd_std = ws_std.float() * noise_std
            thresh = None
            for it in range(num_iters):
                rnd = noise[it] if noise is not None else th.randn_like(ws, dtype=th.float32)
                spin_rand = ws + rnd * rd_std
                thresh = th.kthvalue(spin_rand, k=kth, dim=1)[0][:, None] if thresh is None else thresh
                spin_mask = spin_rand.gt(thresh)
                ops.maxcut_propose_accept(sim.graph, prev_xs, spin_mask, prev_vs)
            ops.maxcut_greedy_sweep(sim.graph, prev_xs, prev_vs)

        num_update = update_xs_by_vs(self.good_xs, self.good_vs, prev_xs, prev_vs)
        return self.good_xs, self.good_vs, num_update
