"""ISCO_TSP -- drop-in for rlsolver/envs/env_ISCO.py:176-363 (TSP sampler) on a HIP device.

``step`` is ONE kernel (rls_isco_tsp_step): the ``path_length`` rounds of opt_2 -> log_softmax -> Gumbel
draw -> reverse-move log-probability -> swap, and the final Metropolis accept, with the tour, its inverse and
the distance matrix resident in LDS.  The reference walks the same rounds as ~40 torch ops each (sort +
searchsorted to invert the tour, a [B, N, N-1] 3-D gather of the random-neighbour table, two softmaxes, two sorts).
``calculate_distance`` (K12), ``opt_2`` (K13) and ``switch`` remain available as the reference's methods.

Unlike the reference, which reads BATCH_SIZE / K / DEVICE from star-imported config modules at call
time, they are explicit constructor arguments here.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .. import ops_mcpg_tsp as mops
from ..ops import _check, _s64, _t
from ..seeding import Sharded, seed_from_torch as _seed_from_torch  # noqa: F401


class ISCO_TSP(Sharded):
    def __init__(self, params_dict, batch_size: int = 1, K: int = 20, device=None, chain_length: int = 10000,
                 init_temperature: float = 1.0, final_temperature: float = 0.1, env_offset: int = 0, seed: Optional[int] = None):
        """``env_offset`` / ``seed``: rlsolver_amd/seeding.py (``batch_size`` tours whose global ids start at env_offset)."""
        self._init_shard(env_offset, seed)
        self.distance = params_dict['distance']
        self.device = torch.device(device) if device is not None else self.distance.device
        if self.device.type != 'cuda':
            raise TypeError(f"rlsolver_amd.ISCO_TSP needs a HIP device (got {self.device}); there is no CPU path")
        self.batch_size = batch_size
        self.K = K
        self.chain_length = chain_length
        self.init_temperature = torch.tensor(init_temperature, device=self.device)
        self.final_temperature = torch.tensor(final_temperature, device=self.device)
        self.num_nodes = params_dict['num_nodes']
        self.distance = self.distance.to(self.device, torch.float32).contiguous()
        self.nearest_indices = params_dict['nearest_indices'].to(self.device)
        self.random_indices = params_dict['random_indices'].to(self.device)
        self._near32 = self.nearest_indices.to(torch.int32).contiguous()
        self._rand32 = self.random_indices.to(torch.int32).contiguous()
        # rand < K / (K + 1): the python double is rounded to f32 by the comparison with an f32 tensor
        self._near_thr = float(np.float32(K / (K + 1)))
        self._tab8 = mops.tsp_tables8(self._near32, self._rand32)          # the byte form opt_2's kernel keeps in LDS (N <= 256)

    def step(self, x, path_length, temperature, draws: Optional[dict] = None, want_terms: bool = False):
        """env_ISCO.py:188-201 -> (y int64 [B, N], mean acceptance probability 0-dim f32).

        ``draws`` (test hook) = the reference's torch draws in call order: u_partner / u_gumbel f32 and r_near /
        r_rand int64, each [path_length, B, N], u_accept f32 [B].  ``want_terms`` additionally returns
        (log_acc [B], walked tour before the accept [B, N])."""
        x = _check(x.contiguous(), "x", (torch.int64,), self.device)
        B, N = x.shape
        L = int(path_length)
        y = torch.empty_like(x)
        acc = torch.empty(B, dtype=torch.float32, device=self.device)
        log_acc = torch.empty(B, dtype=torch.float32, device=self.device) if want_terms else None
        cur = torch.empty_like(x) if want_terms else None
        d = {}
        if draws is not None:
            for k, dt in (("u_partner", torch.float32), ("r_near", torch.int64), ("r_rand", torch.int64),
                          ("u_gumbel", torch.float32)):
                d[k] = _check(draws[k].to(self.device).contiguous(), k, (dt,), self.device, (L, B, N))
            d["u_accept"] = _check(draws["u_accept"].to(self.device).contiguous(), "u_accept", (torch.float32,), self.device, (B,))
        _t.isco_tsp_step(self.distance, self._near32, self._near_thr, self._rand32, x, y, L, float(temperature), d.get("u_partner"),
                         d.get("r_near"), d.get("r_rand"), d.get("u_gumbel"), d.get("u_accept"),
                         _s64(0 if draws is not None else self._next_seed()), self.env_offset, log_acc, acc, cur)
        if want_terms:
            return y, acc.mean(), log_acc, cur
        return y, acc.mean()

    # ---- the pieces of a step as the reference exposes them
    def draw_partners(self, sample):
        """First half of opt_2 (env_ISCO.py:246-266): the partner CITY for every position, drawn with
        torch's generator exactly like the reference (rand, randint K, randint N-K-1) but with 2-D
        gathers instead of materialising nearest_indices[sample] / random_indices[sample]."""
        B, N, K = sample.shape[0], self.num_nodes, self.K
        coin = torch.rand(B, N, device=self.device) < (K / (K + 1))
        pick_near = torch.randint(0, K, (B, N), device=self.device)
        pick_far = torch.randint(0, N - K - 1, (B, N), device=self.device)
        return torch.where(coin, self.nearest_indices[sample, pick_near], self.random_indices[sample, pick_far])

    def opt_2(self, sample, temperature, selected=None, return_selected: bool = False):
        """env_ISCO.py:238-335 -> (-delta/T f32 [B,N], indices int64 [B,N], ban bool [B,N]).  ONE kernel: the partner cities
        (:245-262) are drawn inside it (the counter-based generator of ``step``, keyed by this object's seed stream and the
        GLOBAL tour id) -- ``selected`` (int64 [B,N] partner cities, e.g. from ``draw_partners``: torch's generator, the
        reference's three draws) replaces the draw; ``return_selected`` appends what the kernel drew."""
        sample = sample.contiguous()
        if selected is not None:
            return mops.tsp_swap_delta_all(self.distance, sample, selected.contiguous(), float(temperature))
        return mops.tsp_swap_delta_all(self.distance, sample, None, float(temperature), nearest=self._near32, random=self._rand32,
                                       near_threshold=self._near_thr, seed=self._next_seed(), env_offset=self.env_offset,
                                       return_selected=return_selected, tables8=self._tab8)

    def switch(self, sample, swap_env_mask, swap_sample_mask, indices):
        """env_ISCO.py:337-344 (at most one position per env, as proposal() produces)."""
        x = sample.clone()
        pos = torch.full((x.shape[0],), -1, dtype=torch.int64, device=self.device)
        pos[swap_env_mask] = swap_sample_mask
        mops.tsp_apply_swap(x, pos, indices.contiguous())
        return x

    def calculate_distance(self, sample):
        """env_ISCO.py:346-350 -> f32 [B]."""
        return mops.tsp_tour_length(self.distance, sample.contiguous())

    def random_gen_init_sample(self, params_dict=None):
        """env_ISCO.py:352-354: batch_size random permutations (Philox Fisher-Yates kernel seeded from torch)."""
        return mops.rand_perms(self.batch_size, self.num_nodes, self._next_seed(), self.device, env_offset=self.env_offset)


def __getattr__(name):
    # the reference keeps ISCO_maxcut in this module too (env_ISCO.py:17-100); here it has a file of its own
    if name == "ISCO_maxcut":
        from .env_ISCO_maxcut import ISCO_maxcut
        return ISCO_maxcut
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
