"""ISCO_TSP -- drop-in for rlsolver/envs/env_ISCO.py:176-363 (TSP sampler) on a HIP device.

The objective pieces run in HIP kernels: ``calculate_distance`` (K12), the delta part of ``opt_2``
(K13: inverse permutation, ban mask, 3-case swap delta -- replacing sort + searchsorted + a
[B, N, N-1] 3-D gather) and ``switch``.  The sampler's softmax / Gumbel-top-k / MH bookkeeping
(``proposal``, ``y2x``, ``step``) is the reference's own [B, N] torch arithmetic.

Unlike the reference, which reads BATCH_SIZE / K / DEVICE from star-imported config modules at call
time, they are explicit constructor arguments here.
"""
from __future__ import annotations

import torch

from .. import ops_mcpg_tsp as mops
from ..methods.util import mh_step, multinomial, noreplacement_sampling_renormalize


class ISCO_TSP:
    def __init__(self, params_dict, batch_size: int = 1, K: int = 20, device=None, chain_length: int = 10000,
                 init_temperature: float = 1.0, final_temperature: float = 0.1):
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

    # ---- sampler loop (reference arithmetic, env_ISCO.py:188-236)
    def step(self, x, path_length, temperature):
        cur_x = x.clone()
        traj = torch.zeros((self.batch_size, 3, path_length), dtype=torch.float, device=self.device)
        for i in range(path_length):
            cur_x, logits, trajectory, delta_yx = self.proposal(cur_x, temperature)
            ll_x2y = trajectory['ll_x2y']
            ll_y2x = self.y2x(logits, trajectory)
            traj[:, 0, i], traj[:, 1, i], traj[:, 2, i] = delta_yx, -ll_x2y, ll_y2x
        log_acc = torch.clamp(torch.sum(traj, dim=(1, 2)), max=0.0)
        y, accepted = self.select_sample(log_acc, x, cur_x)
        return y, torch.mean(log_acc.exp())

    def proposal(self, sample, temperature):
        x = sample.clone()
        logits, log_prob, indices, ban_mask, delta_yx = self.get_local_dist(x, temperature)
        selected_idx, ll_selected = multinomial(log_prob, torch.ones(self.batch_size, dtype=torch.int64,
                                                                     device=self.device))
        logits = logits * (1 - 2 * selected_idx['selected_mask'])
        swap_env_mask, swap_sample_mask = torch.where(((selected_idx['selected_mask'] == 1) & (~ban_mask)) == 1)
        x = self.switch(sample, swap_env_mask, swap_sample_mask, indices)
        trajectory = {'ll_x2y': torch.sum(ll_selected, dim=-1), 'selected_idx': selected_idx}
        return x, logits, trajectory, torch.sum(delta_yx * selected_idx['selected_mask'], dim=-1)

    def get_local_dist(self, sample, temperature):
        x = sample.detach()
        logratio, indices, ban_mask = self.opt_2(x, temperature)
        logratio[ban_mask] = -1e6
        logits = self.apply_weight_function_logscale(logratio)
        log_prob = torch.nn.functional.log_softmax(logits, dim=-1)
        return logits, log_prob, indices, ban_mask, logratio

    def y2x(self, logits, forward_trajectory):
        log_prob = torch.nn.functional.log_softmax(logits, dim=-1)
        selected_mask = forward_trajectory['selected_idx']['selected_mask']
        order_info = forward_trajectory['selected_idx']['perturbed_ll']
        backwd_idx = torch.argsort(order_info, dim=-1)
        log_prob = torch.where(selected_mask.bool(), log_prob, torch.tensor(-1e18, device=self.device))
        backwd_ll = torch.gather(log_prob, dim=-1, index=backwd_idx)
        backwd_mask = torch.gather(selected_mask, dim=-1, index=backwd_idx)
        ll_backwd = noreplacement_sampling_renormalize(backwd_ll)
        return torch.sum(torch.where(backwd_mask.bool(), ll_backwd, torch.tensor(0.0, device=self.device)), dim=-1)

    # ---- hot path
    def draw_partners(self, sample):
        """First half of opt_2 (env_ISCO.py:246-266): the partner CITY for every position, drawn with
        torch's generator exactly like the reference (rand, randint K, randint N-K-1) but with 2-D
        gathers instead of materialising nearest_indices[sample] / random_indices[sample]."""
        B, N, K = sample.shape[0], self.num_nodes, self.K
        rand_numbers = torch.rand(B, N, device=self.device)
        condition = rand_numbers < (K / (K + 1))
        nearest_rand = torch.randint(0, K, (B, N), device=self.device)
        random_rand = torch.randint(0, N - K - 1, (B, N), device=self.device)
        near = self.nearest_indices[sample, nearest_rand]
        rnd = self.random_indices[sample, random_rand]
        return torch.where(condition, near, rnd)

    def opt_2(self, sample, temperature, selected=None):
        """env_ISCO.py:238-335 -> (-delta/T f32 [B,N], indices int64 [B,N], ban bool [B,N])."""
        sample = sample.contiguous()
        if selected is None:
            selected = self.draw_partners(sample)
        return mops.tsp_swap_delta_all(self.distance, sample, selected.contiguous(), float(temperature))

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
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        return mops.rand_perms(self.batch_size, self.num_nodes, seed, self.device)

    def select_sample(self, log_acc, x, y):
        y, accepted = mh_step(log_acc, x, y)
        return y, accepted

    def apply_weight_function_logscale(self, logratio):
        return logratio / 2
