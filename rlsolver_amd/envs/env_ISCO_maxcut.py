"""ISCO_maxcut -- drop-in for rlsolver/envs/env_ISCO.py:10-92 (the MaxCut ISCO sampler).

The two quantities the reference gets from ``vmap(model)`` + ``autograd.grad`` (env_ISCO.py:51-63,
79-86) are closed-form on a graph and come from HIP kernels here:
    energy_x[b]          = cut(x_b) / T                          (K1, rls_maxcut_obj)
    score_change_x[b, i] = (1 - 2x_i) * dE/dx_i / 2 = delta_i / (2T)   (K3, rls_maxcut_delta_all)
where delta_i is the cut gain of flipping node i.  Samples keep the reference's dtype/shape
(float32 0/1, [B, N]); the path-length / Gumbel bookkeeping is the reference's [B, N] torch math.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from ..graph import build_csr
from ..methods.util import mh_step, multinomial, noreplacement_sampling_renormalize


class ISCO_maxcut:
    def __init__(self, params_dict, batch_size: int = 1, device=None, chain_length: int = 200,
                 init_temperature: float = 1.0, final_temperature: float = 0.0):
        self.edge_from = params_dict['edge_from']
        self.edge_to = params_dict['edge_to']
        self.device = torch.device(device) if device is not None else self.edge_from.device
        if self.device.type != 'cuda':
            raise TypeError(f"rlsolver_amd.ISCO_maxcut needs a HIP device (got {self.device}); there is no CPU path")
        self.batch_size = batch_size
        self.chain_length = chain_length
        self.init_temperature = torch.tensor(init_temperature, device=self.device)
        self.final_temperature = torch.tensor(final_temperature, device=self.device)
        self.max_num_nodes = params_dict['num_nodes']
        self.num_edges = params_dict['num_edges']
        eu = self.edge_from.detach().cpu().numpy().astype(np.int64)
        ev = self.edge_to.detach().cpu().numpy().astype(np.int64)
        csr = build_csr((eu, ev, np.ones_like(eu)), num_nodes=self.max_num_nodes, if_bidirectional=False)
        self.graph = ops.DeviceGraph(csr, self.device)

    def random_gen_init_sample(self, params_dict=None):
        return torch.bernoulli(torch.full((self.batch_size, self.max_num_nodes), 0.5, device=self.device))

    def step(self, x, path_length, temperature):
        ll_x, y, trajectory = self.proposal(x, path_length, temperature)
        ll_x2y = trajectory['ll_x2y']
        ll_y, ll_y2x = self.ll_y2x(trajectory, y, temperature)
        log_acc = torch.clamp(ll_y + ll_y2x - ll_x - ll_x2y, max=0.0)
        y = self.select_sample(log_acc, x, y)
        return y, ll_y * temperature, log_acc.exp()

    def proposal(self, x, path_length, temperature):
        ll_x, log_prob = self.get_local_dist(x, temperature)
        selected_idx, ll_selected = multinomial(log_prob, path_length)
        mask = selected_idx['selected_mask']
        y = x * (1 - mask) + mask * (1 - x)
        return ll_x, y, {'ll_x2y': torch.sum(ll_selected, dim=-1), 'selected_idx': selected_idx}

    def get_local_dist(self, sample, temperature):
        """-> (energy f32 [B], log_prob f32 [B, N])"""
        xb = (sample > 0).contiguous()
        t = float(temperature)
        energy_x = ops.maxcut_obj(self.graph, xb).to(torch.float32) / t
        score_change_x = ops.maxcut_delta_all(self.graph, xb).to(torch.float32) / (2.0 * t)
        return energy_x, torch.log_softmax(score_change_x, dim=-1)

    def ll_y2x(self, forward_trajectory, y, temperature):
        ll_y, log_prob = self.get_local_dist(y, temperature)
        selected_mask = forward_trajectory['selected_idx']['selected_mask']
        order_info = forward_trajectory['selected_idx']['perturbed_ll']
        backwd_idx = torch.argsort(order_info, dim=-1)
        log_prob = torch.where(selected_mask.bool(), log_prob, torch.tensor(-1e18, device=self.device))
        backwd_ll = torch.gather(log_prob, dim=-1, index=backwd_idx)
        backwd_mask = torch.gather(selected_mask, dim=-1, index=backwd_idx)
        ll_backwd = noreplacement_sampling_renormalize(backwd_ll)
        ll_y2x = torch.sum(torch.where(backwd_mask.bool(), ll_backwd, torch.tensor(0.0, device=self.device)), dim=-1)
        return ll_y, ll_y2x

    def model(self, sample, temperature):
        """energy of a batch (the reference vmaps a per-sample version): #cut / T."""
        xb = (sample > 0).contiguous()
        if xb.dim() == 1:
            xb = xb[None, :]
        return ops.maxcut_obj(self.graph, xb).to(torch.float32) / float(temperature)

    def select_sample(self, log_acc, x, y):
        y, acc = mh_step(log_acc, x, y)
        return y
