"""ISCO_maxcut -- drop-in for rlsolver/envs/env_ISCO.py:10-92 (the MaxCut ISCO sampler).

``step`` is ONE kernel (rls_isco_maxcut_step): energies and single-flip scores of x and of the proposal in
closed form (the reference gets them from ``vmap(model)`` + ``autograd.grad``, env_ISCO.py:51-63), the Gumbel
top-k selection of ``path_length`` nodes, the forward / reverse path log-probabilities and the Metropolis accept.
    energy_x[b]          = cut(x_b) / T
    score_change_x[b, i] = (1 - 2 x_i) * dE/dx_i / 2 = gain_i / (2T),  gain_i = cut gain of flipping node i
Samples keep the reference's dtype / shape (float32 0/1, [B, N]).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .. import _abi, ops
from ..graph import build_csr
from ..ops import _check, _s64, _t
from ..seeding import Sharded


class ISCO_maxcut(Sharded):
    def __init__(self, params_dict, batch_size: int = 1, device=None, chain_length: int = 200,
                 init_temperature: float = 1.0, final_temperature: float = 0.0, env_offset: int = 0, seed: Optional[int] = None):
        """``env_offset`` / ``seed``: rlsolver_amd/seeding.py (``batch_size`` samples whose global ids start at env_offset)."""
        self._init_shard(env_offset, seed)
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
        """env_ISCO.py:22-25: Bernoulli(1/2) samples as float32 (Philox kernel seeded from torch)."""
        bits = ops.rand_spins(self.batch_size, self.max_num_nodes, self._next_seed(), self.device, env_offset=self.env_offset)
        # no gauge fixing here: node 0 is a coin like the others (bit 0 of a second keyed draw's node 1)
        bits[:, 0] = ops.rand_spins(self.batch_size, 2, self._next_seed(), self.device, env_offset=self.env_offset)[:, 1]
        return bits.to(torch.float32)

    def _step_scratch(self, B: int):
        """The step's scratch for rows past the LDS (N > ~15 900: rls_isco_maxcut_scratch_bytes), kept between steps; else None."""
        need = int(_abi.lib().rls_isco_maxcut_scratch_bytes(self.graph.ref, int(B)))
        if need == 0:
            return None
        have = getattr(self, "_scratch", None)
        if have is None or have.numel() < need:
            have = self._scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        return have

    def step(self, x, path_length, temperature, draws: Optional[dict] = None, want_terms: bool = False):
        """env_ISCO.py:26-35 -> (y f32 [B, N], ll_y * temperature f32 [B], acceptance probability f32 [B]).

        ``draws`` (test hook) = {"u_gumbel": f32 [B, N], "u_accept": f32 [B]}, the reference's two torch.rand
        draws.  ``want_terms`` additionally returns (terms f32 [B, 5] = ll_x, ll_x2y, ll_y, ll_y2x, log_acc;
        mask bool [B, N])."""
        x = _check(x.contiguous(), "x", (torch.float32,), self.device)
        B, N = x.shape
        if N != self.max_num_nodes:
            raise ValueError(f"x must be [B, {self.max_num_nodes}]")
        pl = torch.as_tensor(path_length, device=self.device).to(torch.int64).expand(B).contiguous()
        y = torch.empty_like(x)
        energy = torch.empty(B, dtype=torch.float32, device=self.device)
        acc = torch.empty(B, dtype=torch.float32, device=self.device)
        terms = torch.empty((B, 5), dtype=torch.float32, device=self.device) if want_terms else None
        mask = torch.empty((B, N), dtype=torch.bool, device=self.device) if want_terms else None
        ug = ua = None
        if draws is not None:
            ug = _check(draws["u_gumbel"].to(self.device).contiguous(), "u_gumbel", (torch.float32,), self.device, (B, N))
            ua = _check(draws["u_accept"].to(self.device).contiguous(), "u_accept", (torch.float32,), self.device, (B,))
        _t.isco_maxcut_step(self.graph.handle, x, y, pl, float(temperature), ug, ua,
                            _s64(0 if draws is not None else self._next_seed()), self.env_offset, energy, acc, terms, mask,
                            self._step_scratch(B))
        if want_terms:
            return y, energy, acc, terms, mask
        return y, energy, acc

    def get_local_dist(self, sample, temperature):
        """env_ISCO.py:51-63 -> (energy f32 [B], log_prob f32 [B, N])  (K1 + K3)."""
        xb = (sample > 0).contiguous()
        t = float(temperature)
        energy_x = ops.maxcut_obj(self.graph, xb).to(torch.float32) / t
        score_change_x = ops.maxcut_delta_all(self.graph, xb).to(torch.float32) / (2.0 * t)
        return energy_x, torch.log_softmax(score_change_x, dim=-1)

    def model(self, sample, temperature):
        """energy of a batch (the reference vmaps a per-sample version, env_ISCO.py:79-86): #cut / T."""
        xb = (sample > 0).contiguous()
        if xb.dim() == 1:
            xb = xb[None, :]
        return ops.maxcut_obj(self.graph, xb).to(torch.float32) / float(temperature)
