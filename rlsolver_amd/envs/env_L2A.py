"""EnvMaxcut, L2A / MCPG flavour -- drop-in for rlsolver/envs/env_L2A.py:24-116
(== env_MCPG.py:24-116, the simulators of env_k_spin.py).

Same constructor keywords, attributes, method names, shapes and dtypes; every method that
touches spins launches a HIP kernel (rlsolver_amd.ops).  Differences, all deliberate:

* ``device`` must be a HIP device: there is no CPU path;
* no per-env index tensors are materialised (the reference caches three int64 [B, E'] tensors);
  ``n0_ids`` / ``n1_ids`` keep their [1, E'] form for callers that read them;
* ``adjacency_bool`` (dense N x N) is built lazily on first access instead of in ``__init__``;
* two extra keywords, ``env_offset`` and ``seed`` (rlsolver_amd/seeding.py): the global id of this object's env 0 in a
  batch sharded over ranks, and an optional private seed stream.  Every random draw is keyed by (seed, global env id, ...),
  so rank r of W running envs [r B / W, (r + 1) B / W) with ``env_offset = r B / W`` computes exactly those rows of the
  one-process run.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch as th

from .. import ops
from ..graph import MyGraph, build_csr, calc_num_nodes_in_mygraph, load_mygraph2
from ..seeding import Sharded, seed_from_torch as _seed_from_torch  # noqa: F401  (re-exported: other modules import it from here)

TEN = th.Tensor


class EnvMaxcut(Sharded):
    def __init__(self, sim_name: str = 'max_cut', mygraph: MyGraph = (),
                 device=th.device('cpu'), if_bidirectional: bool = False, num_nodes: int = 0,
                 env_offset: int = 0, seed: Optional[int] = None, group=None):
        self._init_shard(env_offset, seed, group)
        self.device = th.device(device)
        if self.device.type != 'cuda':
            raise TypeError(f"rlsolver_amd.EnvMaxcut needs a HIP device (got {self.device}); "
                            "there is no CPU path")
        self.sim_name = sim_name
        self.int_type = th.long
        self.if_maximize = True
        self.if_bidirectional = if_bidirectional

        mygraph = mygraph if mygraph else load_mygraph2(graph_name=sim_name)
        self._mygraph = mygraph
        # the reference sizes the env by distinct endpoints (util.py:35-40); ``num_nodes`` lets a
        # caller that knows the file header keep isolated nodes
        self.num_nodes = num_nodes if num_nodes else calc_num_nodes_in_mygraph(mygraph)
        self.num_edges = len(mygraph)
        csr = build_csr(mygraph, num_nodes=self.num_nodes, if_bidirectional=if_bidirectional)
        self.graph = ops.DeviceGraph(csr, self.device)  # weights are ignored, as in the reference
        self.n0_ids = self.graph.eu.to(th.long)[None, :]
        self.n1_ids = self.graph.ev.to(th.long)[None, :]
        counts = np.bincount(csr.eu, minlength=self.num_nodes) if csr.eu.size else np.zeros(self.num_nodes, np.int64)
        self.n0_num_n1 = th.from_numpy(counts.astype(np.int64)).to(self.device)[None, :]
        self._adjacency_indies = None
        self._adjacency_bool = None
        self.fused_local_search = True   # False: K2 + torch weights/noise/kthvalue + K6 + K5 as separate launches
        self.force_ls_rounds = False     # True: the threshold / proposal-round kernels even where the fused kernel fits (tests)
        self.force_ls_fused = False      # True: the fused kernel wherever it fits, also for batches of few tiles (tests)

    # ---- lazily built forms of the reference attributes
    @property
    def adjacency_indies(self):
        if self._adjacency_indies is None:
            erp = self.graph.erowptr.cpu().numpy()
            ev = self.graph.ev.to(th.long)
            self._adjacency_indies = [ev[erp[i]:erp[i + 1]] for i in range(self.num_nodes)]
        return self._adjacency_indies

    @property
    def adjacency_bool(self):
        if self._adjacency_bool is None:
            csr = self.graph.csr  # build_adjacency_bool(mygraph, if_bidirectional=True), env_L2A.py:37
            adj = np.zeros((self.num_nodes, self.num_nodes), dtype=bool)
            adj[np.repeat(np.arange(self.num_nodes), np.diff(csr.rowptr)), csr.col] = True
            self._adjacency_bool = th.from_numpy(adj).to(self.device)
        return self._adjacency_bool

    # ---- objective
    def calculate_obj_values(self, xs: TEN, if_sum: bool = True) -> TEN:
        """env_L2A.py:54-66.  int64 [B] (bool [B, E'] when if_sum=False)."""
        if if_sum:
            return ops.maxcut_obj(self.graph, xs)
        values = ops.maxcut_edge_cut_mask(self.graph, xs)
        if self.if_bidirectional:
            values = values // 2
        return values

    def calculate_obj_values_for_loop(self, xs: TEN, if_sum: bool = True) -> TEN:
        """env_L2A.py:68-80: per-node cut degree (int64), summed if asked; float / 2 when
        bidirectional -- the reference's dtype quirk is kept."""
        values = ops.maxcut_node_cutdeg(self.graph, xs)
        if if_sum:
            values = values.sum(dim=1)
        if self.if_bidirectional:
            values = values.float() / 2
        return values

    def generate_xs_randomly(self, num_sims):
        """env_L2A.py:82-85: Bernoulli(1/2) spins, node 0 := 0 (Philox kernel keyed by (seed, env_offset + row))."""
        return ops.rand_spins(num_sims, self.num_nodes, self._next_seed(), self.device, env_offset=self.env_offset)

    # ---- local search
    def local_search_inplace(self, good_xs: TEN, good_vs: TEN,
                             num_iters: int = 8, num_spin: int = 8, noise_std: float = 0.3,
                             noise: Optional[TEN] = None):
        """env_L2A.py:87-116: ``num_iters`` noisy top-``num_spin`` multi-flip proposals (threshold from a first,
        separate draw), then the N-step greedy 'addition' loop; both accept ties.  ``good_vs`` may be a 0-dim tensor
        meaning "compute it" (:91).  ``noise`` (f32 [num_iters + 1, B, N]) replaces the randn_like draws -- test hook."""
        compute_vs = good_vs.shape == ()
        vs = th.empty(good_xs.shape[0], dtype=th.long, device=self.device) if compute_vs else good_vs.long()
        self.local_search_pipeline(good_xs, vs, weight_mult=1, num_iters=num_iters, num_spin=num_spin, noise_std=noise_std,
                                   noise=noise, first_draw_proposes=False, compute_vs=compute_vs)
        return good_xs, vs

    def _num_cus(self) -> int:
        n = getattr(self, "_cus", None)
        if n is None:
            n = self._cus = int(th.cuda.get_device_properties(self.device).multi_processor_count)
        return n

    def local_search_pipeline(self, xs: TEN, vs: TEN, weight_mult: int, num_iters: int, num_spin: int, noise_std: float,
                              noise: Optional[TEN], first_draw_proposes: bool, compute_vs: bool = False) -> None:
        """The common body of local_search_inplace (env_L2A.py:87-116) and LocalSearch.random_search
        (methods/LocalSearch.py:53-83), in place on xs [B, N] bool / vs [B] int64:
            ws = n0_num_n1 - weight_mult * cut-degree;  rd_std = (max_b ws - min_b ws) * noise_std
            thresh = kthvalue(ws + draw_0 * rd_std, N - num_spin)
            proposals mask = (ws + draw_t * rd_std) > thresh for t = 1..num_iters (t = 0.. when first_draw_proposes),
            each kept where the cut does not decrease; then the greedy single-flip sweep.
        One pre-pass + ONE fused kernel when the library covers the shape (rls_maxcut_local_search_supported), else
        K2-weights + torch noise / kthvalue + K6 per round + K5."""
        B = xs.shape[0]
        if noise is not None:
            noise = noise.to(device=self.device, dtype=th.float32).contiguous()
        wdt = ops.ls_weight_dtype(self.graph, weight_mult)
        rounds_can = (noise is None and (self.fused_local_search or self.force_ls_rounds) and wdt != th.int32
                      and ops.ls_rounds_supported(self.graph, num_spin))
        # a batch of few tiles: the round kernels spread each tile's noise passes over several workgroups and beat the fused
        # kernel's one workgroup per tile (G22-sized, 256 - 8192 envs: 0.32 - 0.35 vs 0.39 - 0.41 ms; same result bit for bit)
        # (also where a short row leaves one slice per tile -- G14-sized, 2048 / 8192 envs: 117 / 150 vs 170 / 176 us --: all rounds' mask
        # words are one launch of tiles x rounds workgroups; tools/timing/ls_forms.py)
        few_tiles = (rounds_can and num_iters > 0 and not self.force_ls_fused
                     and (ops.ls_slices(self.graph, B, wdt) > 1 or 2 * ((B + 63) // 64) <= self._num_cus()))
        # rows at the end of the fused kernel's LDS layout (N > ~6200 of <= ~6500): the round kernels win at every batch size
        # (N = 6496, 2^15 / 2^16 envs: 1.58 / 3.17 vs 1.94 / 3.85 ms; N = 6000: 1.46 vs 1.44; tools/timing/ls_forms_n.py)
        if rounds_can and num_iters > 0 and not self.force_ls_fused and self.num_nodes > 6200:
            few_tiles = True
        fused_ok = (self.fused_local_search and not self.force_ls_rounds and not few_tiles
                    and ops.local_search_fusable(self.graph, num_spin, B)
                    and xs.data_ptr() % 16 == 0 and (noise is None or noise.data_ptr() % 16 == 0))   # (views that start mid-row)
        rounds_ok = not fused_ok and rounds_can
        # exact integers (int8 / int16) in both env flavours; the round kernels read them on a 16-byte row pitch (any N)
        fused_now = fused_ok and (num_iters > 0 or not first_draw_proposes)
        ws32, mm = ops.maxcut_ls_weights(self.graph, xs, weight_mult, padded=rounds_ok or fused_now, return_minmax=True)
        mm = self._global_minmax(mm)        # a statistic of the WHOLE batch (env_L2A.py:93-94): reduced over the ranks of a sharded one
        rd_std = (mm[1] - mm[0]).to(th.float32).mul_(float(noise_std))   # f32 whatever torch's default dtype is: float(span) * noise_std
        off = self.env_offset
        if fused_now:
            ops.maxcut_local_search(self.graph, xs, ws32, rd_std, vs, num_iters, num_spin, noise=noise,
                                    seed=0 if noise is not None else self._next_seed(), env_offset=off,
                                    first_draw_proposes=first_draw_proposes, compute_obj=compute_vs)
            return
        if compute_vs:
            ops.maxcut_obj(self.graph, xs, out=vs)
        if rounds_ok:
            # the same steps as kernels with the fused kernel's draws (same seed => the fused kernel's result): graphs too
            # large for the fused kernel's LDS layout
            seed = self._next_seed()
            # small batches: a tile's noise pass over several workgroups, all rounds applied on one load of the tile
            scratch = ops.ls_scratch(self.graph, B, ws32, num_draws=num_iters)
            thresh = ops.maxcut_ls_threshold(self.graph, ws32, rd_std, seed, num_spin, draw=0, env_offset=off, scratch=scratch)
            ops.maxcut_ls_rounds(self.graph, xs, ws32, rd_std, thresh, vs, seed, 0 if first_draw_proposes else 1, num_iters,
                                 env_offset=off, scratch=scratch)
            ops.maxcut_greedy_sweep(self.graph, xs, vs)
            return
        if ws32.shape[1] != self.num_nodes:
            ws32 = ws32[:, :self.num_nodes]
        if noise is not None:
            draws = lambda t: noise[t]                                      # noqa: E731
        else:
            # the kernels' own draws as a tensor (keyed by the global env like them; torch.randn is not): same seed => the result
            # of the fused / round kernels wherever those also cover the shape
            seed = self._next_seed()
            draws = lambda t: ops.maxcut_ls_normals(B, self.num_nodes, seed, t, self.device, env_offset=off)   # noqa: E731
        thresh, t = None, 0
        for it in range(num_iters + (0 if first_draw_proposes else 1)):
            noisy = ws32 + draws(t) * rd_std
            t += 1
            if thresh is None:
                thresh = th.kthvalue(noisy, k=self.num_nodes - num_spin, dim=1)[0][:, None]
                if not first_draw_proposes:
                    continue
            ops.maxcut_propose_accept(self.graph, xs, noisy.gt(thresh), vs)
        ops.maxcut_greedy_sweep(self.graph, xs, vs)
