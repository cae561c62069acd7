"""Tensor-level wrappers for the MCPG and TSP entry points (same conventions as ops.py: outputs allocated here, the work
goes through ``torch.ops.rlsolver_hip.*``)."""
from __future__ import annotations

from typing import Optional

import torch

from .ops import DeviceGraph, _check, _ptr, _s64, _stream, _t  # noqa: F401  (_ptr / _stream: raw C-ABI calls in tests)

TEN = torch.Tensor
_NM_DTYPES = (torch.float32, torch.uint8, torch.bool)


# ------------------------------------------------------------------------------ MCPG
class PackedChains:
    """C chains over N nodes, bit-packed tile-major: ``words`` int64 [ceil(C / 64), N] (the uint64 bit patterns; word
    (t, n) bit e = node n of chain 64 t + e).  include/rlsolver_hip.h, "Layouts of a batch of C chains"."""

    __slots__ = ("words", "num_chains")

    def __init__(self, words: TEN, num_chains: int):
        _check(words, "words", (torch.int64,))
        if words.dim() != 2 or words.shape[0] != (num_chains + 63) // 64:
            raise ValueError(f"words must be [ceil({num_chains} / 64), N]")
        self.words, self.num_chains = words, num_chains

    @property
    def num_nodes(self) -> int:
        return self.words.shape[1]

    @property
    def device(self):
        return self.words.device

    @classmethod
    def empty(cls, num_nodes: int, num_chains: int, device):
        return cls(torch.empty(((num_chains + 63) // 64, num_nodes), dtype=torch.int64, device=device), num_chains)

    @classmethod
    def pack(cls, xs: TEN):
        """node-major [N, C] float32 / uint8 / bool -> PackedChains (one kernel)."""
        _check(xs, "xs", _NM_DTYPES)
        N, Cc = xs.shape
        out = cls.empty(N, Cc, xs.device)
        _t.mcpg_pack_chains(xs, out.words)
        return out

    def unpack(self) -> TEN:
        """-> node-major float32 [N, C] holding 0.0 | 1.0 (the reference's surface)."""
        xs = torch.empty((self.num_nodes, self.num_chains), dtype=torch.float32, device=self.device)
        _t.mcpg_unpack_chains(self.words, self.num_chains, xs)
        return xs

    def clone(self):
        return PackedChains(self.words.clone(), self.num_chains)


def _chains(x, name):
    """(pointer tensor, spin_bytes, N, C) of a node-major tensor or a PackedChains."""
    if isinstance(x, PackedChains):
        return x.words, 0, x.num_nodes, x.num_chains
    _check(x, name, _NM_DTYPES)
    if x.dim() != 2:
        raise ValueError(f"{name} must be [N, C]")
    return x, (4 if x.dtype == torch.float32 else 1), x.shape[0], x.shape[1]


def _chain_ids(chain_ids) -> tuple:
    """(offset, period, skip) of include/rlsolver_hip.h: rls_chain_ids; None = the single-process numbering."""
    if chain_ids is None:
        return (0, 0, 0)
    off, per, skip = (int(v) for v in chain_ids)
    return (off, per, skip)


def mcpg_metro_rounds(samples, probs: TEN, T: int, index: Optional[TEN] = None, u: Optional[TEN] = None,
                      seed: int = 0, t_limit: Optional[TEN] = None, write_back: bool = True,
                      accepts: Optional[TEN] = None, t_offset: int = 0, samples_in=None, chain_ids=None, scratch: Optional[TEN] = None) -> None:
    """K9 (see include/rlsolver_hip.h).  samples [N, C] f32/uint8 or PackedChains, in place, or read from
    ``samples_in`` (same layout; a PackedChains with fewer chains -- a multiple of 64 -- is broadcast) and written to
    ``samples``.  ``chain_ids`` = (offset, period, skip): the global ids of a shard's chains (rls_chain_ids)."""
    st, sb, N, Cc = _chains(samples, "samples")
    dev = st.device
    _check(probs, "probs", (torch.float32,), dev, (N,))
    if (index is None) != (u is None):
        raise ValueError("index and u must be given together")
    if index is not None:
        _check(index, "index", (torch.int64,), dev)
        _check(u, "u", (torch.float32,), dev)
        if index.shape[0] < t_offset + T or u.shape[0] < t_offset + T or index.shape[1:] != (Cc,) or u.shape[1:] != (Cc,):
            raise ValueError("index/u must be [>= t_offset + T, C]")
    if t_limit is not None:
        _check(t_limit, "t_limit", (torch.int64,), dev)
    accept_rows = 1
    if accepts is not None:     # [T] or [rows, T]: workgroups spread their adds over the rows, a round's count = its column sum
        _check(accepts, "accepts", (torch.int64,), dev)
        if accepts.dim() == 2:
            accept_rows = accepts.shape[0]
        if accepts.dim() not in (1, 2) or accepts.shape[-1] != T:
            raise ValueError("accepts must be [T] or [rows, T]")
    sin, c_in = None, Cc
    if samples_in is not None:
        sin, sb_in, n_in, c_in = _chains(samples_in, "samples_in")
        if sb_in != sb or n_in != N or (sb != 0 and (c_in != Cc or sin.dtype != st.dtype)) or sin.device != dev:
            raise ValueError("samples_in must have the layout and node count of samples")
        if not write_back:
            raise ValueError("samples_in needs write_back")
    if scratch is None and sb == 0:     # the packed walk's draw windows (include/rlsolver_hip.h: rls_mcpg_metro_scratch_bytes)
        scratch = metro_scratch(N, Cc, dev)
    _t.mcpg_metro_rounds(st, sin, c_in, Cc, probs, T, t_offset, index, u, _s64(seed), t_limit, bool(write_back), accepts,
                         *_chain_ids(chain_ids), scratch)


def metro_scratch(N: int, C: int, device) -> Optional[TEN]:
    """The scratch the bit-packed walk wants at this size (None: it keeps its draw windows in LDS).  A caller that walks in chunks
    allocates it once and passes it to every mcpg_metro_rounds call."""
    from . import _abi
    need = int(_abi.lib().rls_mcpg_metro_scratch_bytes(int(N), int(C)))
    return torch.empty(need, dtype=torch.uint8, device=device) if need > 0 else None


def mcpg_metro_max_rounds(N: int, spin_bytes: int) -> int:
    """Rounds one mcpg_metro_rounds launch can take with accept counts in this layout (0: N does not fit at all)."""
    from . import _abi
    return int(_abi.lib().rls_mcpg_metro_max_rounds(int(N), int(spin_bytes)))


def mcpg_metro_stop(accepts: TEN, target: int, first: int, next_T: int, ctl: TEN, apply_limit: Optional[TEN] = None) -> None:
    """The stop rule between two chunks of metro rounds (include/rlsolver_hip.h: rls_mcpg_metro_stop).  accepts int64 [rows, T],
    ctl int64 [3] = {accepts so far, live, limit of the next dry pass} (in / out), apply_limit int64 [1] (out).  first: 1 the
    call's first chunk, 2 a later chunk applied directly (inside the first T rounds), 0 a dry pass."""
    dev = accepts.device
    _check(accepts, "accepts", (torch.int64,), dev)
    _check(ctl, "ctl", (torch.int64,), dev, (3,))
    if apply_limit is not None:
        _check(apply_limit, "apply_limit", (torch.int64,), dev, (1,))
    _t.mcpg_metro_stop(accepts, int(target), int(first), int(next_T), ctl, apply_limit)


def mcpg_local_search(g: DeviceGraph, xs_in: TEN, order: TEN, num_ls: int, uniforms: Optional[TEN] = None,
                      seed: int = 0, visit_stream: Optional[TEN] = None, edge_weights: Optional[TEN] = None,
                      gauge_node: int = -1, chain_ids=None):
    """K7 + expected cut.  Returns (xs_out f32 [N, C], expected f32 [C]).  ``edge_weights`` int32 [E'] (in the order of
    the graph's stored edges) + a weighted visit stream select the weighted sampler; ``gauge_node`` >= 0 XORs every
    chain with its value at that node first (rlsolver/methods/MCPG/sampling.py:101-104)."""
    _check(xs_in, "xs_in", _NM_DTYPES, g.device)
    if xs_in.dim() != 2 or xs_in.shape[0] != g.num_nodes:
        raise ValueError(f"xs_in must be [{g.num_nodes}, C]")
    Cc = xs_in.shape[1]
    _check(order, "order", (torch.int32,), g.device, (g.num_nodes,))
    if uniforms is not None:
        _check(uniforms, "uniforms", (torch.float32,), g.device, (num_ls, g.num_nodes, Cc))
    xs_out = torch.empty((g.num_nodes, Cc), dtype=torch.float32, device=g.device)
    expected = torch.empty(Cc, dtype=torch.float32, device=g.device)
    if visit_stream is not None:
        _check(visit_stream, "visit_stream", (torch.int32,), g.device)
    if edge_weights is not None:
        _check(edge_weights, "edge_weights", (torch.int32,), g.device, (g.num_stored_edges,))
    _t.mcpg_local_search(g.handle, xs_in, xs_out, order, visit_stream, num_ls, uniforms, _s64(seed), edge_weights, int(gauge_node), expected,
                         *_chain_ids(chain_ids))
    return xs_out, expected


def mcpg_local_search_levels(g: DeviceGraph, xs_in, lv_ptr: TEN, lv_data: TEN, num_ls: int, seed: int = 0,
                             coins: Optional[TEN] = None, out=None, num_chains: Optional[int] = None, chain_ids=None):
    """K7 + expected cut on the level-parallel schedule (rls_mcpg_visit_levels).  ``xs_in``: node-major [N, C] or
    PackedChains (which may hold fewer chains than ``num_chains``, a multiple of 64: broadcast).  ``coins`` int64 (bit
    pattern of uint64) [num_ls * N, ceil(C / 64)]: the tie coins "u < 1/2" -- test hook; None = counter hash keyed by
    seed.  ``out``: a PackedChains to write (may be xs_in itself) or None = node-major f32.
    Returns (xs_out, expected f32 [C])."""
    st, sb, N, c_in = _chains(xs_in, "xs_in")
    if st.device != g.device or N != g.num_nodes:
        raise ValueError(f"xs_in must hold {g.num_nodes} nodes on {g.device}")
    Cc = c_in if num_chains is None else num_chains
    _check(lv_ptr, "lv_ptr", (torch.int32,), g.device)
    _check(lv_data, "lv_data", (torch.int32,), g.device)
    if coins is not None:
        _check(coins, "coins", (torch.int64,), g.device, (num_ls * g.num_nodes, (Cc + 63) // 64))
    if out is None:
        xs_out = torch.empty((g.num_nodes, Cc), dtype=torch.float32, device=g.device)
        ot, osb = xs_out, 4
    else:
        if not isinstance(out, PackedChains) or out.num_chains != Cc or out.num_nodes != N or out.device != g.device:
            raise ValueError("out must be a PackedChains of num_chains chains")
        xs_out, ot, osb = out, out.words, 0
    expected = torch.empty(Cc, dtype=torch.float32, device=g.device)
    _t.mcpg_local_search_levels(g.handle, st, c_in, ot, Cc, lv_ptr, lv_data, num_ls, coins, _s64(seed), expected,
                                *_chain_ids(chain_ids))
    return xs_out, expected


def mcpg_pick_best(expected: TEN, xs, total_mcmc_num: int, repeat_times: int, num_edges: int):
    """K8 second half.  Returns (best_index int64 [M], vs_good f32 [M], xs_good: f32 [N, M] or PackedChains of M chains)."""
    st, sb, N, Cc = _chains(xs, "xs")
    if sb == 1:
        raise TypeError("xs must be float32 node-major or PackedChains")
    dev = st.device
    if Cc != total_mcmc_num * repeat_times:
        raise ValueError("xs must hold total_mcmc_num * repeat_times chains")
    _check(expected, "expected", (torch.float32,), dev, (Cc,))
    idx = torch.empty(total_mcmc_num, dtype=torch.int64, device=dev)
    vs = torch.empty(total_mcmc_num, dtype=torch.float32, device=dev)
    if sb == 0:
        xg = PackedChains.empty(N, total_mcmc_num, dev)
        xgt = xg.words
    else:
        xg = xgt = torch.empty((N, total_mcmc_num), dtype=torch.float32, device=dev)
    _t.mcpg_pick_best(expected, st, N, total_mcmc_num, repeat_times, num_edges, idx, vs, xgt)
    return idx, vs, xg


def mcpg_merge_best(temp_max: TEN, temp_info: PackedChains, now_max_res: TEN, now_info: PackedChains, replace_worst: bool = True):
    """The best-merge of methods/MCPG.py:376-391 in place on the device.  Returns (best value f32 [1], its chain int64 [1]);
    with ``replace_worst=False`` (a shard of the kept chains) only the per-chain merge is applied and the results are
    [2] = {max, min} of now_max_res and their first chains."""
    M, N, dev = temp_info.num_chains, temp_info.num_nodes, temp_info.device
    _check(temp_max, "temp_max", (torch.float32,), dev, (M,))
    _check(now_max_res, "now_max_res", (torch.float32,), dev, (M,))
    if now_info.num_chains != M or now_info.num_nodes != N:
        raise ValueError("now_info must match temp_info")
    mask = torch.empty((M + 63) // 64, dtype=torch.int64, device=dev)
    bv = torch.empty(1 if replace_worst else 2, dtype=torch.float32, device=dev)
    bi = torch.empty(1 if replace_worst else 2, dtype=torch.int64, device=dev)
    _t.mcpg_merge_best(temp_max, temp_info.words, now_max_res, now_info.words, M, mask, bv, bi, bool(replace_worst))
    return bv, bi


def mcpg_value_bit_sums(samples: PackedChains, value: TEN) -> TEN:
    """A[n] = sum_c value[c] * s[n, c]  (f32 [N]); see rls_mcpg_value_bit_sums."""
    _check(value, "value", (torch.float32,), samples.device, (samples.num_chains,))
    A = torch.zeros(samples.num_nodes, dtype=torch.float32, device=samples.device)
    _t.mcpg_value_bit_sums(samples.words, samples.num_chains, value, A)
    return A


# ------------------------------------------------------------------------------ TSP
def _perm(p: TEN, name="perm"):
    _check(p, name, (torch.int64,))
    if p.dim() != 2:
        raise ValueError(f"{name} must be [B, N]")
    return p.shape


def tsp_tour_length(dist: TEN, perm: TEN) -> TEN:
    B, N = _perm(perm)
    _check(dist, "dist", (torch.float32,), perm.device, (N, N))
    out = torch.empty(B, dtype=torch.float32, device=perm.device)
    _t.tsp_tour_length(dist, perm, out)
    return out


def tsp_tables8(nearest: TEN, random: TEN) -> Optional[TEN]:
    """The two neighbour tables of ISCO_TSP as the byte block K13 keeps in LDS (include/rlsolver_hip.h: rls_tsp_tables8_bytes):
    uint8 [N, K] then the first N - K - 1 columns of ``random`` as uint8, each zero-padded to 16 bytes; None when N > 256."""
    from . import _abi
    N, K = int(nearest.shape[0]), int(nearest.shape[1])
    nb = int(_abi.lib().rls_tsp_tables8_bytes(N, K))
    if nb == 0:
        return None
    out = torch.zeros(nb, dtype=torch.uint8, device=nearest.device)
    a = (N * K + 15) // 16 * 16
    out[:N * K] = nearest.reshape(-1).to(torch.uint8)
    out[a:a + N * (N - K - 1)] = random[:, :N - K - 1].reshape(-1).to(torch.uint8)
    return out


def tsp_swap_delta_all(dist: TEN, perm: TEN, selected: Optional[TEN], temperature: float, nearest: Optional[TEN] = None,
                       random: Optional[TEN] = None, near_threshold: float = 0.0, seed: int = 0, env_offset: int = 0,
                       return_selected: bool = False, tables8: Optional[TEN] = None):
    """K13, ISCO_TSP.opt_2 (env_ISCO.py:238-335) -> (logratio f32, indices int64, ban bool), all [B, N].
    ``selected`` int64 [B, N]: the partner cities, given (the recorded-draw hook of the golden tests); None (production): drawn
    in the kernel from (seed, env_offset + b, position) through ``nearest`` int32 [N, K] / ``random`` int32 [N, >= N - K - 1]
    with ``near_threshold`` = K / (K + 1) -- ``return_selected`` appends the drawn cities (int64 [B, N]); ``tables8`` =
    ``tsp_tables8(nearest, random)``, built once by the caller, lets the kernel keep both tables in LDS (N <= 256)."""
    B, N = _perm(perm)
    dev = perm.device
    _check(dist, "dist", (torch.float32,), dev, (N, N))
    sel_out = None
    if selected is not None:
        _check(selected, "selected", (torch.int64,), dev, (B, N))
        if return_selected:
            raise ValueError("return_selected records the in-kernel draw: it needs selected=None")
    else:
        if nearest is None or random is None:
            raise ValueError("selected=None draws the partners in the kernel: nearest / random tables must be given")
        _check(nearest, "nearest", (torch.int32,), dev)
        _check(random, "random", (torch.int32,), dev)
        if return_selected:
            sel_out = torch.empty((B, N), dtype=torch.int64, device=dev)
    logratio = torch.empty((B, N), dtype=torch.float32, device=dev)
    indices = torch.empty((B, N), dtype=torch.int64, device=dev)
    ban = torch.empty((B, N), dtype=torch.bool, device=dev)
    _t.tsp_swap_delta_all(dist, perm, selected, nearest, random, tables8, float(near_threshold), _s64(seed), int(env_offset), sel_out,
                          float(temperature), logratio, indices, ban)
    return (logratio, indices, ban, sel_out) if return_selected else (logratio, indices, ban)


def tsp_apply_swap(perm: TEN, pos: TEN, indices: TEN) -> None:
    B, N = _perm(perm)
    _check(pos, "pos", (torch.int64,), perm.device, (B,))
    _check(indices, "indices", (torch.int64,), perm.device, (B, N))
    _t.tsp_apply_swap(perm, pos, indices)


def tsp_2opt_delta(dist: TEN, perm: TEN, i: TEN, j: TEN) -> TEN:
    B, N = _perm(perm)
    dev = perm.device
    _check(dist, "dist", (torch.float32,), dev, (N, N))
    _check(i, "i", (torch.int64,), dev, (B,))
    _check(j, "j", (torch.int64,), dev, (B,))
    out = torch.empty(B, dtype=torch.float32, device=dev)
    _t.tsp_2opt_delta(dist, perm, i, j, out)
    return out


def rand_perms(B: int, N: int, seed: int, device, env_offset: int = 0) -> TEN:
    device = torch.device(device)
    if device.type != "cuda":
        raise TypeError("rand_perms needs a HIP device")
    out = torch.empty((B, N), dtype=torch.int64, device=device)
    _t.rand_perms(out, _s64(seed), env_offset)
    return out


def tsp_2opt_best(dist64: TEN, perm: TEN, cur_length: Optional[TEN] = None, slices: Optional[int] = None):
    """One best-improvement 2-opt pass per tour (rls_tsp_2opt_best): -> (best_i, best_j int64 [B], best_value f64 [B]).
    ``cur_length`` f64 [B]: candidates ranked by their whole length summed as the reference's distance_calc does (its own
    comparison values); best_value = the best candidate's length, or cur_length and (-1, -1) where none is shorter.
    Without it: ranked by the O(1) reversal delta (symmetric dist); best_value = the most negative delta, or 0.
    ``slices`` workgroups share one tour's candidates (default: enough to put ~2 workgroups on every CU)."""
    dev = perm.device
    _check(dist64, "dist", (torch.float64,), dev)
    _check(perm, "perm", (torch.int64,), dev)
    B, N = perm.shape
    if dist64.shape != (N, N):
        raise ValueError(f"dist must be [{N}, {N}]")
    if cur_length is not None:
        _check(cur_length, "cur_length", (torch.float64,), dev, (B,))
    if slices is None:   # ~512 workgroups in all; a workgroup gets >= 256 candidates (whole-length ranking: O(N) each) or >= 8192 (deltas)
        per_wg = 256 if cur_length is not None else 8192
        slices = max(1, min(512 // max(B, 1), (N * (N - 1) // 2 + per_wg - 1) // per_wg))
    bi = torch.empty(slices * B, dtype=torch.int64, device=dev)
    bj = torch.empty(slices * B, dtype=torch.int64, device=dev)
    bv = torch.empty(slices * B, dtype=torch.float64, device=dev)
    _t.tsp_2opt_best(dist64, perm, cur_length, bi, bj, bv)
    return bi[:B], bj[:B], bv[:B]
