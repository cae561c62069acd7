"""True 2-opt local search -- drop-in for rlsolver/methods_problem_specific/TSP/opt_2.py:17-57.

    distance_calc(distance_matrix, city_tour) -> float                 (the reference's sequential float64 sum)
    local_search_2_opt(distance_matrix, city_tour, recursive_seeding=-1, verbose=True) -> (route, distance)
    local_search_2_opt_batch(distance_matrix, perms, max_passes=-1) -> (perms, lengths)      many tours at once

``city_tour = [route, distance]`` with ``route`` the closed tour as 1-based city numbers (first city repeated at the end),
as in the reference.  A pass of the reference tries every reversal [i..j] of the pass's seed tour, recomputes the whole
tour length of every candidate in Python and keeps the best; here a pass is ONE kernel (rls_tsp_2opt_best, a workgroup per
tour) and a gather.  Two rankings of the candidates:

  exact (local_search_2_opt, and the batch form with exact=True): each candidate's whole length, summed in float64 edge
        after edge as distance_calc does -- the reference's own comparison values, so routes AND distances are
        bit-identical to the reference's, ties and rounding included, for any matrix (symmetric or not);
  delta (batch form, default): the O(1) reversal delta in float64 -- N times less work per pass, symmetric matrices only;
        candidates whose lengths agree to rounding may rank differently than under the reference's re-summation.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops_mcpg_tsp as mops


def distance_calc(distance_matrix, city_tour):
    """opt_2.py:17-22 on the host: sum of D[t_k - 1, t_{k+1} - 1] over the closed tour, sequentially in float64."""
    d = np.asarray(distance_matrix.detach().cpu() if torch.is_tensor(distance_matrix) else distance_matrix, dtype=np.float64)
    t = [int(c) for c in city_tour[0]]
    total = 0
    for k in range(len(t) - 1):
        total = total + d[t[k] - 1, t[k + 1] - 1]
    return total


def _apply_reversals(perms, bi, bj):
    """perms[b, i..j] reversed where bi[b] >= 0 (one gather)."""
    B, N = perms.shape
    idx = torch.arange(N, device=perms.device).unsqueeze(0)
    i, j = bi.unsqueeze(1), bj.unsqueeze(1)
    inside = (idx >= i) & (idx <= j) & (i >= 0)
    return torch.gather(perms, 1, torch.where(inside, i + j - idx, idx))


def _device_matrix(distance_matrix, device, symmetric: bool):
    d = torch.as_tensor(distance_matrix, dtype=torch.float64).to(device).contiguous()
    if d.dim() != 2 or d.shape[0] != d.shape[1]:
        raise ValueError("distance_matrix must be square")
    if symmetric and not torch.equal(d, d.t()):
        raise ValueError("the delta ranking needs a symmetric distance matrix (segment reversal); use exact=True")
    return d


def _lengths(d, perms):
    return d[perms, torch.roll(perms, -1, dims=1)].sum(dim=1)


def local_search_2_opt_batch(distance_matrix, perms, max_passes: int = -1, exact: bool = False, lengths=None):
    """Best-improvement 2-opt passes over a batch of open 0-based tours int64 [B, N] on a HIP device until no tour
    improves (max_passes < 0) or for max_passes passes.  -> (perms, float64 lengths [B]).  ``exact``: rank candidates as
    the reference does (see the module docstring); ``lengths`` then seeds the comparison (default: computed here)."""
    if perms.device.type != "cuda":
        raise TypeError("local_search_2_opt_batch needs a HIP device; there is no CPU path")
    d = _device_matrix(distance_matrix, perms.device, symmetric=not exact)
    perms = perms.to(torch.int64).contiguous().clone()
    if perms.dim() != 2 or perms.shape[1] != d.shape[0] or not bool((perms.sort(dim=1).values == torch.arange(d.shape[0], device=perms.device)).all()):
        raise ValueError("perms must be [B, N] permutations of 0..N-1")
    if exact:
        cur = (_lengths(d, perms) if lengths is None else lengths.to(torch.float64)).contiguous().clone()
    done = 0
    while max_passes < 0 or done < max_passes:
        bi, bj, bv = mops.tsp_2opt_best(d, perms, cur if exact else None)
        if not bool((bi >= 0).any()):         # the host decides when to stop, as the reference's while loop does
            break
        perms = _apply_reversals(perms, bi, bj)
        if exact:
            cur = bv
        done += 1
    return perms, (cur if exact else _lengths(d, perms))


def local_search_2_opt(distance_matrix, city_tour, recursive_seeding: int = -1, verbose: bool = True, device=None):
    """opt_2.py:27-57: (route, distance) after best-improvement 2-opt passes -- until a pass brings nothing when
    ``recursive_seeding < 0``, else exactly ``recursive_seeding`` passes.  Bit-identical to the reference."""
    device = torch.device(device) if device is not None else torch.device("cuda:0")
    if device.type != "cuda":
        raise TypeError("local_search_2_opt needs a HIP device; there is no CPU path")
    route = [int(c) for c in city_tour[0]]
    d = _device_matrix(distance_matrix, device, symmetric=False)
    n = d.shape[0]
    if len(route) != n + 1 or route[0] != route[-1] or sorted(route[:-1]) != list(range(1, n + 1)):
        raise ValueError("city_tour[0] must be a closed tour over cities 1..N (first city repeated at the end)")
    perm = torch.tensor([[c - 1 for c in route[:-1]]], dtype=torch.int64, device=device)
    distance = city_tour[1]
    cur = torch.tensor([float(distance)], dtype=torch.float64, device=device)
    iteration = 0
    while recursive_seeding < 0 or iteration < recursive_seeding:
        if verbose:
            print('Iteration = ', iteration, 'Distance = ', round(float(distance), 2))
        bi, bj, bv = mops.tsp_2opt_best(d, perm, cur)
        iteration += 1
        if int(bi[0]) < 0:                    # nothing shorter than the pass's seed
            if recursive_seeding < 0:
                break
            continue
        perm = _apply_reversals(perm, bi, bj)
        cur = bv
        distance = float(bv[0])
    if iteration and perm is not None:
        route = [int(c) + 1 for c in perm[0].tolist()]
        route.append(route[0])
    return route, distance
