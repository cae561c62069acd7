"""Select ops of the reference's rlsolver/methods/util_read_data.py and util.py on HIP tensors
(same names and in-place semantics), plus the host-side readers re-exported from ..graph."""
from __future__ import annotations

import torch as th

from .. import ops
from ..graph import (MyGraph, build_adjacency_bool, build_adjacency_indies,  # noqa: F401
                     calc_num_nodes_in_mygraph, load_mygraph2, read_mygraph, read_tsp_file)

TEN = th.Tensor


def update_xs_by_vs(xs0: TEN, vs0: TEN, xs1: TEN, vs1: TEN, if_maximize: bool = True) -> int:
    """rlsolver/methods/util_read_data.py:190-202: rows of (xs1, vs1) that are not worse replace
    (xs0, vs0) in place.  Returns xs0.shape[0] like the reference (``good_is.shape[0]``, sic)."""
    if vs0.dtype != th.int64 or vs1.dtype != th.int64:
        if vs0.dtype != vs1.dtype:  # same failure mode as torch index_put in the reference
            raise RuntimeError(f"Index put requires the source and destination dtypes match, "
                               f"got {vs0.dtype} for the destination and {vs1.dtype} for the source.")
        vs0i, vs1i = vs0.to(th.int64), vs1.to(th.int64)
        ops.select_better_rows(xs0, vs0i, xs1, vs1i, if_maximize)
        vs0.copy_(vs0i)
        return xs0.shape[0]
    ops.select_better_rows(xs0, vs0, xs1, vs1, if_maximize)
    return xs0.shape[0]


def pick_xs_by_vs(xs: TEN, vs: TEN, num_repeats: int, if_maximize: bool = True):
    """rlsolver/methods/util_read_data.py:204-216."""
    return ops.pick_best_of_repeats(xs, vs.to(th.int64), num_repeats, if_maximize)


def evolutionary_replacement(xs: TEN, vs: TEN, low_k: int, if_maximize: bool = True):
    """rlsolver/methods/util.py:87-94 (the reference's index arithmetic, including its naming:
    the ``low_k`` BEST rows overwrite ``low_k`` random others when maximising).  Row moves are
    [low_k, N] gathers -- control-plane sized, left to torch indexing."""
    num_sims = xs.shape[0]
    ids = vs.argsort()
    top_ids, low_ids = (ids[:-low_k], ids[-low_k:]) if if_maximize else (ids[:low_k], ids[low_k:])
    replace_ids = top_ids[th.randperm(num_sims - low_k, device=xs.device)[:low_k]]
    xs[replace_ids] = xs[low_ids]
    vs[replace_ids] = vs[low_ids]
