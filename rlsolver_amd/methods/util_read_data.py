"""Select ops of the reference's rlsolver/methods/util_read_data.py and util.py on HIP tensors
(same names and in-place semantics), plus the host-side readers re-exported from ..graph."""
from __future__ import annotations

import torch as th

from .. import ops
from ..graph import (MyGraph, build_adjacency_bool, build_adjacency_indies,  # noqa: F401
                     calc_num_nodes_in_mygraph, load_mygraph, load_mygraph2, read_mygraph, read_tsp_file)

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
    """rlsolver/methods/util.py:87-94, in place: the ``low_k`` rows with the best values (sic -- the reference calls
    them ``low_ids``) overwrite ``low_k`` rows drawn without replacement from the others.  The ranking and the draw
    are [B]-sized torch ops (they define how torch's generator is consumed: one randperm(B - low_k)); the row
    moves are one kernel (rls_copy_rows).  With if_maximize=False the reference indexes a low_k-long tensor with
    randperm(B - low_k) and raises IndexError unless the draw happens to stay in range: reproduced."""
    B = xs.shape[0]
    rank = vs.argsort()
    if if_maximize:
        others, donors = rank[:B - low_k], rank[B - low_k:]
    else:
        others, donors = rank[:low_k], rank[low_k:]
    draw = th.randperm(B - low_k, device=xs.device)[:low_k]
    if (not if_maximize) and (draw.numel() and int(draw.max()) >= others.numel() or donors.numel() != draw.numel()):
        raise IndexError(f"index out of range: evolutionary_replacement(if_maximize=False) with B={B}, low_k={low_k} "
                         "(rlsolver/methods/util.py:91-92 fails the same way)")
    targets = others[draw].contiguous()
    donors = donors.contiguous()
    if xs.dtype in (th.bool, th.uint8) and xs.is_cuda and xs.is_contiguous():
        v64 = vs if (vs.dtype == th.int64 and vs.is_contiguous()) else None
        ops._t.copy_rows(xs, v64, targets, donors)
        if v64 is None:
            vs[targets] = vs[donors]
    else:
        raise TypeError("evolutionary_replacement needs contiguous bool / uint8 xs on a HIP device; there is no CPU path")
