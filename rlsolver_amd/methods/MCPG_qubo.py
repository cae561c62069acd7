"""QUBO samplers of the upstream MCPG package -- drop-in for
rlsolver/methods/MCPG/sampling.py:323-370 (mcpg_sampling_qubo, mcpg_sampling_qubo_bin) and the
loader rlsolver/methods/MCPG/dataloader.py:278-294, on HIP tensors."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..ops import _check, _t
from .MCPG import metro_sampling

TEN = torch.Tensor


def qubo_dataloader(filename, device):
    """Comma-separated dense matrix rows -> {'Q': f32 [n, n], 'nvar': n} (dataloader.py:278-294)."""
    rows = []
    with open(filename, "r", encoding="utf-8") as f:
        for line in f:
            vals = [v for v in line.replace(" ", "").strip().split(",") if v]
            if not vals:
                break
            rows.append([float(v) for v in vals])
    Q = torch.tensor(np.asarray(rows, dtype=np.float64)).float().to(device)
    return {"Q": Q, "nvar": Q.shape[0]}, Q.shape[0]


def qubo_local_search_value(Q: TEN, xs: TEN, num_ls: int, binary: bool):
    _check(Q, "Q", (torch.float32,))
    dev = Q.device
    n = Q.shape[0]
    if Q.shape != (n, n):
        raise ValueError("Q must be square")
    _check(xs, "xs", (torch.float32,), dev)
    if xs.dim() != 2 or xs.shape[0] != n:
        raise ValueError(f"xs must be [{n}, C]")
    Cc = xs.shape[1]
    out = torch.empty_like(xs)
    value = torch.empty(Cc, dtype=torch.float32, device=dev)
    _t.qubo_local_search_value(Q, xs, out, num_ls, bool(binary), value)
    return out, value


def qubo_to_csr(Q: TEN, levels: bool = True):
    """Dense Q [n, n] -> (rowptr int32 [n+1], col int32 [nnz], val f32 [nnz], lv_ptr, lv_rows) on Q's device, diagonal included;
    the last two are the level schedule of ``qubo_levels`` (``levels=False``: the bare CSR triple)."""
    Qc = Q.detach().to(torch.float32)
    nz = Qc != 0
    counts = nz.sum(dim=1)
    rowptr = torch.zeros(Qc.shape[0] + 1, dtype=torch.int32, device=Q.device)
    rowptr[1:] = counts.cumsum(0).to(torch.int32)
    idx = nz.nonzero()
    rowptr, col = rowptr.contiguous(), idx[:, 1].to(torch.int32).contiguous()
    if not levels:
        return rowptr, col, Qc[nz].contiguous()
    return (rowptr, col, Qc[nz].contiguous()) + qubo_levels(rowptr, col)


def qubo_levels(rowptr, col):
    """Level schedule of the Gauss-Seidel sweep over a CSR pattern: level(i) = 1 + max(level(j): j < i, Q_ij != 0).  Rows of one
    level share no entry and every neighbour above a row sits in a later level, so a level's rows can be updated side by side
    with the sequential sweep's result.  -> (lv_ptr int32 [L + 1], lv_rows int32 [n]: rows by (level, index)), on rowptr's
    device.  Host work, once per matrix."""
    rp = rowptr.detach().cpu().numpy().astype(np.int64)
    cl = col.detach().cpu().numpy().astype(np.int64)
    n = rp.size - 1
    level = np.zeros(n, dtype=np.int64)
    for i in range(n):
        nb = cl[rp[i]:rp[i + 1]]
        nb = nb[nb < i]
        if nb.size:
            level[i] = level[nb].max() + 1
    order = np.lexsort((np.arange(n), level))
    L = int(level.max()) + 1 if n else 1
    lv_ptr = np.zeros(L + 1, dtype=np.int32)
    lv_ptr[1:] = np.cumsum(np.bincount(level, minlength=L))
    dev = rowptr.device
    return torch.from_numpy(lv_ptr).to(dev), torch.from_numpy(order.astype(np.int32)).to(dev)


def qubo_sparse_local_search_value(csr, xs: TEN, num_ls: int, binary: bool):
    """The same coordinate search + value on a CSR matrix (rowptr, col, val[, lv_ptr, lv_rows]) -- O(nnz) per sweep.
    With the level schedule of ``qubo_levels`` (``qubo_to_csr`` appends it) the waves of a workgroup sweep the rows of a level side
    by side; a bare 3-tuple runs one wave per 64 chains over the rows in order.  Same result either way."""
    rowptr, col, val = csr[:3]
    lv_ptr, lv_rows = (csr[3], csr[4]) if len(csr) >= 5 else (None, None)
    dev = xs.device
    _check(rowptr, "rowptr", (torch.int32,), dev)
    _check(col, "col", (torch.int32,), dev)
    _check(val, "val", (torch.float32,), dev)
    n = rowptr.numel() - 1
    _check(xs, "xs", (torch.float32,), dev)
    if xs.dim() != 2 or xs.shape[0] != n:
        raise ValueError(f"xs must be [{n}, C]")
    out = torch.empty_like(xs)
    value = torch.empty(xs.shape[1], dtype=torch.float32, device=dev)
    _t.qubo_sparse_local_search_value(rowptr, col, val, lv_ptr, lv_rows, xs, out, num_ls, bool(binary), value)
    return out, value


def qubo_prefers_sparse(n: int, nnz: int, num_chains: int) -> bool:
    """Which K11 kernel finishes a sweep first (measured on MI355X, tools/sweeps/time_qubo_sparse.py; round 6: the CSR kernel by
    levels).

    dense (MFMA):  a sweep is n / 32 blocks of about 4 + 5.4 n / 1000 us while one workgroup per CU suffices, and
                   2 n^2 C flops at ~80 TFLOP/s beyond that;
    sparse (CSR, level schedule): W = 16 / 8 / 4 waves per 64-chain tile share a level's rows.  A row costs a wave
                   0.6 + 0.075 deg us when it has the SIMD to itself (a dependent chain) and a quarter of that in issue slots
                   once four waves share a SIMD; a level ends in a barrier (~0.8 us; ~1.65 deg + 8 levels in a random pattern).
    n = 1000, 2 % fill, 2 sweeps + value: 2^13 chains 410 us vs 883 dense, 2^15 770 vs 2402; at 10 % fill the dense kernel wins
    (1884 vs 882, 3119 vs 2405)."""
    if n <= 0 or nnz >= n * n:
        return False
    deg = nnz / float(n)
    dense_us = max(2.0 * n * n * num_chains / 80e6, (n / 32.0) * (4.0 + 5.4 * n / 1000.0))
    tiles = (num_chains + 63) // 64
    W = 16 if tiles * 16 <= 2048 else (8 if tiles * 8 <= 4096 else 4)
    row_us = 0.6 + 0.075 * deg
    sparse_us = max((n / float(W)) * row_us, tiles * n * (row_us / 4.0) / 1024.0) + (1.65 * deg + 8.0) * 0.8
    return sparse_us < dense_us


def _sample(data, start_result, probs, num_ls, change_times, total_mcmc_num, device, binary, index, u):
    Q = data['Q'].to(device=device, dtype=torch.float32).contiguous()
    raw_samples = metro_sampling(probs, start_result, change_times, device, index=index, u=u)   # never modifies its input
    if 'csr' not in data:
        nnz = int((Q != 0).sum())
        data['csr'] = qubo_to_csr(Q) if qubo_prefers_sparse(Q.shape[0], nnz, raw_samples.shape[1]) else None
    if data.get('csr') is not None:
        samples, res_sample = qubo_sparse_local_search_value(data['csr'], raw_samples.contiguous(), num_ls, binary)
    else:
        samples, res_sample = qubo_local_search_value(Q, raw_samples.contiguous(), num_ls, binary)
    res_reshape = res_sample.reshape(-1, total_mcmc_num)
    idx = torch.argmax(res_reshape, dim=0)
    idx = torch.arange(total_mcmc_num, device=res_sample.device) + idx * total_mcmc_num
    max_res = res_sample[idx]
    return max_res, samples[:, idx], raw_samples, -(res_sample - torch.mean(res_sample.float()))


def mcpg_sampling_qubo(data, start_result, probs, num_ls, change_times, total_mcmc_num, device=None,
                       index: Optional[TEN] = None, u: Optional[TEN] = None):
    device = start_result.device if device is None else torch.device(device)
    return _sample(data, start_result, probs, num_ls, change_times, total_mcmc_num, device, False, index, u)


def mcpg_sampling_qubo_bin(data, start_result, probs, num_ls, change_times, total_mcmc_num, device=None,
                           index: Optional[TEN] = None, u: Optional[TEN] = None):
    device = start_result.device if device is None else torch.device(device)
    return _sample(data, start_result, probs, num_ls, change_times, total_mcmc_num, device, True, index, u)
