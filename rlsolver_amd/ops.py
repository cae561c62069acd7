"""Tensor-level wrappers over the kernels: allocate outputs, check device / dtype and call
``torch.ops.rlsolver_hip.*`` (csrc/torch_ops.cpp: the native custom ops over the C ABI of
include/rlsolver_hip.h, which check every shape, switch to the tensors' device and pass torch's
current HIP stream).  ctypes is used only for the host-side schedule builders and queries of the
C ABI.  PyTorch is plumbing here (memory + streams); all arithmetic happens in the HIP kernels.

No CPU path: a CPU tensor is a TypeError, a missing library an ImportError.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _abi
from .graph import GraphCSR
from .torch_ops import ops as _t

_U64_MASK = (1 << 64) - 1


def _s64(seed: int) -> int:
    """A 64-bit seed as the int64 an op schema carries (the op casts it back to uint64)."""
    seed &= _U64_MASK
    return seed - (1 << 64) if seed >= (1 << 63) else seed

TEN = torch.Tensor


_raw_stream = torch._C._cuda_getCurrentRawStream   # (device index) -> hipStream_t as int; ~0.1 us


def _stream(device) -> C.c_void_p:
    """torch's CURRENT HIP stream of ``device``, resolved at call time (so a capture stream is honoured)."""
    idx = device.index
    return C.c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))


def _check(t: TEN, name: str, dtypes, device=None, shape=None) -> TEN:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise TypeError(f"{name} must live on a HIP device (got {t.device}); rlsolver_amd has no CPU path")
    if device is not None and t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")
    if t.dtype not in dtypes:
        raise TypeError(f"{name} has dtype {t.dtype}, expected one of {dtypes}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


def _ptr(t: Optional[TEN]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


_SPIN_DTYPES = (torch.bool, torch.uint8)
SWEEP_BATCH_NODES, SWEEP_BATCH_ENTRIES = 64, 768   # what one batch may hold (rls_sweep.h: LDS ring window)


class DeviceGraph:
    """One shared graph resident on a HIP device + the rls_graph descriptor."""

    def __init__(self, csr: GraphCSR, device, use_weights: bool = False):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise TypeError(f"DeviceGraph needs a HIP device, got {self.device}")
        self.csr = csr
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)
        self.eu, self.ev = i32(csr.eu), i32(csr.ev)
        counts = np.bincount(csr.eu, minlength=csr.num_nodes) if csr.eu.size else np.zeros(csr.num_nodes, np.int64)
        erp = np.zeros(csr.num_nodes + 1, np.int64)
        np.cumsum(counts, out=erp[1:])
        self.erowptr = i32(erp)
        self.rowptr, self.col = i32(csr.rowptr), i32(csr.col)
        self.wgt = i32(csr.wgt) if use_weights else None
        # level schedule of the greedy sweep (host pass over the CSR in the C library): nodes sorted by dependency
        # level, each followed by its neighbours, as one int32 stream for the kernels' LDS ring
        rp_h = np.ascontiguousarray(csr.rowptr, dtype=np.int32)
        col_h = np.ascontiguousarray(csr.col, dtype=np.int32)
        flagged = np.empty(csr.num_nodes + 1, dtype=np.int32)
        stream = np.empty(csr.nnz + csr.num_nodes, dtype=np.int32)
        nb, nl = C.c_int64(0), C.c_int64(0)
        _abi.call("rls_graph_sweep_schedule", rp_h.ctypes.data_as(C.c_void_p), col_h.ctypes.data_as(C.c_void_p),
                  csr.num_nodes, SWEEP_BATCH_NODES, SWEEP_BATCH_ENTRIES, flagged.ctypes.data_as(C.c_void_p),
                  stream.ctypes.data_as(C.c_void_p), C.byref(nb), C.byref(nl))
        self.num_sweep_batches, self.num_sweep_levels = int(nb.value), int(nl.value)
        self.sweep_rowptr = torch.from_numpy(flagged).to(self.device)
        self.sweep_stream = torch.from_numpy(stream).to(self.device)
        # lane-per-node slabs of both adjacencies for the bit-sliced per-node kernels (K2 / K3 / local-search weights)
        erp_h = np.ascontiguousarray(erp, dtype=np.int32)
        ev_h = np.ascontiguousarray(csr.ev, dtype=np.int32)
        self.ell_sym_ptr, self.ell_sym = self._ell(rp_h, col_h, csr.num_nodes)
        self.ell_st_ptr, self.ell_st = self._ell(erp_h, ev_h, csr.num_nodes)
        # the same level schedule in lane-per-node groups for the level-parallel sweep (N < 2^20, degrees < 4096: rows of 256
        # or more entries are groups of their own)
        self.sweep_lv_ptr = self.sweep_lv_data = None
        self.num_sweep_groups = 0
        if csr.num_nodes < (1 << 20) and csr.max_degree < 4096:
            ng, tot = C.c_int64(0), C.c_int64(0)
            a = (rp_h.ctypes.data_as(C.c_void_p), col_h.ctypes.data_as(C.c_void_p), csr.num_nodes)
            _abi.call("rls_graph_sweep_levels", *a, None, 0, None, 0, C.byref(ng), C.byref(tot))
            lvp = np.empty(int(ng.value) + 1, dtype=np.int32)
            lvd = np.empty(max(int(tot.value), 1), dtype=np.int32)
            _abi.call("rls_graph_sweep_levels", *a, lvp.ctypes.data_as(C.c_void_p), lvp.size,
                      lvd.ctypes.data_as(C.c_void_p), lvd.size, C.byref(ng), C.byref(tot))
            self.num_sweep_groups = int(ng.value)
            self.sweep_lv_ptr = torch.from_numpy(lvp).to(self.device)
            self.sweep_lv_data = torch.from_numpy(lvd).to(self.device)
        self.num_nodes, self.num_stored_edges, self.nnz = csr.num_nodes, csr.num_stored_edges, csr.nnz
        self.if_bidirectional = csr.if_bidirectional
        self.struct = _abi.RlsGraph(
            num_nodes=csr.num_nodes, num_stored_edges=csr.num_stored_edges, nnz=csr.nnz,
            if_bidirectional=int(csr.if_bidirectional), max_degree=csr.max_degree,
            eu=self.eu.data_ptr(), ev=self.ev.data_ptr(), erowptr=self.erowptr.data_ptr(),
            rowptr=self.rowptr.data_ptr(), col=self.col.data_ptr(),
            wgt=0 if self.wgt is None else self.wgt.data_ptr(), sweep_rowptr=self.sweep_rowptr.data_ptr(),
            sweep_stream=self.sweep_stream.data_ptr(),
            ell_sym_ptr=self.ell_sym_ptr.data_ptr(), ell_sym=self.ell_sym.data_ptr(),
            ell_st_ptr=self.ell_st_ptr.data_ptr(), ell_st=self.ell_st.data_ptr(),
            sweep_lv_ptr=0 if self.sweep_lv_ptr is None else self.sweep_lv_ptr.data_ptr(),
            sweep_lv_data=0 if self.sweep_lv_data is None else self.sweep_lv_data.data_ptr(),
            num_sweep_groups=self.num_sweep_groups)
        self.ref = C.byref(self.struct)              # for the host-side queries of the C ABI (ctypes)
        self.handle = C.addressof(self.struct)       # the `graph` argument of torch.ops.rlsolver_hip.*


def _ell_build(self, rowptr_h, col_h, n):
    groups = (n + 63) // 64
    ptr = np.empty(groups + 1, dtype=np.int32)
    total = C.c_int64(0)
    args = (rowptr_h.ctypes.data_as(C.c_void_p), col_h.ctypes.data_as(C.c_void_p), n, ptr.ctypes.data_as(C.c_void_p))
    _abi.call("rls_graph_ell", *args, None, 0, C.byref(total))
    ell = np.empty(max(int(total.value), 1), dtype=np.int32)
    _abi.call("rls_graph_ell", *args, ell.ctypes.data_as(C.c_void_p), int(total.value), C.byref(total))
    return torch.from_numpy(ptr).to(self.device), torch.from_numpy(ell).to(self.device)


DeviceGraph._ell = _ell_build


def _spins(x: TEN, name: str, g: DeviceGraph, allow_f32=False):
    dt = _SPIN_DTYPES + ((torch.float32,) if allow_f32 else ())
    _check(x, name, dt, g.device)
    if x.dim() != 2 or x.shape[1] != g.num_nodes:
        raise ValueError(f"{name} must be [B, {g.num_nodes}], got {tuple(x.shape)}")
    return x.shape[0], (4 if x.dtype == torch.float32 else 1)


# ------------------------------------------------------------------------ MaxCut
def maxcut_obj(g: DeviceGraph, xs: TEN, out: Optional[TEN] = None) -> TEN:
    B, sb = _spins(xs, "xs", g, allow_f32=True)
    if out is None:
        out = torch.empty(B, dtype=torch.int64, device=g.device)
    _check(out, "out", (torch.int64,), g.device, (B,))
    _t.maxcut_obj(g.handle, xs, out)
    return out


def maxcut_edge_cut_mask(g: DeviceGraph, xs: TEN) -> TEN:
    B, _ = _spins(xs, "xs", g)
    out = torch.empty((B, g.num_stored_edges), dtype=torch.bool, device=g.device)
    _t.maxcut_edge_cut_mask(g.handle, xs, out)
    return out


def maxcut_node_cutdeg(g: DeviceGraph, xs: TEN) -> TEN:
    B, _ = _spins(xs, "xs", g)
    out = torch.empty((B, g.num_nodes), dtype=torch.int64, device=g.device)
    _t.maxcut_node_cutdeg(g.handle, xs, out)
    return out


def maxcut_delta_all(g: DeviceGraph, xs: TEN, out: Optional[TEN] = None) -> TEN:
    B, _ = _spins(xs, "xs", g)
    if out is None:
        out = torch.empty((B, g.num_nodes), dtype=torch.int32, device=g.device)
    _check(out, "out", (torch.int32,), g.device, (B, g.num_nodes))
    _t.maxcut_delta_all(g.handle, xs, out)
    return out


def maxcut_step(g: DeviceGraph, x_in: TEN, x_out: TEN, action: TEN, obj: TEN, reward: TEN,
                cur: Optional[TEN] = None, done: Optional[TEN] = None, done_value: float = 0.0) -> None:
    """K4.  x_out may be x_in (in-place flip) or a different buffer (emit next state)."""
    B, sb = _spins(x_in, "x_in", g, allow_f32=True)
    B2, sb2 = _spins(x_out, "x_out", g, allow_f32=True)
    if (B2, sb2) != (B, sb):
        raise ValueError("x_in and x_out must have the same shape and dtype")
    _check(action, "action", (torch.int64,), g.device, (B,))
    _check(obj, "obj", (torch.int32,), g.device, (B,))
    _check(reward, "reward", (torch.float32,), g.device, (B,))
    if cur is not None:
        _check(cur, "cur", (torch.float32,), g.device, (B,))
    if done is not None:
        _check(done, "done", (torch.float32,), g.device, (B,))
    _t.maxcut_step(g.handle, x_in, x_out, action, obj, reward, cur, done, float(done_value))


def maxcut_step_launcher(g: DeviceGraph, x_in: TEN, x_out: TEN, action: TEN, obj: TEN, reward: TEN,
                         cur: Optional[TEN] = None, done: Optional[TEN] = None, done_value: float = 0.0):
    """Validate once, launch many times: returns a zero-argument callable that enqueues K4 on torch's
    current stream (resolved at every call, so the launcher also works under hipGraph capture) with the
    given (fixed) buffers -- for rollout loops that cycle through a ring of pre-allocated slots, where
    per-call argument checking would otherwise dominate the host cost of a 50 us kernel."""
    B, sb = _spins(x_in, "x_in", g, allow_f32=True)
    B2, sb2 = _spins(x_out, "x_out", g, allow_f32=True)
    if (B2, sb2) != (B, sb):
        raise ValueError("x_in and x_out must have the same shape and dtype")
    _check(action, "action", (torch.int64,), g.device, (B,))
    _check(obj, "obj", (torch.int32,), g.device, (B,))
    _check(reward, "reward", (torch.float32,), g.device, (B,))
    if cur is not None:
        _check(cur, "cur", (torch.float32,), g.device, (B,))
    if done is not None:
        _check(done, "done", (torch.float32,), g.device, (B,))
    op = _t.maxcut_step.default      # the overload itself: skips the packet's overload resolution on every call
    # g.handle is a plain int (the address of the ctypes struct the DeviceGraph owns): the closure holds g itself, or a caller that
    # drops its graph would leave the launcher with a dangling host pointer; the tuple keeps the buffers alive
    args = (g.handle, x_in, x_out, action, obj, reward, cur, done, float(done_value))

    def launch(_op=op, _args=args, _keep=g):
        _op(*_args)
    return launch


def maxcut_greedy_sweep(g: DeviceGraph, xs: TEN, obj: TEN) -> None:
    """K5 in place.  ``obj[b]`` must equal the cut of ``xs[b]`` on entry (include/rlsolver_hip.h: the level-parallel form overwrites
    obj with the swept rows' cut, the stream forms add the accepted gains: the same thing exactly under that precondition)."""
    B, _ = _spins(xs, "xs", g)
    _check(obj, "obj", (torch.int64,), g.device, (B,))
    _t.maxcut_greedy_sweep(g.handle, xs, obj)


def maxcut_propose_accept(g: DeviceGraph, xs: TEN, mask: TEN, obj: TEN) -> None:
    """K6 in place.  ``mask``: bool / uint8 [B, N], or bit-packed int64 words [ceil(B / 64), N] (bit e of word (t, n) = env 64 t + e:
    what ``ops_mcpg_tsp.PackedChains.pack(mask.t())`` gives) -- an eighth of the bytes."""
    B, _ = _spins(xs, "xs", g)
    if mask.dtype == torch.int64:
        _check(mask, "mask", (torch.int64,), g.device, ((B + 63) // 64, g.num_nodes))
    else:
        _spins(mask, "mask", g)
        if mask.shape[0] != B:
            raise ValueError("mask must have the same shape as xs")
    _check(obj, "obj", (torch.int64,), g.device, (B,))
    _t.maxcut_propose_accept(g.handle, xs, mask, obj)


LOCAL_SEARCH_MAX_SPIN = 15


def local_search_fusable(g: DeviceGraph, num_spin: int, B: int = 1) -> bool:
    """Whether rls_maxcut_local_search covers this graph / batch / setting (else: K2 + K6 + K5 path).  The library
    answers with the test its launcher applies, so the gate and the launcher cannot disagree."""
    return bool(_abi.lib().rls_maxcut_local_search_supported(g.ref, int(B), int(num_spin)))


LS_PITCH_BYTES = 16      # row pitch of the padded local-search weights: what the kernels need.  Whole cache lines (128) were measured
                         # (tools/timing/ls_pitch.py, interleaved): local_search_inplace G22 2^16 1.292 vs 1.306 ms, N = 10^4 5.48 vs 5.43,
                         # G22 4096 envs 0.296 vs 0.301 -- nothing, so the rows stay as short as they can be


def ls_weight_dtype(g: DeviceGraph, mult: int):
    """Narrowest integer type that holds ws = deg - mult * cutdeg on this graph (|ws| <= max(1, mult - 1) * max degree): the
    fused local search streams ws once per proposal round, so its width is that kernel's HBM traffic."""
    span = g.csr.max_degree * max(1, int(mult) - 1)
    return torch.int8 if span <= 127 else (torch.int16 if span <= 32767 else torch.int32)


def maxcut_ls_weights(g: DeviceGraph, xs: TEN, mult: int, dtype=None, padded: bool = False, return_minmax: bool = False):
    """Pre-pass of the local search: (ws [B, N] int8 / int16 / int32 -- ``dtype`` or the narrowest that fits --,
    ws_std int32 [N] = max_b ws - min_b ws, folded in by the same kernel).  ``padded``: ws comes back as [B, P], P = N rounded
    up to 16 bytes of entries (the layout the round kernels read for any N; entries N .. P are padding).  ``return_minmax``:
    the second result is the int32 [2, N] table (min_b ws, max_b ws) itself -- what a sharded batch reduces over its ranks."""
    B, _ = _spins(xs, "xs", g)
    dt = ls_weight_dtype(g, mult) if dtype is None else dtype
    # rows a whole number of LS_PITCH_BYTES apart (16: what the kernels need)
    per = LS_PITCH_BYTES // torch.empty((), dtype=dt).element_size()
    P = (g.num_nodes + per - 1) // per * per if padded else g.num_nodes
    ws = torch.empty((B, P), dtype=dt, device=g.device)    # (every kernel that reads 16-byte pieces asks for the padded pitch)
    mm = torch.empty((2, g.num_nodes), dtype=torch.int32, device=g.device)
    _t.maxcut_ls_weights(g.handle, xs, int(mult), ws, mm)
    return ws, (mm if return_minmax else mm[1] - mm[0])


def maxcut_local_search(g: DeviceGraph, xs: TEN, ws: TEN, rd_std: TEN, obj: TEN, num_iters: int, num_spin: int,
                        noise: Optional[TEN] = None, seed: int = 0, env_offset: int = 0,
                        first_draw_proposes: bool = False, compute_obj: bool = False) -> None:
    """Fused local search (include/rlsolver_hip.h: rls_maxcut_local_search).  xs / obj in place.  ws [B, P >= N] with rows a
    multiple of 16 bytes apart (maxcut_ls_weights(padded=True)); the library refuses any other pitch."""
    B, _ = _spins(xs, "xs", g)
    _check(ws, "ws", (torch.int8, torch.int16), g.device)
    if ws.dim() != 2 or ws.shape[0] != B or ws.shape[1] < g.num_nodes:
        raise ValueError(f"ws must be [{B}, >= {g.num_nodes}]")
    _check(rd_std, "rd_std", (torch.float32,), g.device, (g.num_nodes,))
    _check(obj, "obj", (torch.int64,), g.device, (B,))
    if noise is not None:
        need = num_iters + (0 if first_draw_proposes else 1)
        _check(noise, "noise", (torch.float32,), g.device)
        if noise.dim() != 3 or noise.shape[0] < need or tuple(noise.shape[1:]) != (B, g.num_nodes):
            raise ValueError(f"noise must be [>= {need}, {B}, {g.num_nodes}]")
    _t.maxcut_local_search(g.handle, xs, ws, rd_std, noise, _s64(seed), env_offset, num_iters, num_spin,
                           bool(first_draw_proposes), obj, bool(compute_obj))


def maxcut_ls_normals(B: int, N: int, seed: int, draw: int, device, env_offset: int = 0, out: Optional[TEN] = None) -> TEN:
    """f32 [B, N]: the standard normals the local-search kernels draw for (seed, env_offset + b, node, draw)."""
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=torch.device(device))
    _check(out, "out", (torch.float32,), None, (B, N))
    _t.maxcut_ls_normals(out, _s64(seed), int(env_offset), int(draw))
    return out


def node_stats_form(g: DeviceGraph, B: int, symmetric: bool) -> str:
    """Which kernel family K2 / the local-search weights (symmetric=False) or K3 (True) take at this batch size: "bits" |
    "tile" | "elem" (rls_maxcut_node_stats_form)."""
    return ("elem", "bits", "tile")[int(_abi.lib().rls_maxcut_node_stats_form(g.ref, int(B), int(bool(symmetric))))]


def ls_rounds_supported(g: DeviceGraph, num_spin: int) -> bool:
    """Whether rls_maxcut_ls_threshold / rls_maxcut_ls_propose cover this graph (the library's own test)."""
    return bool(_abi.lib().rls_maxcut_ls_rounds_supported(g.ref, int(num_spin)))


def ls_scratch_bytes(g: DeviceGraph, B: int, ws_dtype, num_draws: int = 1) -> int:
    """Bytes of scratch with which a batch this small splits its noise passes (0: the batch fills the chip by itself, or the
    rows are too short to split) -- rls_maxcut_ls_scratch_bytes."""
    return int(_abi.lib().rls_maxcut_ls_scratch_bytes(g.ref, int(B), torch.empty((), dtype=ws_dtype).element_size(), int(num_draws)))


def ls_scratch(g: DeviceGraph, B: int, ws: TEN, num_draws: int = 1) -> Optional[TEN]:
    """The scratch buffer with which the round kernels split a small batch's noise passes over more workgroups -- sized for the
    mask words of ``num_draws`` rounds at once (maxcut_ls_rounds) -- or None when there is nothing to gain."""
    n = int(_abi.lib().rls_maxcut_ls_scratch_bytes(g.ref, int(B), ws.element_size(), int(num_draws)))
    return torch.empty(n, dtype=torch.uint8, device=g.device) if n > 0 else None


def maxcut_ls_threshold(g: DeviceGraph, ws: TEN, rd_std: TEN, seed: int, num_spin: int, draw: int = 0, env_offset: int = 0,
                        out: Optional[TEN] = None, scratch: Optional[TEN] = None) -> TEN:
    """thresh f32 [B] = kthvalue(ws + normal(draw) * rd_std, k = N - num_spin) with the fused local search's draws."""
    _check(ws, "ws", (torch.int8, torch.int16), g.device)
    if ws.dim() != 2 or ws.shape[1] < g.num_nodes:
        raise ValueError(f"ws must be [B, >= {g.num_nodes}]")
    _check(rd_std, "rd_std", (torch.float32,), g.device, (g.num_nodes,))
    out = torch.empty(ws.shape[0], dtype=torch.float32, device=g.device) if out is None else out
    _check(out, "out", (torch.float32,), g.device, (ws.shape[0],))
    _t.maxcut_ls_threshold(g.handle, ws, rd_std, _s64(seed), env_offset, int(draw), int(num_spin), out, scratch)
    return out


def maxcut_ls_propose(g: DeviceGraph, xs: TEN, ws: TEN, rd_std: TEN, thresh: TEN, obj: TEN, seed: int, draw: int,
                      env_offset: int = 0, scratch: Optional[TEN] = None) -> None:
    """One proposal round in place: rows of xs whose xs ^ (ws + normal(draw) * rd_std > thresh) has cut >= obj take it."""
    B, _ = _spins(xs, "xs", g)
    _check(ws, "ws", (torch.int8, torch.int16), g.device)
    if ws.dim() != 2 or ws.shape[0] != B or ws.shape[1] < g.num_nodes:
        raise ValueError(f"ws must be [{B}, >= {g.num_nodes}]")
    _check(rd_std, "rd_std", (torch.float32,), g.device, (g.num_nodes,))
    _check(thresh, "thresh", (torch.float32,), g.device, (B,))
    _check(obj, "obj", (torch.int64,), g.device, (B,))
    _t.maxcut_ls_propose(g.handle, xs, ws, rd_std, thresh, _s64(seed), env_offset, int(draw), obj, scratch)


def ls_slices(g: DeviceGraph, B: int, ws_dtype) -> int:
    """Workgroups per tile the noise passes of a batch of B envs are split over (1: the tiles fill the chip by themselves)."""
    return int(_abi.lib().rls_maxcut_ls_slices(g.ref, int(B), torch.empty((), dtype=ws_dtype).element_size()))


def maxcut_ls_rounds(g: DeviceGraph, xs: TEN, ws: TEN, rd_std: TEN, thresh: TEN, obj: TEN, seed: int, first_draw: int, num_draws: int,
                     env_offset: int = 0, scratch: Optional[TEN] = None) -> None:
    """``num_draws`` proposal rounds in place (draws first_draw, first_draw + 1, ...): maxcut_ls_propose per round, or -- scratch
    from ls_scratch(.., num_draws) -- every round's mask words first and all rounds on one load of each tile."""
    B, _ = _spins(xs, "xs", g)
    _check(ws, "ws", (torch.int8, torch.int16), g.device)
    if ws.dim() != 2 or ws.shape[0] != B or ws.shape[1] < g.num_nodes:
        raise ValueError(f"ws must be [{B}, >= {g.num_nodes}]")
    _check(rd_std, "rd_std", (torch.float32,), g.device, (g.num_nodes,))
    _check(thresh, "thresh", (torch.float32,), g.device, (B,))
    _check(obj, "obj", (torch.int64,), g.device, (B,))
    _t.maxcut_ls_rounds(g.handle, xs, ws, rd_std, thresh, _s64(seed), env_offset, int(first_draw), int(num_draws), obj, scratch)


def select_better_rows(xs0: TEN, vs0: TEN, xs1: TEN, vs1: TEN, if_maximize: bool = True) -> None:
    _check(xs0, "xs0", _SPIN_DTYPES)
    dev = xs0.device
    if xs0.dim() != 2:
        raise ValueError("xs0 must be [B, N]")
    B, N = xs0.shape
    _check(xs1, "xs1", _SPIN_DTYPES, dev, (B, N))
    _check(vs0, "vs0", (torch.int64,), dev, (B,))
    _check(vs1, "vs1", (torch.int64,), dev, (B,))
    _t.select_better_rows(xs0, vs0, xs1, vs1, bool(if_maximize))


def pick_best_of_repeats(xs: TEN, vs: TEN, num_repeats: int, if_maximize: bool = True):
    _check(xs, "xs", _SPIN_DTYPES)
    dev = xs.device
    if xs.dim() != 2 or xs.shape[0] % num_repeats != 0:
        raise ValueError("xs must be [R*S, N]")
    S, N = xs.shape[0] // num_repeats, xs.shape[1]
    _check(vs, "vs", (torch.int64,), dev, (xs.shape[0],))
    gx = torch.empty((S, N), dtype=xs.dtype, device=dev)
    gv = torch.empty(S, dtype=torch.int64, device=dev)
    _t.pick_best_of_repeats(xs, vs, num_repeats, bool(if_maximize), gx, gv)
    return gx, gv


def rand_spins(B: int, N: int, seed: int, device, env_offset: int = 0, out: Optional[TEN] = None) -> TEN:
    device = torch.device(device)
    if out is None:
        out = torch.empty((B, N), dtype=torch.bool, device=device)
    _check(out, "out", _SPIN_DTYPES, None, (B, N))
    _t.rand_spins(out, _s64(seed), env_offset)
    return out


def rand_spins_repeats(seeds, S: int, N: int, device, env_offset: int = 0, out: Optional[TEN] = None) -> TEN:
    """R repeats of S envs in one launch: row r * S + s = what ``rand_spins(S, N, seeds[r], env_offset)`` writes for env s
    (LocalSearch.reset_search's candidates).  ``seeds``: a sequence of ints (uploaded) or an int64 device tensor [R]."""
    device = torch.device(device)
    if not isinstance(seeds, torch.Tensor):
        seeds = torch.tensor([_s64(int(v)) for v in seeds], dtype=torch.int64).to(device, non_blocking=True)
    R = int(seeds.shape[0])
    if out is None:
        out = torch.empty((R * S, N), dtype=torch.bool, device=device)
    _check(out, "out", _SPIN_DTYPES, None, (R * S, N))
    if R * S:
        _t.rand_spins_repeats(out.view(R, S, N), _check(seeds, "seeds", (torch.int64,), out.device, (R,)), env_offset)
    return out


def rand_actions(B: int, N: int, seed: int, step: int, device, env_offset: int = 0,
                 out: Optional[TEN] = None) -> TEN:
    device = torch.device(device)
    if out is None:
        out = torch.empty(B, dtype=torch.int64, device=device)
    _check(out, "out", (torch.int64,), None, (B,))
    _t.rand_actions(out, N, _s64(seed), _s64(step), env_offset)
    return out
