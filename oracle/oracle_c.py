"""ctypes wrapper over oracle/oracle.c (TEST INFRASTRUCTURE; see the header of oracle.c).
Builds oracle/_build/liboracle.so with `make` on first use if it is missing."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_DIR, "oracle.c")):
        subprocess.run(["make", "-C", _DIR, "-B", "_build/liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def num_threads() -> int:
    return lib().orc_num_threads()


def _edges(eu, ev):
    return np.ascontiguousarray(eu, np.int32), np.ascontiguousarray(ev, np.int32)


def maxcut_obj(xs, eu, ev, bidir):
    xs = np.ascontiguousarray(xs, np.uint8)
    eu, ev = _edges(eu, ev)
    out = np.empty(xs.shape[0], np.int64)
    lib().orc_maxcut_obj(_p(xs), C.c_int64(xs.shape[0]), C.c_int64(xs.shape[1]), _p(eu), _p(ev),
                         C.c_int64(len(eu)), int(bidir), _p(out))
    return out


def ppo_step(xs_f32, action, eu, ev, bidir, last):
    """in place on xs_f32 / last; returns (reward, cur)"""
    assert xs_f32.dtype == np.float32 and xs_f32.flags.c_contiguous and last.dtype == np.float32
    eu, ev = _edges(eu, ev)
    action = np.ascontiguousarray(action, np.int64)
    B, N = xs_f32.shape
    reward, cur = np.empty(B, np.float32), np.empty(B, np.float32)
    lib().orc_ppo_step(_p(xs_f32), C.c_int64(B), C.c_int64(N), _p(action), _p(eu), _p(ev), C.c_int64(len(eu)),
                       int(bidir), _p(last), _p(reward), _p(cur))
    return reward, cur


def step_u8(xs_u8, action, eu, ev, bidir, last_i64):
    assert xs_u8.dtype == np.uint8 and xs_u8.flags.c_contiguous and last_i64.dtype == np.int64
    eu, ev = _edges(eu, ev)
    action = np.ascontiguousarray(action, np.int64)
    B, N = xs_u8.shape
    reward = np.empty(B, np.int64)
    lib().orc_step_u8(_p(xs_u8), C.c_int64(B), C.c_int64(N), _p(action), _p(eu), _p(ev), C.c_int64(len(eu)),
                      int(bidir), _p(last_i64), _p(reward))
    return reward


def greedy_sweep(xs_u8, vs_i64, eu, ev, bidir):
    assert xs_u8.dtype == np.uint8 and xs_u8.flags.c_contiguous and vs_i64.dtype == np.int64
    eu, ev = _edges(eu, ev)
    B, N = xs_u8.shape
    lib().orc_greedy_sweep(_p(xs_u8), C.c_int64(B), C.c_int64(N), _p(eu), _p(ev), C.c_int64(len(eu)), int(bidir),
                           _p(vs_i64))
    return xs_u8, vs_i64


def node_cutdeg(xs, erowptr, ev):
    xs = np.ascontiguousarray(xs, np.uint8)
    erowptr, ev = np.ascontiguousarray(erowptr, np.int32), np.ascontiguousarray(ev, np.int32)
    B, N = xs.shape
    out = np.empty((B, N), np.int64)
    lib().orc_node_cutdeg(_p(xs), C.c_int64(B), C.c_int64(N), _p(erowptr), _p(ev), _p(out))
    return out


def tsp_tour_length(dist, perm):
    dist = np.ascontiguousarray(dist, np.float32)
    perm = np.ascontiguousarray(perm, np.int64)
    out = np.empty(perm.shape[0], np.float32)
    lib().orc_tsp_tour_length(_p(dist), C.c_int64(dist.shape[0]), _p(perm), C.c_int64(perm.shape[0]), _p(out))
    return out
