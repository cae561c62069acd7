"""ctypes binding of the C ABI in include/rlsolver_hip.h.

This is the only place the shared library is opened.  There is NO fallback: if
``librlsolver_hip.so`` is missing or a call returns an error code, an exception
is raised -- the product path never computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "librlsolver_hip.so")

RLS_OK = 0
ERROR_NAMES = {-1: "RLS_EINVAL", -2: "RLS_EUNSUPPORTED", -3: "RLS_ELAUNCH", -4: "RLS_ENODEVICE"}


class RlsError(RuntimeError):
    def __init__(self, fn: str, code: int, msg: str):
        super().__init__(f"{fn} failed: {ERROR_NAMES.get(code, code)}: {msg}")
        self.code = code


class RlsGraph(C.Structure):
    """struct rls_graph (host struct of device pointers)."""
    _fields_ = [
        ("num_nodes", C.c_int64),
        ("num_stored_edges", C.c_int64),
        ("nnz", C.c_int64),
        ("if_bidirectional", C.c_int32),
        ("max_degree", C.c_int32),
        ("eu", C.c_void_p),
        ("ev", C.c_void_p),
        ("erowptr", C.c_void_p),
        ("rowptr", C.c_void_p),
        ("col", C.c_void_p),
        ("wgt", C.c_void_p),
        ("sweep_rowptr", C.c_void_p),
        ("sweep_stream", C.c_void_p),
        ("ell_sym_ptr", C.c_void_p),
        ("ell_sym", C.c_void_p),
        ("ell_st_ptr", C.c_void_p),
        ("ell_st", C.c_void_p),
        ("sweep_lv_ptr", C.c_void_p),
        ("sweep_lv_data", C.c_void_p),
        ("num_sweep_groups", C.c_int64),
    ]


_P = C.c_void_p
_I64 = C.c_int64
_U64 = C.c_uint64
_INT = C.c_int
_F32 = C.c_float
_F64 = C.c_double
_G = C.POINTER(RlsGraph)


class RlsSpinEnv(C.Structure):
    """struct rls_spin_env (host struct of device pointers)."""
    _fields_ = ([(n, C.c_void_p) for n in ("state", "delta", "score", "best_score", "best_spins", "num_nonpos", "dist_best",
                                           "packed", "hash", "hist", "hist_hash")] + [("hist_cap", C.c_int64)]
                + [(n, C.c_void_p) for n in ("last_flip", "scalars", "time_table")] + [("table_len", C.c_int64)]
                + [(n, C.c_void_p) for n in ("best_obs_score", "mem_spins", "mem_score")] + [("mem_len", C.c_int64), ("allow_pass", C.c_int64)])


_SE = C.POINTER(RlsSpinEnv)

# name -> argtypes; every function returns int.  Keep in sync with include/rlsolver_hip.h
# (tests/test_abi.py parses the header and checks this table against it).
SIGNATURES = {
    "rls_graph_sweep_batches": [_P, _P, _I64, C.c_int32, C.c_int32, _P, _P],
    "rls_graph_sweep_levels": [_P, _P, _I64, _P, _I64, _P, _I64, _P, _P],
    "rls_mcpg_visit_levels": [_P, _P, _I64, _P, _P, _I64, _P, _I64, _P, _P],
    "rls_mcpg_local_search_levels": [_G, _P, _INT, _I64, _P, _INT, _I64, _P, _P, _I64, _I64, _P, _U64, _P, _P, _P],
    "rls_graph_ell": [_P, _P, _I64, _P, _P, _I64, _P],
    "rls_graph_sweep_schedule": [_P, _P, _I64, C.c_int32, C.c_int32, _P, _P, _P, _P],
    "rls_maxcut_obj": [_G, _P, _INT, _I64, _P, _P],
    "rls_maxcut_edge_cut_mask": [_G, _P, _I64, _P, _P],
    "rls_maxcut_node_cutdeg": [_G, _P, _I64, _P, _P],
    "rls_maxcut_delta_all": [_G, _P, _I64, _P, _P],
    "rls_maxcut_step": [_G, _P, _P, _INT, _I64, _P, _P, _P, _P, _P, _F32, _P],
    "rls_maxcut_greedy_sweep": [_G, _P, _I64, _P, _P],
    "rls_maxcut_propose_accept": [_G, _P, _I64, _P, C.c_int32, _P, _P],
    "rls_maxcut_ls_weights": [_G, _P, _I64, C.c_int32, _P, C.c_int32, _I64, _P, _P],
    "rls_maxcut_local_search": [_G, _P, _I64, _P, C.c_int32, _I64, _P, _P, _U64, _I64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P],
    "rls_maxcut_ls_normals": [_P, _I64, _I64, _U64, _I64, C.c_int32, _P],
    "rls_maxcut_ls_threshold": [_G, _I64, _P, C.c_int32, _I64, _P, _U64, _I64, C.c_int32, C.c_int32, _P, _P, _I64, _P],
    "rls_maxcut_ls_propose": [_G, _P, _I64, _P, C.c_int32, _I64, _P, _P, _U64, _I64, C.c_int32, _P, _P, _I64, _P],
    "rls_maxcut_ls_rounds": [_G, _P, _I64, _P, C.c_int32, _I64, _P, _P, _U64, _I64, C.c_int32, C.c_int32, _P, _P, _I64, _P],
    "rls_select_better_rows": [_P, _P, _P, _P, _I64, _I64, _INT, _P],
    "rls_pick_best_of_repeats": [_P, _P, _I64, _I64, _I64, _INT, _P, _P, _P],
    "rls_rand_spins": [_P, _I64, _I64, _U64, _I64, _P],
    "rls_rand_spins_repeats": [_P, _I64, _I64, _I64, _P, _I64, _P],
    "rls_rand_actions": [_P, _I64, _I64, _U64, _U64, _I64, _P],
    "rls_spin_observation": [_SE, _P, C.c_int32, _INT, _I64, C.c_int32, _I64, _P, _I64, C.c_int32, _P, _P],
    "rls_spin_materialize": [_SE, _INT, _I64, _I64, C.c_int32, _P, _I64, _P],
    "rls_rand_couplings": [_P, _INT, _I64, _I64, C.c_int32, _F64, C.c_int32, C.c_int32, _U64, _I64, _P],
    "rls_spin_reset_dense": [_P, _SE, _INT, _I64, _I64, C.c_int32, _P, _P, _P, _P, _P],
    "rls_spin_step_dense": [_P, _P, _SE, _INT, _I64, _I64, C.c_int32, _P, _P, _P, _P, _F64, C.c_int32, _F64, _I64, C.c_int32, _F64,
                            C.c_int32, _F64, _P],
    "rls_spin_reset": [_G, _SE, _INT, _I64, C.c_int32, _P, _F64, _I64, _P],
    "rls_spin_step": [_G, _SE, _INT, _I64, C.c_int32, _P, _P, _P, _P, _F64, _F64, C.c_int32, _F64, _I64, C.c_int32, _F64,
                      C.c_int32, _F64, _P],
    "rls_mcpg_metro_rounds": [_P, _P, _I64, _INT, _I64, _I64, _P, _I64, _I64, _P, _P, _U64, _P, _INT, _P, _I64, _P, _P, _I64, _P],
    "rls_mcpg_metro_stop": [_P, _I64, _I64, _I64, C.c_int32, _I64, _P, _P, _P],
    "rls_mcpg_local_search": [_G, _P, _INT, _P, _I64, _P, _P, _I64, _I64, _P, _U64, _P, _I64, _P, _P, _P],
    "rls_mcpg_pick_best": [_P, _P, _INT, _I64, _I64, _I64, _I64, _P, _P, _P, _P],
    "rls_mcpg_merge_best": [_P, _P, _P, _P, _I64, _I64, _P, _P, _P, C.c_int32, _P],
    "rls_mcpg_value_bit_sums": [_P, _I64, _I64, _P, _P, _P],
    "rls_mcpg_pack_chains": [_P, _INT, _I64, _I64, _P, _P],
    "rls_mcpg_unpack_chains": [_P, _I64, _I64, _P, _P],
    "rls_qubo_local_search_value": [_P, _I64, _P, _P, _I64, _I64, _INT, _P, _P],
    "rls_qubo_sparse_local_search_value": [_P, _P, _P, _I64, _P, _P, C.c_int32, _P, _P, _I64, _I64, _INT, _P, _P],
    "rls_tsp_tour_length": [_P, _I64, _P, _I64, _P, _P],
    "rls_tsp_swap_delta_all": [_P, _I64, _P, _I64, _P, _P, C.c_int32, _P, C.c_int32, _P, _F32, C.c_uint64, _I64, _P, _F32, _P, _P, _P, _P],
    "rls_tsp_apply_swap": [_P, _I64, _I64, _P, _P, _P],
    "rls_tsp_2opt_delta": [_P, _I64, _P, _I64, _P, _P, _P, _P],
    "rls_tsp_2opt_best": [_P, _I64, _P, _I64, _P, C.c_int32, _P, _P, _P, _P],
    "rls_rand_perms": [_P, _I64, _I64, _U64, _I64, _P],
    "rls_isco_maxcut_step": [_G, _P, _P, _I64, _P, _F32, _P, _P, _U64, _I64, _P, _P, _P, _P, _P, _I64, _P],
    "rls_isco_tsp_step": [_P, _I64, _P, C.c_int32, _F32, _P, C.c_int32, _P, _P, _I64, C.c_int32, _F32, _P, _P, _P, _P, _P, _U64, _I64,
                          _P, _P, _P, _P],
    "rls_copy_rows": [_P, _P, _I64, _P, _P, _I64, _P],
    "rls_best_update": [_P, _P, _INT, _I64, _I64, _INT, _P, _P, _P, _P, _I64, _INT, _P],
    "rls_best_key": [_P, _INT, _I64, C.c_int32, _I64, _I64, _P, _P, _P, _P],
    "rls_key_unpack": [_P, C.c_int32, _I64, _INT, _P, _P, _I64, _P, _P],
    "rls_winner_message": [_P, _I64, _I64, _P, _P, C.c_int32, _I64, _I64, _P, _P],
    "rls_winner_unpack": [_P, _I64, _P, _P, _P],
    "rls_tuning_set": [_P, _I64],
    "rls_tuning_unset": [_P],
    "rls_tuning_get": [_P, _P, _P],
    "rls_tuning_name": [C.c_int32, _P],
}
# functions that return a value, not an error code
PLAIN = {"rls_version": ([], _INT), "rls_device_count": ([], _INT), "rls_last_error_string": ([], C.c_char_p),
         "rls_maxcut_local_search_supported": ([_G, _I64, C.c_int32], _INT),
         "rls_mcpg_local_search_levels_supported": ([_G, _I64], _INT),
         "rls_maxcut_ls_rounds_supported": ([_G, C.c_int32], _INT),
         "rls_maxcut_node_stats_form": ([_G, _I64, C.c_int32], _INT),
         "rls_mcpg_metro_max_rounds": ([_I64, C.c_int32], _I64),
         "rls_mcpg_metro_scratch_bytes": ([_I64, _I64], _I64),
         "rls_tsp_tables8_bytes": ([_I64, C.c_int32], _I64),
         "rls_isco_maxcut_scratch_bytes": ([_G, _I64], _I64),
         "rls_maxcut_ls_scratch_bytes": ([_G, _I64, C.c_int32, C.c_int32], _I64),
         "rls_maxcut_ls_slices": ([_G, _I64, C.c_int32], _INT)}

_lib = None
_lock = threading.Lock()


def lib() -> C.CDLL:
    """Open the library once.  Raises if it has not been built (python -m rlsolver_amd.build)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                # torch first: it ships its own libamdhip64; whichever copy of that soname is loaded first serves the whole process,
                # and with the system copy in place torch.cuda.is_available() turns False (seen with build() -> smoke() in one
                # interpreter that had not imported torch yet)
                import torch  # noqa: F401
                if not os.path.exists(LIB_PATH):
                    raise ImportError(
                        f"{LIB_PATH} is missing: the HIP extension has not been built "
                        "(run `python -m rlsolver_amd.build`); rlsolver_amd has no CPU fallback")
                l = C.CDLL(LIB_PATH)
                for name, (args, res) in PLAIN.items():
                    f = getattr(l, name)
                    f.argtypes, f.restype = args, res
                for name, args in SIGNATURES.items():
                    f = getattr(l, name)
                    f.argtypes, f.restype = args, _INT
                _lib = l
    return _lib


def call(name: str, *args) -> None:
    rc = getattr(lib(), name)(*args)
    if rc != RLS_OK:
        msg = lib().rls_last_error_string()
        raise RlsError(name, rc, msg.decode() if msg else "")


def version() -> int:
    return lib().rls_version()


def device_count() -> int:
    return lib().rls_device_count()


# ---- tuning table (ABI v11): explicit, in-process; the production library reads no environment variable ----
def tuning_set(name: str, value: int) -> None:
    call("rls_tuning_set", name.encode(), int(value))


def tuning_unset(name=None) -> None:
    call("rls_tuning_unset", None if name is None else name.encode())


def tuning_get(name: str):
    """The forced value of a knob, or None when the launch policy decides."""
    v, st = C.c_int64(0), C.c_int32(0)
    call("rls_tuning_get", name.encode(), C.byref(v), C.byref(st))
    return int(v.value) if st.value else None


def tuning_names():
    out, i = [], 0
    while True:
        p = C.c_char_p()
        if lib().rls_tuning_name(i, C.byref(p)) != RLS_OK:
            return out
        out.append(p.value.decode())
        i += 1


def tuning_from_env(environ=None) -> dict:
    """TEST / TOOL helper: copy RLS_<KNOB> variables of `environ` into the table (what a -DRLS_DEV build does at load)."""
    environ = os.environ if environ is None else environ
    done = {}
    for n in tuning_names():
        if n in environ:
            tuning_set(n, int(environ[n]))
            done[n] = int(environ[n])
    return done
