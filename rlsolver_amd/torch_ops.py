"""The kernels as native PyTorch custom ops: ``torch.ops.rlsolver_hip.*``.

north_star asks for "hand-written HIP kernels through PyTorch-ROCm custom ops over a thin C-ABI".  The ops are C++
(``csrc/torch_ops.cpp`` -> ``librlsolver_torch_ops.so``, built in-tree by ``rlsolver_amd.build``): one op per DEVICE
entry point of ``include/rlsolver_hip.h``, same name without the ``rls_`` prefix, same argument order minus what a
tensor already carries (sizes, spin_bytes, the stream = torch's current HIP stream).  Outputs are caller-allocated
(``Tensor(a!)``), like the C ABI.  Only the HIP dispatch key is implemented: CPU tensors raise NotImplementedError
from the dispatcher.

The shared graph (and the spin-system env) travel as integer handles -- ``graph_handle(g)`` is the address of the
host-side ``struct rls_graph`` that the DeviceGraph keeps alive.

These ops are the ONE host path of the package: ``ops.py`` / ``ops_mcpg_tsp.py``, ``envs/`` and ``methods/`` allocate
outputs and call them; ctypes (``_abi.py``) is kept for the host-side schedule builders and queries, and for the tests
that pin the ops against raw C-ABI calls.  A dispatcher call costs about a third of a ctypes call with twelve converted
arguments.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _abi

_PKG = os.path.dirname(os.path.abspath(__file__))
OPS_PATH = os.path.join(_PKG, "librlsolver_torch_ops.so")

if not os.path.exists(OPS_PATH):
    raise ImportError(f"{OPS_PATH} is missing: the torch.ops extension has not been built "
                      "(run `python -m rlsolver_amd.build`); rlsolver_amd has no fallback for it")
_abi.lib()                                   # librlsolver_hip.so first (the ops library links against it by rpath)
torch.ops.load_library(OPS_PATH)


class _CallRecorder:
    """Test hook (RLS_RECORD_OPS=1, set by tests/conftest.py): remembers which ops the process has asked for, so that the
    GPU suite can assert that every device entry point was executed (tests/test_gpu_zz_op_coverage.py)."""

    def __init__(self, ns):
        self._ns, self.called = ns, set()

    def __getattr__(self, name):
        self.called.add(name)
        return getattr(self._ns, name)


ops = _CallRecorder(torch.ops.rlsolver_hip) if os.environ.get("RLS_RECORD_OPS") == "1" else torch.ops.rlsolver_hip

# C-ABI device entry point -> op name (tests/test_abi.py checks this list against the header)
DEVICE_ENTRY_POINTS = [
    "maxcut_obj", "maxcut_edge_cut_mask", "maxcut_node_cutdeg", "maxcut_delta_all", "maxcut_step", "maxcut_greedy_sweep",
    "maxcut_propose_accept", "maxcut_ls_weights", "maxcut_local_search", "maxcut_ls_normals", "maxcut_ls_threshold", "maxcut_ls_propose", "maxcut_ls_rounds", "select_better_rows", "pick_best_of_repeats", "copy_rows",
    "best_update", "best_key", "key_unpack", "winner_message", "winner_unpack", "rand_spins", "rand_spins_repeats", "rand_actions", "rand_perms", "spin_reset", "spin_step", "spin_observation", "spin_materialize", "spin_reset_dense", "spin_step_dense", "rand_couplings", "mcpg_metro_rounds", "mcpg_metro_stop", "mcpg_local_search",
    "mcpg_local_search_levels", "mcpg_pick_best", "mcpg_merge_best", "mcpg_value_bit_sums", "mcpg_pack_chains", "mcpg_unpack_chains",
    "qubo_local_search_value", "qubo_sparse_local_search_value", "tsp_tour_length", "tsp_swap_delta_all", "tsp_apply_swap", "tsp_2opt_delta", "tsp_2opt_best", "isco_maxcut_step",
    "isco_tsp_step",
]
# declared in the header but not device work: host-side schedule builders and queries (plain C calls, no op)
HOST_ENTRY_POINTS = [
    "version", "last_error_string", "device_count", "graph_sweep_schedule", "graph_sweep_levels", "graph_ell", "graph_sweep_batches",
    "mcpg_visit_levels", "maxcut_local_search_supported", "mcpg_local_search_levels_supported", "maxcut_ls_rounds_supported", "maxcut_ls_scratch_bytes", "maxcut_ls_slices", "maxcut_node_stats_form", "mcpg_metro_max_rounds", "mcpg_metro_scratch_bytes", "isco_maxcut_scratch_bytes", "tsp_tables8_bytes",
    "tuning_set", "tuning_unset", "tuning_get", "tuning_name",
]


def graph_handle(g) -> int:
    """Integer handle of a DeviceGraph for the ``graph`` argument of the ops (address of its host struct)."""
    return C.addressof(g.struct)


def struct_handle(s: C.Structure) -> int:
    return C.addressof(s)
