"""rlsolver_amd -- MI355X-native parallel environment engine for combinatorial optimisation.

Drop-in for the env hot path of Open-Finance-Lab/RLSolver: the classes in
``rlsolver_amd.envs`` keep the reference's constructor keywords, method names, tensor shapes and
dtypes; the arithmetic runs in hand-written HIP kernels behind the C ABI of
``include/rlsolver_hip.h``.  There is no CPU fallback.
"""
__version__ = "0.1.0"

from . import graph  # noqa: F401  (host-only, importable without the HIP library)
