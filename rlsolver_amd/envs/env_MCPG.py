"""rlsolver/envs/env_MCPG.py:24-116 is a byte-identical copy of env_L2A.EnvMaxcut; same here."""
from .env_L2A import EnvMaxcut  # noqa: F401
from ..methods.util_read_data import update_xs_by_vs, pick_xs_by_vs  # noqa: F401
