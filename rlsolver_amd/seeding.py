"""Where the kernels' seeds and env ids come from ("per-env RNG keyed by global env id, not rank").

Every in-kernel generator is keyed by (seed, GLOBAL env / chain id, draw counters).  A class of this package therefore
carries two things besides its batch:

* ``env_offset`` -- the global id of its env 0: rank r of a sharded run owns envs [env_offset, env_offset + B_local)
  (``rlsolver_amd.dist.env_shard``); 0 in a single-process run;
* a ``SeedStream`` -- one 64-bit seed per kernel call.  By default it draws from torch's CPU generator, so
  ``torch.manual_seed(s)`` makes a run reproducible exactly as it does for the reference, and ranks that seed alike
  and make the same calls get the same kernel seeds: together with ``env_offset`` their shards then hold exactly the
  envs of the one-process run, whatever the rank count.  ``seed=<int>`` gives the class a private stream instead
  (splitmix64 of seed + call counter), independent of torch's generator and of what other objects draw.
"""
from __future__ import annotations

from typing import Optional

import torch

_M64 = (1 << 64) - 1


def seed_from_torch() -> int:
    """One draw of torch's CPU generator (no device sync): th.manual_seed() makes the kernels' draws reproducible."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


class SeedStream:
    """next() -> the seed of the next kernel call; see the module docstring."""

    __slots__ = ("seed", "calls")

    def __init__(self, seed: Optional[int] = None):
        self.seed = None if seed is None else int(seed) & _M64
        self.calls = 0

    def next(self) -> int:
        if self.seed is None:
            return seed_from_torch()
        self.calls += 1
        return splitmix64(self.seed + 0xD1B54A32D192ED03 * self.calls) >> 2      # 62 bits, like seed_from_torch

    def derive(self, base: int, k: int) -> int:
        """A seed for sub-call k of a call whose seed is ``base`` (e.g. repeat k of LocalSearch.reset_search)."""
        return splitmix64((base + 0x9E3779B97F4A7C15 * (k + 1)) & _M64) >> 2

    def state_dict(self):
        return {"seed": self.seed, "calls": self.calls}

    def load_state_dict(self, d):
        self.seed, self.calls = d["seed"], int(d["calls"])


class _Keep:
    def __repr__(self):
        return "KEEP"


KEEP = _Keep()          # set_shard(): "leave this attribute as it is" (None is a value there)


class Sharded:
    """Mixin: ``env_offset`` + ``SeedStream`` (+ the process group of the few whole-batch statistics) for the env / sampler
    classes.  ``group``: a torch.distributed group (``dist.group.WORLD`` for the default one) over which statistics of the
    WHOLE batch are reduced -- the local search's per-node weight range (env_L2A.py:93-94: ``max - min over dim 0``), MCPG's
    mean value and policy-gradient sums -- so that they, too, are those of the one-process run; None = this object's batch
    is the whole batch.  ``stat_hook`` (tests): called instead of the collective, ``hook(kind, tensor) -> tensor``."""

    env_offset: int = 0
    group = None
    stat_hook = None
    _seeds: SeedStream

    def _init_shard(self, env_offset: int = 0, seed: Optional[int] = None, group=None):
        if env_offset < 0:
            raise ValueError("env_offset must be >= 0")
        self.env_offset = int(env_offset)
        self.group = group
        self._seeds = SeedStream(seed)

    def set_shard(self, env_offset: int, seed=KEEP, group=KEEP):
        """Make this object rank-aware after construction: global id of its env 0 (+ optionally a private seed stream and
        the group over which whole-batch statistics are reduced).  ``seed`` / ``group`` left out = unchanged (KEEP); None is
        a value: back to torch's generator / back to "this object's batch is the whole batch".  A seed stream that is kept
        keeps its call counter -- re-sharding an object that has already drawn must not hand out the same kernel seeds
        again; only a NEW seed starts a fresh stream."""
        if env_offset < 0:
            raise ValueError("env_offset must be >= 0")
        self.env_offset = int(env_offset)
        if group is not KEEP:
            self.group = group
        if seed is not KEEP:
            new = None if seed is None else int(seed) & _M64
            if new != self._seeds.seed:
                self._seeds = SeedStream(seed)
        return self

    def _next_seed(self) -> int:
        return self._seeds.next()

    def _global_minmax(self, mm):
        """mm int32 [2, N] (min, max over the local envs) -> over every rank's envs."""
        if self.stat_hook is not None:
            return self.stat_hook("minmax", mm)
        if self.group is not None:
            from . import dist
            return dist.all_reduce_minmax(mm, self.group)
        return mm

    def _global_sum(self, t):
        """Sum of a small tensor over the ranks (float64 / int64 on the way: the order of a float32 sum would depend on the
        rank count)."""
        if self.stat_hook is not None:
            return self.stat_hook("sum", t)
        if self.group is not None:
            from . import dist
            return dist.all_reduce_sum(t, self.group)
        return t
