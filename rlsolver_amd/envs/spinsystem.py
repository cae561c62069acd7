"""SpinSystem -- the S2V / ECO / PECO environment surface (SURVEY.md section 8b "S2V/ECO obs
contract", 8f item 1) on a HIP device, for ONE shared signed-weight graph (instance-wise inference)
or, with ``graph_generator=``, for per-env couplings matrix [B, N, N] redrawn at every reset (training).

Mirrors the batched env of rlsolver/methods/ECO_S2V/src/envs/spinsystem_PECO.py (class
SpinSystemUnbiased) and its instance-wise inference twin inference_network_env.py:

    reset(spins=None) -> obs f32 [B, R + N, N]     rows 0..R-1 observables (row 0 = spins, {0,1}
                                                   under SpinBasis.BINARY), rows R.. = adjacency
    step(action int64 [B]) -> (obs, reward f32 [B], done bool [B])
    attrs: num_envs, n_spins, max_steps, current_step, score, best_score, best_spins,
           action_space.n, observation_space.shape, get_best_cut(), get_allowed_action_states(),
           get_observation(), matrix (dense [N, N] f32, built lazily)

What runs where: the flip, the score change, the all-node gain cache ("immediate cuts available",
kept as int32 [B, N] and updated in O(deg) instead of the reference's dense [B,N,N] matmul), reward,
best tracking, every observable row AND the visited-state memory behind stag_punishment /
basin_reward (HistoryBuffer, util_envs_PECO.py:228-288: bit-packed states of the episode in a
pre-allocated [B, max_steps, N/64] ring, exact compare behind a Zobrist-hash pre-filter) are ONE HIP
kernel per step (rls_spin_step); reset is the K3 gain kernel + rls_spin_reset.  Edge weights must
be integers (EdgeType.DISCRETE / unweighted).

Training envs: ``SpinSystem(None, None, num_envs, graph_generator=gg, ...)`` keeps what the reference's training env keeps --
``gg.get()`` (a float tensor [B, N, N] on the device; ``gg.n_spins``) is called at every reset, graphs the reference would
reject (no edges / zero max local reward) are drawn again -- and steps with rls_spin_step_dense: the flipped node's row of
its env's matrix is the neighbour list, max_local_reward_available is per env.  Generators: envs/util_envs_PECO.py.

``dtype=torch.float64`` gives the arithmetic of the reference's numpy env
(ECO_S2V/src/envs/spinsystem.py); ``SpinSystemUnbiased`` below is that env's single-instance surface.
"""
from __future__ import annotations

import ctypes as C
from enum import Enum
from typing import Optional, Sequence

import numpy as np
import torch

from .. import _abi, ops
from ..graph import build_csr
from ..ops import _t
from ..seeding import Sharded


class Observable(Enum):  # ECO_S2V/src/envs/util_envs.py:40-51
    SPIN_STATE = 1
    IMMEDIATE_REWARD_AVAILABLE = 2
    TIME_SINCE_FLIP = 3
    EPISODE_TIME = 4
    TERMINATION_IMMANENCY = 5
    NUMBER_OF_GREEDY_ACTIONS_AVAILABLE = 6
    DISTANCE_FROM_BEST_SCORE = 7
    DISTANCE_FROM_BEST_STATE = 8


ECO_PECO_OBSERVABLES = [Observable.SPIN_STATE, Observable.IMMEDIATE_REWARD_AVAILABLE, Observable.TIME_SINCE_FLIP,
                        Observable.DISTANCE_FROM_BEST_SCORE, Observable.DISTANCE_FROM_BEST_STATE,
                        Observable.NUMBER_OF_GREEDY_ACTIONS_AVAILABLE, Observable.TERMINATION_IMMANENCY]
S2V_OBSERVABLES = [Observable.SPIN_STATE]


class RewardSignal(Enum):
    DENSE = 1
    BLS = 2
    SINGLE = 3
    CUSTOM_BLS = 4


class SpinBasis(Enum):
    SIGNED = 1
    BINARY = 2


class ExtraAction(Enum):  # util_envs.py:24-27
    PASS = 1
    RANDOMISE = 2
    NONE = 3


class OptimisationTarget(Enum):  # util_envs.py:30-32
    CUT = 1
    ENERGY = 2


_ROW_ORDER = [Observable.IMMEDIATE_REWARD_AVAILABLE, Observable.TIME_SINCE_FLIP, Observable.EPISODE_TIME,
              Observable.TERMINATION_IMMANENCY, Observable.NUMBER_OF_GREEDY_ACTIONS_AVAILABLE,
              Observable.DISTANCE_FROM_BEST_SCORE, Observable.DISTANCE_FROM_BEST_STATE]
_REWARD_MODE = {RewardSignal.DENSE: 0, RewardSignal.BLS: 1, RewardSignal.CUSTOM_BLS: 2}


class SpinSystem(Sharded):
    class _ActionSpace:
        def __init__(self, n, device):
            self.n, self.device = n, device
            self.actions = torch.arange(n, device=device)

        def sample(self, n=1):
            return self.actions[torch.randint(0, self.n, (n,), device=self.device)].tolist()

    class _ObservationSpace:
        def __init__(self, n_spins, n_observables):
            self.shape = [n_spins, n_observables]

    def __init__(self, mygraph, num_nodes: Optional[int], num_envs: int, max_steps: int = 20,
                 observables: Sequence[Observable] = ECO_PECO_OBSERVABLES,
                 reward_signal: RewardSignal = RewardSignal.DENSE, spin_basis: SpinBasis = SpinBasis.SIGNED,
                 norm_rewards: bool = False, horizon_length: Optional[int] = None,
                 stag_punishment: Optional[float] = None, basin_reward: Optional[float] = None,
                 device=None, include_adjacency: bool = True, dtype=torch.float32, graph_generator=None,
                 extra_action: ExtraAction = ExtraAction.NONE, memory_length: Optional[int] = None,
                 env_offset: int = 0, seed: Optional[int] = None, group=None, _numpy_env_rules: bool = False,
                 _score_scale: float = 1.0):
        # _numpy_env_rules / _score_scale: set by SpinSystemUnbiased below, the single-instance numpy surface -- the maximum local
        # reward is taken over the NONZERO entries (spinsystem.py:190-196; the batched env takes it over all and draws again at
        # 0); rewards count a score kept in units of 1 / _score_scale (OptimisationTarget.ENERGY = 2 x CUT on the negated graph)
        self._numpy_env_rules, self._score_scale = bool(_numpy_env_rules), float(_score_scale)
        # env_offset / seed / group: rlsolver_amd/seeding.py -- num_envs is this rank's share of a sharded batch, reset draws
        # are keyed by the global env id (a graph_generator carries its own env_offset: give it the same one)
        self._init_shard(env_offset, seed, group)
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise TypeError(f"rlsolver_amd.SpinSystem needs a HIP device (got {self.device}); there is no CPU path")
        if dtype not in (torch.float32, torch.float64):
            raise TypeError("dtype must be torch.float32 or torch.float64")
        self.dtype = dtype
        if observables[0] != Observable.SPIN_STATE:
            raise AssertionError("First observable must be Observation.SPIN_STATE.")
        if reward_signal not in _REWARD_MODE:
            raise NotImplementedError(f"reward_signal {reward_signal} is not supported on the batched env")
        self.observables = list(enumerate(observables))
        if graph_generator is not None:
            if mygraph is not None:
                raise ValueError("pass either mygraph (one shared graph) or graph_generator (per-env matrices)")
            num_nodes = int(graph_generator.n_spins) if num_nodes is None else num_nodes
        self.num_envs, self.n_spins, self.max_steps = num_envs, num_nodes, max_steps
        self._seed = None
        # ExtraAction.PASS and a finite memory_length are options of the single-instance numpy env (spinsystem.py:349-351,
        # 398-404; SpinSystemUnbiased below passes them).  The reference's BATCHED env cannot even be constructed with them
        # (spinsystem_PECO.py raises in its constructor), so SpinSystemFactory.get keeps refusing them.
        if extra_action not in (ExtraAction.NONE, ExtraAction.PASS):
            raise NotImplementedError("ExtraAction.RANDOMISE: the reference itself raises on its first use (spinsystem.py:352-358 "
                                      "multiplies an [n_spins + 1] row with an [n_spins] draw)")
        if memory_length is not None and memory_length < 2:
            raise ValueError("memory_length must be >= 2 or None (the reference starts writing its ring at index 1, spinsystem.py:209: "
                             "a ring of one slot raises IndexError on its first step)")
        self._pass = extra_action == ExtraAction.PASS
        self.n_actions = num_nodes + int(self._pass)
        self.reward_signal, self.norm_rewards, self.spin_basis = reward_signal, norm_rewards, spin_basis
        self.horizon_length = horizon_length if horizon_length is not None else max_steps
        self.stag_punishment, self.basin_reward = stag_punishment, basin_reward
        self.reversible_spins = True
        self.extra_action, self.optimisation_target = extra_action, OptimisationTarget.CUT
        self.memory_length = memory_length
        self.include_adjacency = include_adjacency
        self.action_space = self._ActionSpace(self.n_actions, self.device)
        self.observation_space = self._ObservationSpace(self.n_spins, len(observables))
        self.gg = graph_generator
        self._dense = graph_generator is not None
        if self._dense:
            if getattr(graph_generator, "biased", False):
                raise NotImplementedError("biased graph generators are not supported")
            self.graph = None
            self._matrix = None
            self._max_local = None
            self.max_local_reward_available_ = torch.ones(num_envs, device=self.device, dtype=dtype)
            self._weight_sum_env = torch.zeros(num_envs, device=self.device, dtype=dtype)
            self._flags = torch.zeros(num_envs, device=self.device, dtype=torch.uint8)
        else:
            csr = build_csr(mygraph, num_nodes=num_nodes, if_bidirectional=False)
            if np.any(csr.wgt != csr.wgt.astype(np.int32)):
                raise ValueError("SpinSystem needs integer edge weights")
            self.graph = ops.DeviceGraph(csr, self.device, use_weights=True)
            wdeg = np.zeros(num_nodes, np.int64)
            np.add.at(wdeg, np.repeat(np.arange(num_nodes), np.diff(csr.rowptr)), csr.wgt)
            wmax = wdeg[wdeg != 0].max() if self._numpy_env_rules and np.any(wdeg != 0) else wdeg.max()
            self._max_local = float(wmax)                # host copy: reading it back from the device synced every step
            self.max_local_reward_available_ = torch.full((num_envs,), self._max_local, device=self.device, dtype=dtype)
            if float(wmax) == 0.0 or np.abs(wdeg).sum() == 0:
                raise ValueError("empty graph / zero max local reward (the reference re-draws the graph here)")
            self._matrix = None
            self._weight_sum = int(csr.wgt.sum())         # sum of W over ordered pairs
        self.max_local_reward_available = self.max_local_reward_available_.unsqueeze(1).expand(-1, num_nodes)
        # host int32[7]: where each observable of _ROW_ORDER sits in `state` (-1 = not observed)
        self._rows = torch.tensor([next((i for i, o in self.observables if o == want), -1) for want in _ROW_ORDER], dtype=torch.int32)
        R, B, N = len(observables), num_envs, num_nodes
        dt = self.dtype
        # `_state`: rows 0 (signed spins) and IMMEDIATE_REWARD_AVAILABLE are current after every step; the rows a step changes
        # everywhere live as last-flip steps + four per-env scalars and are written into it on demand (the `state` property)
        self._state = torch.zeros((B, R, N), dtype=dt, device=self.device)
        self._state_stale = False
        self._last_flip = torch.zeros((B, N), dtype=torch.int32, device=self.device)
        self._scalars = torch.zeros((B, 4), dtype=dt, device=self.device)
        self._time_table = self._build_time_table()
        self._delta = torch.zeros((B, N), dtype=torch.int32, device=self.device)
        self._num_nonpos = torch.zeros(B, dtype=torch.int32, device=self.device)
        self._dist_best = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.score = torch.zeros(B, dtype=dt, device=self.device)
        self.best_score = torch.zeros(B, dtype=dt, device=self.device)
        self.best_spins = torch.zeros((B, N), dtype=dt, device=self.device)
        self._use_hist = stag_punishment is not None or basin_reward is not None
        self._use_packed = self._use_hist or memory_length is not None
        W = (N + int(self._pass) + 63) // 64
        self.best_obs_score = torch.zeros(B, dtype=dt, device=self.device)
        if self._use_packed:   # torch.int64 carries the uint64 bit patterns
            self._packed = torch.zeros((B, W), dtype=torch.int64, device=self.device)
            self._hash = torch.zeros(B, dtype=torch.int64, device=self.device)
        if self._use_hist:     # one slot per step of an episode
            self._hist = torch.zeros((B, max_steps, W), dtype=torch.int64, device=self.device)
            self._hist_hash = torch.zeros((B, max_steps), dtype=torch.int64, device=self.device)
        if memory_length is not None:
            self._mem_spins = torch.zeros((B, memory_length, (N + 63) // 64), dtype=torch.int64, device=self.device)
            self._mem_score = torch.zeros((B, memory_length), dtype=dt, device=self.device)
        self._visited_new = torch.ones(B, dtype=torch.uint8, device=self.device)
        self._env = _abi.RlsSpinEnv(
            state=self._state.data_ptr(), delta=self._delta.data_ptr(), score=self.score.data_ptr(),
            best_score=self.best_score.data_ptr(), best_spins=self.best_spins.data_ptr(),
            num_nonpos=self._num_nonpos.data_ptr(), dist_best=self._dist_best.data_ptr(),
            packed=self._packed.data_ptr() if self._use_packed else 0, hash=self._hash.data_ptr() if self._use_packed else 0,
            hist=self._hist.data_ptr() if self._use_hist else 0,
            hist_hash=self._hist_hash.data_ptr() if self._use_hist else 0, hist_cap=max_steps if self._use_hist else 0,
            last_flip=self._last_flip.data_ptr(), scalars=self._scalars.data_ptr(), time_table=self._time_table.data_ptr(),
            table_len=self._time_table.numel(), best_obs_score=self.best_obs_score.data_ptr(),
            mem_spins=self._mem_spins.data_ptr() if memory_length is not None else 0,
            mem_score=self._mem_score.data_ptr() if memory_length is not None else 0,
            mem_len=memory_length or 0, allow_pass=int(self._pass))
        self._env_handle = C.addressof(self._env)     # the `env` argument of torch.ops.rlsolver_hip.spin_*
        self.current_step = 0
        self.reset()

    def _build_time_table(self):
        """time_table[k] = what `state[TIME_SINCE_FLIP] += 1. / max_steps` (spinsystem_PECO.py:417, spinsystem.py:436) holds
        after k steps from 0: k additions of the increment in the env's float type, done here once."""
        npdt = np.float32 if self.dtype == torch.float32 else np.float64
        t = np.zeros(self.max_steps + 2, dtype=npdt)
        inc = npdt(1.0 / self.max_steps)
        for k in range(1, t.size):
            t[k] = t[k - 1] + inc
        return torch.from_numpy(t).to(self.device)

    @property
    def state(self):
        """[B, R, N] observable rows as the reference's ``self.state`` holds them (row 0 = signed spins).  Rows that change
        everywhere every step are written on the first read after a step (rls_spin_materialize)."""
        if self._state_stale:
            _t.spin_materialize(self._env_handle, self._state, self._rows, self.current_step)
            self._state_stale = False
        return self._state

    # ---- dense adjacency only when somebody asks for it (N^2 floats)
    @property
    def matrix(self):
        """[N, N] (shared graph) or [B, N, N] (graph_generator): the couplings in the env's float type."""
        if self._matrix is None:
            csr = self.graph.csr
            m = np.zeros((self.n_spins, self.n_spins), dtype=np.float64)
            np.add.at(m, (np.repeat(np.arange(self.n_spins), np.diff(csr.rowptr)), csr.col), csr.wgt.astype(np.float64))
            self._matrix = torch.from_numpy(m).to(device=self.device, dtype=self.dtype)
        return self._matrix

    matrix_obs = matrix

    def _draw_matrix(self):
        m = torch.as_tensor(self.gg.get(), device=self.device).to(self.dtype).contiguous()
        if m.shape != (self.num_envs, self.n_spins, self.n_spins):
            raise ValueError(f"graph_generator.get() must return [{self.num_envs}, {self.n_spins}, {self.n_spins}], got {tuple(m.shape)}")
        return m

    def _round(self, v: float) -> float:
        """A host scalar as the env's float type sees it (the f32 env rounds python floats to f32 first)."""
        return float(np.float32(v)) if self.dtype == torch.float32 else float(v)

    def reset(self, spins=None):
        """spinsystem_PECO.py:150-195.  spins: optional [B, N] in the env's spin basis."""
        self.current_step = 0
        self._state_stale = False            # reset rewrites every row
        if self._time_table.numel() != self.max_steps + 2:      # max_steps was reassigned: the increments change
            self._time_table = self._build_time_table()
            self._env.time_table, self._env.table_len = self._time_table.data_ptr(), self._time_table.numel()
        B, N = self.num_envs, self.n_spins
        if spins is None:
            bits = ops.rand_spins(B, N, self._next_seed(), self.device, env_offset=self.env_offset)
            # no gauge fixing here: node 0 is a coin like the others (node 1 of a second keyed draw)
            bits[:, 0] = ops.rand_spins(B, 2, self._next_seed(), self.device, env_offset=self.env_offset)[:, 1]
        else:
            spins = torch.as_tensor(spins, device=self.device, dtype=self.dtype).reshape(B, N)
            # _format_spins_to_signed (spinsystem_PECO.py:548-557): 0/1 input under the BINARY basis -> 2 s - 1 (sic:
            # not the inverse of get_observation's (1 - s) / 2); already-signed input passes through
            signed = (2 * spins - 1) if (self.spin_basis == SpinBasis.BINARY and spins.min() >= 0) else spins
            bits = (signed > 0).contiguous()
        self._state[:, 0, :] = 2 * bits.to(self.dtype) - 1
        if self._dense:
            # a fresh matrix per env (spinsystem_PECO.py:150-170); graphs the reference rejects are drawn again
            for _ in range(64):
                self._matrix = self._draw_matrix()
                _t.spin_reset_dense(self._matrix, self._env_handle, self._state, self._rows, self.max_local_reward_available_,
                                    self._weight_sum_env, self._flags)
                f = self._flags.to(torch.int64)
                # bit 2: a zero maximum over all rows although nonzero ones exist -- the numpy env goes on with the maximum over those
                # (already in max_local), the batched env draws again
                again = (f & 1) & (~(f >> 2) & 1) if self._numpy_env_rules else f & 1
                # (a redraw is a decision about the WHOLE batch: every rank of a sharded one takes it together)
                flags = self._global_sum(torch.stack([again.max(), ((f >> 1) & 1).max()])).tolist()   # the one host read of a reset (the reference's .any() tests)
                if flags[1]:
                    raise ValueError("graph_generator.get() must return symmetric integer-valued matrices")
                if not flags[0]:
                    break
            else:
                raise ValueError("graph_generator keeps producing empty graphs / zero max local reward")
        else:
            # gains of all single flips: delta_i = s_i sum_j W_ij s_j = sum_j W_ij (x_i == x_j ? 1 : -1): the K3 kernel
            ops.maxcut_delta_all(self.graph, bits, out=self._delta)
            _t.spin_reset(self.graph.handle, self._env_handle, self._state, self._rows, self._max_local, self._weight_sum)
        self.best_obs_spins = self.best_spins          # (with a finite memory the best observable spins live bit-packed in the ring)
        return self.get_observation()

    def calculate_cut(self, spins=None):
        """cut = 1/4 * sum_ij W_ij (1 - s_i s_j)  (spinsystem_PECO.py:601-607) = (sum(W) - sum_i delta_i) / 4,
        exact in integers.  spins: None (the env's own) or signed [B, N] (used as given, like the reference: its
        basis conversion is commented out there)."""
        wsum = self._weight_sum_env.to(torch.int64) if self._dense else self._weight_sum
        if spins is None:
            return (wsum - self._delta.sum(dim=1)).to(self.dtype) / 4
        sp = torch.as_tensor(spins, device=self.device)
        if sp.dim() == 1:
            sp = sp[None, :]
        if sp.shape[-1] != self.n_spins or sp.dim() != 2:
            raise ValueError(f"spins must be [B, {self.n_spins}], got {tuple(sp.shape)}")
        if self._dense:
            # per-env matrices: a helper outside the step path, so the reference's own expression on the device
            if sp.shape[0] != self.num_envs:
                raise ValueError(f"spins must hold one row per env ({self.num_envs}), got {sp.shape[0]}")
            sf = sp.to(self._matrix.dtype)
            gains = torch.matmul(self._matrix, sf.unsqueeze(-1)).squeeze(-1) * sf
            return (wsum.to(self.dtype) - gains.sum(dim=-1).to(self.dtype)) / 4
        delta = ops.maxcut_delta_all(self.graph, (sp > 0).contiguous())        # K3: s_i sum_j W_ij s_j
        return (wsum - delta.sum(dim=1)).to(self.dtype) / 4

    def seed(self, seed=None):
        """spinsystem_PECO.py:299-300 (sic: the reference returns the attribute that set_seed stores)."""
        return self._seed

    def set_seed(self, seed):
        """spinsystem_PECO.py:302-304 (called by set_global_seed, ECO_S2V/src/agents/util.py:26-30, next to
        torch.manual_seed -- which is what seeds this env's reset draws)."""
        self._seed = seed
        np.random.seed(seed)

    def calculate_score(self, spins=None):
        return self.calculate_cut(spins)

    def _termination(self):
        """max(0, (current_step - max_steps) / horizon_length + 1), evaluated in the env's float type like the reference."""
        if self.dtype == torch.float32:
            return float(max(np.float32(0.0), np.float32((self.current_step - self.max_steps) / self.horizon_length) + np.float32(1)))
        return max(0.0, ((self.current_step - self.max_steps) / self.horizon_length) + 1)

    def _step_consts(self):
        """Everything a step passes that does not change from step to step (rebuilt when a public knob was reassigned)."""
        key = (self.reward_signal, self.norm_rewards, self.stag_punishment, self.basin_reward, self.max_steps)
        c = getattr(self, "_consts", None)
        if c is None or c["key"] != key:
            mode = _REWARD_MODE[self.reward_signal]
            if self._score_scale != 1.0:              # (only ever 2: the ABI's mode 3 is CUSTOM_BLS on a score in half units)
                assert self._score_scale == 2.0
                mode = 3 if mode == 2 else mode
            c = self._consts = dict(
                key=key, mode=mode,
                # (the ratio impr / (impr + eps) of CUSTOM_BLS is scale-free once eps is halved: only DENSE / BLS count double)
                div=(float(self.n_spins) if self.norm_rewards else 1.0) / (1.0 if mode == 3 else self._score_scale),
                tail=(self.stag_punishment is not None, self._round(self.stag_punishment or 0.0),
                      self.basin_reward is not None, self._round(self.basin_reward or 0.0)))
        return c

    def step(self, action):
        """spinsystem_PECO.py:306-486 -> (obs, reward [B], done bool [B])"""
        self.current_step += 1
        if self.current_step > self.max_steps:
            print("The environment has already returned done. Stop it!")
            raise NotImplementedError
        B = self.num_envs
        if not (torch.is_tensor(action) and action.dtype == torch.int64 and action.device == self.device and action.shape == (B,)
                and action.is_contiguous()):
            action = torch.as_tensor(action, device=self.device).to(torch.int64).reshape(B).contiguous()
        rew = torch.empty(B, dtype=self.dtype, device=self.device)
        c = self._step_consts()
        if self._dense:
            _t.spin_step_dense(self._matrix, self.max_local_reward_available_, self._env_handle, self._state, self._rows, action, rew,
                               self._visited_new, self._termination(), c["mode"], c["div"], self.current_step - 1, *c["tail"])
        else:
            _t.spin_step(self.graph.handle, self._env_handle, self._state, self._rows, action, rew, self._visited_new, self._max_local,
                         self._termination(), c["mode"], c["div"], self.current_step - 1, *c["tail"])
        self._state_stale = True
        done = torch.full((B,), self.current_step == self.max_steps, dtype=torch.bool, device=self.device)
        return self.get_observation(), rew, done

    def get_observation(self, out=None):
        """spinsystem_PECO.py:455: the observable rows (row 0 in the spin basis the agent sees) followed, with
        ``include_adjacency``, by the N rows of the shared matrix -- [B, R (+ N), N], ONE streaming kernel
        (rls_spin_observation; the reference clones the state, rewrites row 0 and concatenates a [B, N, N] expansion).
        ``out``: a caller-owned buffer of that shape to write into (a rollout ring), else a fresh tensor."""
        B, R, N = self._state.shape
        rows = R + (N if self.include_adjacency else 0)
        if out is None:
            out = torch.empty((B, rows, N), dtype=self.dtype, device=self.device)
        elif out.shape != (B, rows, N) or out.dtype != self.dtype or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous {self.dtype} tensor of shape {(B, rows, N)}")
        _t.spin_observation(self._env_handle, self._state, self._rows, self.current_step,
                            self.matrix if self.include_adjacency else None, self.spin_basis == SpinBasis.BINARY, out)
        return out

    def get_immeditate_rewards_avaialable(self, spins=None):
        return self._delta.to(self.dtype)

    def get_allowed_action_states(self):
        return (0, 1) if self.spin_basis == SpinBasis.BINARY else (1, -1)

    def get_best_cut(self):
        return self.best_score

    # ---- checkpoint of the env state: everything a later step depends on
    _STATE_KEYS = ("_state", "_delta", "score", "best_score", "best_spins", "_num_nonpos", "_dist_best", "_last_flip", "_scalars")
    _DENSE_KEYS = ("_matrix", "max_local_reward_available_", "_weight_sum_env")
    _HIST_KEYS = ("_packed", "_hash", "_hist", "_hist_hash")
    _MEM_KEYS = ("_packed", "_hash", "_mem_spins", "_mem_score")

    def _dict_keys(self):
        keys = self._STATE_KEYS + ("best_obs_score",) + (self._HIST_KEYS if self._use_hist else ()) + (self._DENSE_KEYS if self._dense else ())
        if self.memory_length is not None:
            keys += tuple(k for k in self._MEM_KEYS if k not in keys)
        return keys

    def state_dict(self):
        keys = self._dict_keys()
        self.state                                            # (materialises the lazy rows first)
        d = {k.lstrip("_"): getattr(self, k).clone() for k in keys}
        d["current_step"] = self.current_step
        d["format"] = self.STATE_DICT_FORMAT
        d["seeds"] = self._seeds.state_dict()
        return d

    STATE_DICT_FORMAT = 2     # 1 (round 2): no last_flip / scalars / best_obs_score -- the O(deg) step keeps those instead of five rows

    def load_state_dict(self, d):
        keys = self._dict_keys()
        missing = [k.lstrip("_") for k in keys if k.lstrip("_") not in d]
        if missing:
            raise ValueError(f"SpinSystem.load_state_dict: this checkpoint (format {d.get('format', 1)}) lacks {missing}; format "
                             f"{self.STATE_DICT_FORMAT} keeps per-node flip times and per-env scalars where format 1 kept the "
                             "observation rows, and they cannot be rebuilt from those rows: reset() the env instead")
        if "seeds" in d:
            self._seeds.load_state_dict(d["seeds"])
        for k in keys:
            getattr(self, k).copy_(d[k.lstrip("_")])     # in place: the kernel's pointer table stays valid
        self.current_step = int(d["current_step"])
        self._state_stale = False


class SpinSystemFactory:
    """SpinSystemFactory.get of spinsystem_PECO.py:16-47 (what ``ising_env.make("SpinSystem", ...)`` of core.py:9-16 calls, as
    train_PECO.py:75-86 does): the batched env on the generator's graphs -- and of spinsystem.py:24-60 (train_ECO.py:83-92):
    with a single-instance generator (no ``num_envs``, get() -> [N, N]) and no ``num_envs`` argument, the single-instance numpy
    surface ``SpinSystemUnbiased`` with every option its callers use: CUT or ENERGY, reversible or irreversible spins (S2V-DQN:
    train_S2V.py:37-47), NONE or PASS, infinite or finite memory.  The BATCHED env: OptimisationTarget.CUT, reversible spins,
    ExtraAction.NONE, infinite memory -- the reference's batched env cannot be constructed with ENERGY, an extra action or a
    finite memory, its irreversible reset fills env 0's rows and leaves the other envs' spins at 0 (facts recorded in
    tests/golden/spinsystem_options.npz / spinsystem_s2v.npz), and dqn_PECO.py:252 asserts NONE.  Unbiased graphs only (no
    caller builds a biased generator).  Anything else raises NotImplementedError instead of silently doing something different."""

    @staticmethod
    def get(graph_generator=None, max_steps=20, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.DENSE,
            extra_action=ExtraAction.PASS, optimisation_target=OptimisationTarget.ENERGY, spin_basis=SpinBasis.SIGNED,
            norm_rewards=False, memory_length=None, horizon_length=None, stag_punishment=None, basin_reward=None,
            reversible_spins=True, init_snap=None, seed=None, device=None, num_envs=None, if_greedy=False):
        single = num_envs is None and getattr(graph_generator, "num_envs", None) is None
        unsupported = [name for name, bad in (("extra_action", extra_action.name != "NONE" and not (single and extra_action.name == "PASS")),
                                              ("optimisation_target", optimisation_target.name != "CUT" and not (single and optimisation_target.name == "ENERGY")),
                                              ("memory_length", memory_length is not None and not single),
                                              ("reversible_spins", not reversible_spins and not single),
                                              ("init_snap", init_snap is not None),
                                              ("biased graphs", bool(getattr(graph_generator, "biased", False)))) if bad]
        if unsupported:
            raise NotImplementedError("SpinSystem on the device supports unbiased graphs and no init_snap; the batched env additionally only "
                                      "OptimisationTarget.CUT, reversible spins, ExtraAction.NONE and infinite memory; got " + ", ".join(unsupported))
        if seed is not None:
            np.random.seed(seed)                              # spinsystem_PECO.py:98-99 / spinsystem.py:95-96
        same = lambda enum, v: enum[v.name]                   # the reference's own enum members are accepted by name
        obs = [same(Observable, o) for o in observables]
        if single:                                            # if_greedy: accepted and ignored, as spinsystem.py:46 does
            return SpinSystemUnbiased(None, None, max_steps, obs, same(RewardSignal, reward_signal), same(SpinBasis, spin_basis),
                                      norm_rewards, horizon_length, stag_punishment, basin_reward, device,
                                      graph_generator=graph_generator, extra_action=same(ExtraAction, extra_action),
                                      memory_length=memory_length, reversible_spins=reversible_spins,
                                      optimisation_target=same(OptimisationTarget, optimisation_target))
        if num_envs is None:
            num_envs = graph_generator.num_envs
        return SpinSystem(None, None, num_envs, max_steps, obs, same(RewardSignal, reward_signal),
                          same(SpinBasis, spin_basis), norm_rewards, horizon_length, stag_punishment, basin_reward, device,
                          graph_generator=graph_generator)


def make(id2, *args, **kwargs):
    """ECO_S2V/src/envs/core.py:9-16."""
    if id2 == "SpinSystem":
        return SpinSystemFactory.get(*args, **kwargs)
    raise NotImplementedError()


class SpinSystemUnbiased:
    """The single-instance numpy surface of rlsolver/methods/ECO_S2V/src/envs/spinsystem.py:588-661
    (``SpinSystemUnbiased``; base class :62-520) on the HIP env above with one env in float64:

        reset(spins=None) -> obs  np.float64 [R + N, N]   (np.vstack((state, matrix)), :484-495)
        step(action: int) -> (obs, reward: float, done: bool, None)                      (:333-482)

    attrs: n_spins, max_steps, current_step, score, best_score, best_spins, state (np [R, N]), matrix.
    OptimisationTarget.CUT or ENERGY, reversible or irreversible spins, ExtraAction.NONE or PASS, infinite or finite memory,
    unbiased graphs (no caller in the reference builds a biased generator).

    ``reversible_spins=False`` -- what S2V-DQN trains and infers with (train_S2V.py:37-47, inference.py:59) -- starts an
    episode from all +1 (spinsystem.py:262-264), reports done as soon as no spin is +1 (:476-480) and answers
    get_allowed_action_states() with the one spin value an agent may still flip (:514-527); like the reference, step()
    itself does not refuse to flip a spin back.

    ``optimisation_target=OptimisationTarget.ENERGY`` (the default of SpinSystemFactory.get): score = -E = s'Ws / 2
    (:531-533, :644-647), immediate rewards -2 s (W s) (:498-499, :654-656).  -E = 2 cut_{-W}(s) + sum(W) / 2, so the
    HIP env runs as CUT on the NEGATED couplings with its rewards counted double (reward_div halved; CUSTOM_BLS as the
    ABI's mode 3): every normalised row is the same number, every score and reward the reference's bit for bit
    (tests/golden/spinsystem_s2v.npz), and the observation's matrix rows are negated back on the way out.

    With ``extra_action=ExtraAction.PASS`` (the reference's default) there are n_spins + 1 actions; action n_spins flips
    nothing (spinsystem.py:349-351).  The reference then carries a padding column through its arrays -- state [R, N + 1],
    matrix_obs [N + 1, N + 1], observation [R + N + 1, N + 1] -- whose entries are written by the row-wide assignments
    of its step (:417-447) and by nothing else; this class adds exactly that column on the way out (the kernels work on
    the N real spins).  ``memory_length`` = M: rewards and the two distance rows refer to the best of the last M
    scores / configurations instead of the best ever (:398-404)."""

    class _OneGraph:
        """A single-instance generator (get() -> [N, N] array, ECO_S2V/src/envs/util_envs.py:87-330) as a batch of one."""

        def __init__(self, gg, negate=False):
            self.gg, self.n_spins, self.biased, self.negate = gg, int(gg.n_spins), bool(getattr(gg, "biased", False)), negate

        def get(self):
            m = np.asarray(self.gg.get(), dtype=np.float64)
            return torch.as_tensor(0.0 - m if self.negate else m)[None]

    def __init__(self, mygraph, num_nodes: Optional[int], max_steps: int = 20,
                 observables: Sequence[Observable] = ECO_PECO_OBSERVABLES,
                 reward_signal: RewardSignal = RewardSignal.DENSE, spin_basis: SpinBasis = SpinBasis.SIGNED,
                 norm_rewards: bool = False, horizon_length: Optional[int] = None,
                 stag_punishment: Optional[float] = None, basin_reward: Optional[float] = None, device=None,
                 init_spins=None, graph_generator=None, extra_action: ExtraAction = ExtraAction.NONE,
                 memory_length: Optional[int] = None, reversible_spins: bool = True,
                 optimisation_target: OptimisationTarget = OptimisationTarget.CUT):
        """``graph_generator`` (instead of mygraph): a fresh graph at every reset, as the ECO / S2V training loops run their
        envs (train_ECO.py:83-94; any object with n_spins and get() -> [N, N] symmetric integer-valued array)."""
        if optimisation_target.name not in ("CUT", "ENERGY"):
            raise NotImplementedError(f"Optimisation target {optimisation_target} not recognised.")
        self.optimisation_target = OptimisationTarget[optimisation_target.name]
        self._energy = self.optimisation_target == OptimisationTarget.ENERGY
        self.reversible_spins = bool(reversible_spins)
        if getattr(graph_generator, "biased", False):
            raise NotImplementedError("biased graph generators are not supported")
        gg = self._OneGraph(graph_generator, negate=self._energy) if graph_generator is not None else None
        num_nodes = gg.n_spins if gg is not None and num_nodes is None else num_nodes
        if self._energy and mygraph is not None:
            mygraph = [(int(a), int(b), -w) for a, b, w in mygraph]
        self._env = SpinSystem(mygraph, num_nodes, 1, max_steps, observables, reward_signal, spin_basis, norm_rewards,
                               horizon_length, stag_punishment, basin_reward, device, include_adjacency=True,
                               dtype=torch.float64, graph_generator=gg, extra_action=ExtraAction[extra_action.name],
                               memory_length=memory_length, _numpy_env_rules=True, _score_scale=2.0 if self._energy else 1.0)
        self._env.reversible_spins, self._env.optimisation_target = self.reversible_spins, self.optimisation_target
        self._pass = self._env._pass
        self.extra_action, self.memory_length = self._env.extra_action, memory_length
        self.n_spins, self.max_steps, self.n_actions = num_nodes, max_steps, num_nodes + int(self._pass)
        self._last_pass = 0                     # the step of the last PASS (the padding column's "last flip")
        self._stepped = False
        self.observables = self._env.observables
        self.action_space, self.observation_space = self._env.action_space, self._env.observation_space
        if init_spins is not None:
            self.reset(init_spins)

    def _pad_rows(self, rows: np.ndarray, binary_row0: bool) -> np.ndarray:
        """[R, N] observable rows -> [R, N + 1] as the reference holds them under ExtraAction.PASS: the padding column takes what
        the ROW-WIDE assignments of step() put there (time since the last PASS, episode time, termination, greedy count --
        only once a step has run: reset writes it for the real spins alone, spinsystem.py:259-261 --, distance from the best
        score) and stays 0 elsewhere (spin, immediate reward, distance from the best state: [:n_spins] assignments)."""
        if not self._pass:
            return rows
        env = self._env
        out = np.zeros((rows.shape[0], rows.shape[1] + 1), dtype=rows.dtype)
        out[:, :-1] = rows
        table = env._time_table.cpu().numpy()
        for idx, ob in env.observables:
            if ob == Observable.TIME_SINCE_FLIP:
                out[idx, -1] = table[env.current_step - self._last_pass]
            elif ob in (Observable.EPISODE_TIME, Observable.TERMINATION_IMMANENCY, Observable.DISTANCE_FROM_BEST_SCORE):
                out[idx, -1] = rows[idx, 0]
            elif ob == Observable.NUMBER_OF_GREEDY_ACTIONS_AVAILABLE and self._stepped:
                out[idx, -1] = rows[idx, 0]
        if binary_row0:
            out[0, -1] = 0.5                     # get_observation maps the padding spin 0 to (1 - 0) / 2 (:487-489)
        return out

    def _obs(self, obs):
        o = obs[0].cpu().numpy()
        R, N = len(self._env.observables), self.n_spins
        if self._energy:
            o[R:] = 0.0 - o[R:]                  # the env ran on -W: the agent sees W
        if not self._pass:
            return o
        out = np.zeros((R + N + 1, N + 1), dtype=o.dtype)
        out[:R] = self._pad_rows(o[:R], self._env.spin_basis == SpinBasis.BINARY)
        out[R:R + N, :N] = o[R:]                 # matrix_obs: the couplings padded with a zero row and column (:222-225)
        return out

    def reset(self, spins=None):
        self._last_pass, self._stepped = 0, False
        if spins is not None:
            spins = np.asarray(spins, dtype=np.float64)[: self.n_spins][None, :]
        elif not self.reversible_spins:
            spins = np.ones((1, self.n_spins), dtype=np.float64)      # every spin may still be flipped (spinsystem.py:262-264; signed +1)
        return self._obs(self._env.reset(spins))

    def step(self, action):
        action = int(action)
        if self._pass and action == self.n_spins:
            self._last_pass = self._env.current_step + 1
        obs, rew, done = self._env.step(torch.tensor([action], dtype=torch.int64))
        self._stepped = True
        done = bool(done[0])
        if not self.reversible_spins:                                  # no more spins to flip --> done (:476-480)
            row0 = obs[0, 0]
            left = (row0 == 0) if self._env.spin_basis == SpinBasis.BINARY else (row0 > 0)
            done = done or not bool(left.any())
        return self._obs(obs), float(rew[0]), done, None

    def get_observation(self):
        return self._obs(self._env.get_observation())

    def _weight_sum(self) -> float:
        """sum of the DEVICE env's couplings over ordered pairs (= -sum(W) under ENERGY)."""
        return float(self._env._weight_sum_env[0]) if self._env._dense else float(self._env._weight_sum)

    def _score_out(self, v: float) -> float:
        """A score of the device env in the caller's units: -E = 2 cut_{-W} + sum(W) / 2 under ENERGY (exact: integers and halves)."""
        return 2.0 * v - self._weight_sum() / 2.0 if self._energy else v

    max_local_reward_available = property(lambda self: float(self._env.max_local_reward_available_[0]) * (2.0 if self._energy else 1.0))
    current_step = property(lambda self: self._env.current_step)
    score = property(lambda self: self._score_out(float(self._env.score[0])))
    best_score = property(lambda self: self._score_out(float(self._env.best_score[0])))
    best_spins = property(lambda self: self._env.best_spins[0].cpu().numpy())
    state = property(lambda self: self._pad_rows(self._env.state[0].cpu().numpy(), False))
    best_obs_score = property(lambda self: self._score_out(float(self._env.best_obs_score[0])))

    @property
    def matrix(self):
        m = (self._env.matrix[0] if self._env._dense else self._env.matrix).cpu().numpy()
        return 0.0 - m if self._energy else m

    def get_immeditate_rewards_avaialable(self, spins=None):
        d = self._env._delta[0].cpu().numpy().astype(np.float64)
        return 2.0 * d if self._energy else d     # -2 s (W s) = 2 s ((-W) s)  (spinsystem.py:498-499)

    def _signed(self, spins):
        """_format_spins_to_signed, spinsystem.py:548-557."""
        sp = np.asarray(spins, dtype=np.float64)[: self.n_spins]
        if self._env.spin_basis == SpinBasis.BINARY:
            if not np.isin(sp, [0, 1]).all():
                raise Exception("SpinSystem is configured for binary spins ([0,1]).")
            sp = 2 * sp - 1
        elif not np.isin(sp, [-1, 1]).all():
            raise Exception("SpinSystem is configured for signed spins ([-1,1]).")
        return sp

    def _device_cut(self, spins=None) -> float:
        return float(self._env.calculate_cut(None if spins is None else self._signed(spins)[None, :])[0])

    def calculate_cut(self, spins=None):
        """spinsystem.py:601-607: the env's own spins, or foreign ones in the env's basis (checked and converted like
        _format_spins_to_signed, :548-557).  (Under ENERGY the device env holds -W: cut_W = -cut_{-W}.)"""
        c = self._device_cut(spins)
        return 0.0 - c if self._energy else c

    def calculate_energy(self, spins=None):
        """spinsystem.py:590-599: E = -s'Ws / 2."""
        c = self._device_cut(spins)
        return 0.0 - (2.0 * c - self._weight_sum() / 2.0) if self._energy else 2.0 * c - self._weight_sum() / 2.0

    def calculate_score(self, spins=None):
        """spinsystem.py:529-536."""
        return -1. * self.calculate_energy(spins) if self._energy else self.calculate_cut(spins)

    def seed(self, seed=None):
        return self._env.seed()

    def set_seed(self, seed):
        self._env.set_seed(seed)

    def get_best_cut(self):
        if self._energy:                                              # spinsystem.py:609-613
            raise NotImplementedError("Can't return best cut when optimisation target is set to energy.")
        return self.best_score

    def get_allowed_action_states(self):
        """spinsystem.py:514-527: both spin values while spins are reversible, else the value of a spin not flipped yet."""
        if self.reversible_spins:
            return self._env.get_allowed_action_states()
        return 0 if self._env.spin_basis == SpinBasis.BINARY else 1
