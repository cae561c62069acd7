"""The duplicate MaxCut simulators of rlsolver/envs/env_k_spin.py (SimulatorMaxcutAutoregressive :218,
MaxcutSimulatorReinforce :280, SimulatorGraphMaxCut :497) -- three more copies of the same edge-index
objective as env_L2A.EnvMaxcut, differing only in constructor keywords and in the name of the random
initialiser (``generate_solutions_randomly``).  All of them run on the same HIP kernels here.
(SimulatorMaxcut :62 -- a probabilistic relaxation -- and the dense relaxed EnvMaxcut :338 are out of
scope: SURVEY.md section 2 row 5.)
"""
from __future__ import annotations

import torch as th

from ..graph import MyGraph, load_mygraph2
from .env_L2A import EnvMaxcut as _EnvMaxcut


class _SimulatorBase(_EnvMaxcut):
    def generate_solutions_randomly(self, num_sims):
        """env_k_spin.py:274-278 / :332-336 / :554-558"""
        return self.generate_xs_randomly(num_sims)

    @property
    def adjacency_matrix(self):
        return self.adjacency_bool


class SimulatorGraphMaxCut(_SimulatorBase):
    def __init__(self, sim_name: str = 'max_cut', graph: MyGraph = (), device=th.device('cpu'),
                 if_bidirectional: bool = False, **shard):      # shard: env_offset / seed / group (rlsolver_amd/seeding.py)
        super().__init__(sim_name=sim_name, mygraph=graph, device=device, if_bidirectional=if_bidirectional, **shard)


class MaxcutSimulatorReinforce(_SimulatorBase):
    def __init__(self, graph: MyGraph, device=th.device('cpu'), if_bidirectional: bool = False, **shard):
        super().__init__(sim_name='max_cut', mygraph=graph, device=device, if_bidirectional=if_bidirectional, **shard)


class SimulatorMaxcutAutoregressive(_SimulatorBase):
    def __init__(self, graph_name: str, device=th.device('cpu'), if_bidirectional: bool = False, **shard):
        super().__init__(sim_name=graph_name, mygraph=load_mygraph2(graph_name=graph_name), device=device,
                         if_bidirectional=if_bidirectional, **shard)
