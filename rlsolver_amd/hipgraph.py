"""Launch-bound inner loops as one hipGraph.

At small batches a gym step is ~3 us of kernel behind ~8 us of launch path (ctypes call + HIP launch): a
rollout of T steps over pre-allocated buffers is captured once and replayed with a single launch.  The kernels
are enqueued on torch's current stream, so torch's own graph object (HIP graphs on ROCm) records them."""
from __future__ import annotations

from typing import Callable

import torch


class CapturedLaunches:
    """Record everything ``fn()`` enqueues on the current stream and replay it with one graph launch.

    ``fn`` must be replay-safe: fixed tensors (it may update them in place), no host synchronisation, no
    allocation.  It is run ``warmup`` times eagerly first (lazy one-off initialisation -- kernel attributes,
    device queries -- must not happen under capture), so give it buffers whose content may be overwritten."""

    def __init__(self, fn: Callable[[], None], device, warmup: int = 1):
        self.device = torch.device(device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream(self.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            fn()

    def replay(self) -> None:
        self.graph.replay()
