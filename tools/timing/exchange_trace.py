"""The episode-boundary exchange under `rocprofv3 --kernel-trace`: a 1-rank RCCL group (the caller exports
RLS_FORCE_PG=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR MASTER_PORT before rocprofv3 -- no `env` hop), K exchanges of
rlsolver_amd.dist.BestExchange, then K calls of dist.global_best (exchange + unpack), then K calls of dist.global_best(want_solution=True, env_offset=...)
(C1 + C2: + rls_winner_message, the SUM all-reduce, rls_winner_unpack).  Prints the counts the trace should show;
tools/timing/exchange_summary.py turns the trace into profiles/rNN_exchange.json.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r06_exchange_kt -o r06 -- python3 $R/tools/timing/exchange_trace.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

from rlsolver_amd import dist as rdist

K = 100
rank, local_rank, world = rdist.init_from_env()
dev = torch.device("cuda", local_rank)
obj = (torch.arange(1 << 16, dtype=torch.int64, device=dev) * 7919 % 10007).to(torch.int32)
torch.cuda.synchronize(dev)
ex = rdist.BestExchange(dev, depth=1 << 20)           # (no periodic flag assert inside the counted stretch)
for _ in range(K):
    key = ex.exchange(obj)
torch.cuda.synchronize(dev)
for _ in range(K):
    best, owner, _ = rdist.global_best(obj)
torch.cuda.synchronize(dev)
xs = (torch.arange(1 << 16, device=dev)[:, None] + torch.arange(2000, device=dev)[None, :]) % 3 == 0     # [2^16, 2000] bool rows
torch.cuda.synchronize(dev)
for _ in range(K):
    best, owner, bx, gi = rdist.global_best(obj, xs, want_solution=True, env_offset=0)      # C1 + C2: key, unpack, message, unpack
torch.cuda.synchronize(dev)
ok_c2 = torch.equal(bx, xs[int(obj.argmax())]) and int(gi) == int(obj.argmax())
ex.check()
ok = int(ex.unpack(key)[0]) == int(obj.max()) == int(best) and ok_c2
dist.barrier(device_ids=[local_rank])
dist.destroy_process_group()
print("EXCHANGE_TRACE " + json.dumps({"exchanges": K, "global_best_calls": K, "global_best_with_solution_calls": K, "backend": "nccl", "world": world, "ok": bool(ok)}), flush=True)
