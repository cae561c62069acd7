#!/bin/bash
# After `gpurun -- tools/profile_round.sh rNN`: condense gpurun_out/rNN_* into the tracked files under profiles/.
#   tools/collect_profiles.sh r05
set -eu
P=${1:-r06}
cp gpurun_out/${P}_configs.jsonl profiles/${P}_configs.jsonl                       # WALL-CLOCK rows (no profiler attached)
cp gpurun_out/${P}_configs_profiled.jsonl profiles/${P}_configs_profiled.jsonl     # the same table under rocprofv3 --kernel-trace: never compare the two
for k in n1 20steps via_env; do tail -1 gpurun_out/${P}_bench_${k}.json > profiles/${P}_bench_${k}.json; done
python3 tools/summarize_prof.py ${P}_step --kt gpurun_out/${P}_bench_kt --fetch gpurun_out/${P}_bench_fetch --write gpurun_out/${P}_bench_write \
    --prefix $P --cmd "python3 bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-configs" --bench-json gpurun_out/${P}_bench_kt.log
python3 tools/sq_table.py $P gpurun_out/${P}_sq
python3 tools/kernel_table.py $P --src gpurun_out/${P}
python3 tools/timing/exchange_summary.py $P
ls -la profiles | grep ${P}_
