#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02
# Three SEPARATE passes per command (kernel trace + stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE -- the two TCC
# counters do not fit one pass, and counters are never combined with trace domains other than kernel-trace),
# for the headline bench.py and for tools/bench_configs.py (every kernel of every BASELINE config).
# Outputs land under gpurun_out/<prefix>_*; tools/summarize_prof.py condenses them into profiles/.
set -u
P=${1:-r02}
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --steps 500 --warmup 20 --no-cpu-baseline"
CFG="python3 $R/tools/bench_configs.py --profile"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_bench_kt -o $P -- $BENCH > $R/gpurun_out/${P}_bench_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${P}_bench_fetch -o $P -- $BENCH > $R/gpurun_out/${P}_bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${P}_bench_write -o $P -- $BENCH > $R/gpurun_out/${P}_bench_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_cfg_kt -o $P -- $CFG > $R/gpurun_out/${P}_cfg_kt.jsonl 2> $R/gpurun_out/${P}_cfg_kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${P}_cfg_fetch -o $P -- $CFG > $R/gpurun_out/${P}_cfg_fetch.jsonl 2> $R/gpurun_out/${P}_cfg_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${P}_cfg_write -o $P -- $CFG > $R/gpurun_out/${P}_cfg_write.jsonl 2> $R/gpurun_out/${P}_cfg_write.err
cd $R
# the raw per-dispatch traces are large; keep the stats and the counter tables
find gpurun_out/${P}_* -name "*_kernel_trace.csv" -size +24M -delete; find gpurun_out/${P}_* -name "*.db" -delete
ls -la gpurun_out/${P}_*
