#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r06
# 1. WALL-CLOCK tables, no profiler attached: tools/bench_configs.py (every kernel of every BASELINE config, full iteration counts)
#    -> ${P}_configs.jsonl, and bench.py three ways (default 2000-step regions, the driver's 20-step regions, --via-env).
# 2. PROFILED passes of the same commands, kept apart (a profiler roughly doubles every launch-bound row: VERDICT r4 found the
#    r04 table of wall-clock rows replaced by the kernel-trace pass's stdout) -> ${P}_configs_profiled.jsonl:
#    three SEPARATE rocprofv3 passes per command (kernel trace + stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE -- the two TCC
#    counters do not fit one pass, and counters are never combined with trace domains other than kernel-trace).
# 3. Four SQ counter passes over bench_configs.py --profile (tools/sweeps/pmc_passes.sh).
# Outputs land under gpurun_out/<prefix>_*; tools/collect_profiles.sh <prefix> condenses them into profiles/ (run it here, after gpurun).
set -u
P=${1:-r06}
R=$(pwd)
export TMPDIR=/tmp
python3 $R/tools/bench_configs.py > $R/gpurun_out/${P}_configs.jsonl 2> $R/gpurun_out/${P}_configs.err
python3 $R/bench.py > $R/gpurun_out/${P}_bench_n1.json 2> $R/gpurun_out/${P}_bench_n1.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${P}_bench_20steps.json 2>> $R/gpurun_out/${P}_bench_n1.err
python3 $R/bench.py --via-env --no-cpu-baseline --no-config5 --no-configs > $R/gpurun_out/${P}_bench_via_env.json 2>> $R/gpurun_out/${P}_bench_n1.err
bash $R/tools/timing/exchange_trace.sh $P > $R/gpurun_out/${P}_exchange.log 2>&1      # the exchange: probe + kernel trace
cd /tmp
BENCH="python3 $R/bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-configs"
CFG="python3 $R/tools/bench_configs.py --profile"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_bench_kt -o $P -- $BENCH > $R/gpurun_out/${P}_bench_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${P}_bench_fetch -o $P -- $BENCH > $R/gpurun_out/${P}_bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${P}_bench_write -o $P -- $BENCH > $R/gpurun_out/${P}_bench_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_cfg_kt -o $P -- $CFG > $R/gpurun_out/${P}_configs_profiled.jsonl 2> $R/gpurun_out/${P}_cfg_kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${P}_cfg_fetch -o $P -- $CFG > $R/gpurun_out/${P}_cfg_fetch.jsonl 2> $R/gpurun_out/${P}_cfg_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${P}_cfg_write -o $P -- $CFG > $R/gpurun_out/${P}_cfg_write.jsonl 2> $R/gpurun_out/${P}_cfg_write.err
cd $R
bash tools/sweeps/pmc_passes.sh ${P}_sq tools/bench_configs.py --profile > /dev/null 2>&1
# the raw per-dispatch traces are large; keep the stats and the counter tables
find gpurun_out/${P}_* -name "*_kernel_trace.csv" -size +24M -delete; find gpurun_out/${P}_* -name "*.db" -delete
ls -la gpurun_out/ | grep ${P}_ | head -40
