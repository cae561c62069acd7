#!/bin/bash
# The exchange's evidence (run through gpurun from the repo root): tools/timing/exchange_trace.sh r06
#  1. the probe alone (what bench.py's N = 1 line embeds) -> gpurun_out/<P>_exchange_probe.json
#  2. rocprofv3 --kernel-trace of tools/timing/exchange_trace.py -> gpurun_out/<P>_exchange_kt
# then here: python tools/timing/exchange_summary.py <P>  ->  profiles/<P>_exchange.json
P=${1:-r06}
R=$(pwd)
export TMPDIR=/tmp RLS_FORCE_PG=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29617 HSA_ENABLE_IPC_MODE_LEGACY=0
python3 $R/bench.py --exchange-probe > $R/gpurun_out/${P}_exchange_probe.json 2> $R/gpurun_out/${P}_exchange_probe.err
export MASTER_PORT=29618
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${P}_exchange_kt -o $P -- python3 $R/tools/timing/exchange_trace.py > $R/gpurun_out/${P}_exchange_kt.log 2>&1
cd $R
tail -2 gpurun_out/${P}_exchange_probe.json; tail -3 gpurun_out/${P}_exchange_kt.log
