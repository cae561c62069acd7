#!/bin/bash
# usage: pmc_passes.sh <outdir-under-gpurun_out> <script> [args...]; one rocprofv3 --pmc pass per counter group
R=$PWD; export PYTHONPATH=$R; OUT=$R/gpurun_out/$1; shift; S=$R/$1; shift
cd /tmp; export TMPDIR=/tmp
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES"; do
  rocprofv3 --pmc $c -d $OUT -o p$i --output-format csv -- python3 $S "$@" > /dev/null 2>&1
  i=$((i+1))
done
ls $OUT
