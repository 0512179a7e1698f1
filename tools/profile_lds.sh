#!/bin/bash
# LDS counters (bank conflicts, active cycles, instruction counts, waits) of ONE layer run through the planner:
#   gpurun --timeout 600 -- 'bash tools/profile_lds.sh 10 64 16 3 8192'      (arguments of tools/bench_layer.py)
# One rocprofv3 --pmc pass with --kernel-trace only; the table is printed by tools/pmc_table.py.
ARGS=${*:-10 64 16 3 8192}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_lds; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/lds -o lds -- python $ROOT/tools/bench_layer.py $ARGS > $OUT/lds.log 2>&1
cd $ROOT
python tools/pmc_table.py $(find $OUT -name '*.db' | head -1) 2>&1 | head -12
rm -rf $OUT/*/*.db $OUT/*/*/*.db
