#!/bin/bash
# Per-kernel profile of ONE convolution layer run through the planner (tools/bench_layer.py):
#   gpurun --timeout 600 -- 'TH_WINOGRAD=1 bash tools/profile_layer.sh r04_wino_c4 5 128 256 3 8192'
# kernel trace (+ stats), FETCH_SIZE / WRITE_SIZE and SQ counters in separate rocprofv3 passes.
set -u
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_layer.py $*"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- $CMD > "$OUT/kt.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write -- $CMD > "$OUT/write.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/sq" -o sq -- $CMD > "$OUT/sq.log" 2>&1
cd "$ROOT"
find "$OUT" -name '*.db' | sort > "$OUT/dbs.txt"
KT=$(grep '/kt/' "$OUT/dbs.txt" | head -1); FE=$(grep '/fetch/' "$OUT/dbs.txt" | head -1)
WR=$(grep '/write/' "$OUT/dbs.txt" | head -1); SQ=$(grep '/sq/' "$OUT/dbs.txt" | head -1)
python tools/rocpd_summary.py --kernel-trace "$KT" --pmc FETCH_SIZE="$FE" --pmc WRITE_SIZE="$WR" > "$OUT/summary.txt" 2> "$OUT/summary.err"
[ -n "$SQ" ] && python tools/pmc_table.py "$SQ" > "$OUT/sq_table.txt" 2>> "$OUT/summary.err"
tail -3 "$OUT/kt.log"
cat "$OUT/summary.txt"
cat "$OUT/sq_table.txt"
rm -rf "$OUT"/*/*.db "$OUT"/*/*/*.db 2>/dev/null
