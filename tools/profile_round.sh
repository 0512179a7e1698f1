#!/bin/bash
# Profile recipe behind profiles/<tag>_rocprof.txt and profiles/pmc_latest.json (run on the GPU box):
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01_d'
# Four separate rocprofv3 runs of the same short bench command: kernel trace (+stats), then one --pmc
# pass per counter group (never combined with a trace domain other than --kernel-trace).
set -u
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 1 --frames 20480 --no-cpu-baseline --no-profile --no-extras"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- $CMD > "$OUT/kt.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write -- $CMD > "$OUT/write.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/sq" -o sq -- $CMD > "$OUT/sq.log" 2>&1
cd "$ROOT"
find "$OUT" -name '*.db' | sort > "$OUT/dbs.txt"
KT=$(grep '/kt/' "$OUT/dbs.txt" | head -1); FE=$(grep '/fetch/' "$OUT/dbs.txt" | head -1)
WR=$(grep '/write/' "$OUT/dbs.txt" | head -1); SQ=$(grep '/sq/' "$OUT/dbs.txt" | head -1)
python tools/rocpd_summary.py --kernel-trace "$KT" --pmc FETCH_SIZE="$FE" --pmc WRITE_SIZE="$WR" \
    --json "$OUT/pmc_latest.json" > "$OUT/summary.txt" 2> "$OUT/summary.err"
[ -n "$SQ" ] && python tools/pmc_table.py "$SQ" > "$OUT/sq_table.txt" 2>> "$OUT/summary.err"
# every plan step's roofline fraction recomputed from a kernel trace alone (FLOPs x frames / avg us / 157.3)
timeout 600 python tools/roofline_table.py --chunk 4096 --reps 5 timed densecpd timed_rotamer prodconn > "$OUT/roofline_table.txt" 2>> "$OUT/summary.err"
# BASELINE config 5 (the sampler) under the kernel trace
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/sampler" -o sampler -- python $ROOT/tools/bench_sampler.py > "$OUT/sampler_bench.json" 2> "$OUT/sampler.log"
cd "$ROOT"
SM=$(find "$OUT/sampler" -name '*.db' | head -1)
[ -n "$SM" ] && python tools/rocpd_summary.py --kernel-trace "$SM" > "$OUT/sampler_rocprof.txt" 2>> "$OUT/summary.err"
# the bench line itself, default settings, outside the profiler
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -1 "$OUT/bench_default.json"
rm -rf "$OUT"/*/*.db "$OUT"/*/*/*.db 2>/dev/null   # databases are large; the summaries are what we keep
