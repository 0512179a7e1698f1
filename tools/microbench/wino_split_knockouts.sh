#!/bin/bash
# What limits k_wino_gemm_b3 (csrc/conv_wino.hip)?  Knock-outs in separate processes (TH_WINO_DBG: 1 no slab loads, 2 no M stores,
# 4 no weight loads; results are WRONG when set) + the SQ counters of the shipped form, 128 -> 256 channels.
#   gpurun --timeout 900 -- 'bash tools/microbench/wino_split_knockouts.sh'
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/wino_split
mkdir -p "$OUT"
cd "$ROOT"
: > "$OUT/knockouts.txt"
for dbg in 0 1 2 4 3 7; do
  for var in 0 1; do
    echo "== 128->256 TH_WINO_DBG=$dbg TH_WINO_B3VAR=$var" >> "$OUT/knockouts.txt"
    TH_WINO_DBG=$dbg TH_WINO_B3VAR=$var timeout 300 python tools/bench_layer.py 5 128 256 3 8192 2>&1 | grep gemm | python -c "import sys,json; [print({k:v for k,v in json.loads(l).items() if k in ('ms_per_4096','tflops_algo')}) for l in sys.stdin]" >> "$OUT/knockouts.txt"
  done
done
cat "$OUT/knockouts.txt"
bash tools/profile_layer.sh r05_b3_c4 5 128 256 3 8192 > "$OUT/profile_b3_c4.txt" 2>&1
tail -30 "$OUT/profile_b3_c4.txt"
