#!/bin/bash
# Gate for the bf16x3 split-operand Winograd GEMM (csrc/conv_wino.hip, k_wino_gemm_b3) — run on the GPU box:
#   gpurun --timeout 1500 -- 'bash tools/microbench/wino_split.sh'
# 1. parity: tests/test_gpu_wino.py with the split GEMM (the default) and the single-layer error table of both forms
# 2. rate:   every Winograd layer shape of TIMED-synth / TIMED-rotamer with TH_WINO_SPLIT=0 (fp32 MFMA) and =1 (both A-prefetch variants)
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/wino_split
mkdir -p "$OUT"
cd "$ROOT"
export TH_WINOGRAD=1
( timeout 900 python -m pytest tests/test_gpu_wino.py -x -q -m gpu 2>&1 | tail -15 ) > "$OUT/pytest_wino.txt"
for sp in 0 1; do
  echo "== TH_WINO_SPLIT=$sp" >> "$OUT/layer_error.txt"
  TH_WINO_SPLIT=$sp timeout 600 python tests/wino_layer_error.py >> "$OUT/layer_error.txt" 2>&1
done
for shape in "64 128" "128 128" "128 256" "256 338"; do
  set -- $shape
  for cfg in "0 0" "1 0" "1 1"; do
    set -- $shape $cfg
    echo "== cin $1 cout $2 TH_WINO_SPLIT=$3 TH_WINO_B3VAR=$4" >> "$OUT/layer_rate.txt"
    TH_WINO_SPLIT=$3 TH_WINO_B3VAR=$4 timeout 300 python tools/bench_layer.py 5 $1 $2 3 8192 2>&1 | grep -v "^$" >> "$OUT/layer_rate.txt"
  done
done
cat "$OUT/pytest_wino.txt" "$OUT/layer_error.txt" "$OUT/layer_rate.txt"
