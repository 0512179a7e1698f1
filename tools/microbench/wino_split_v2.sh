#!/bin/bash
# A/B of the split-GEMM variants (TH_WINO_B3VAR: 0 one tile per workgroup, 2 persistent + double-buffered) — GPU box
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/wino_split
mkdir -p "$OUT"
cd "$ROOT"
( TH_WINO_B3VAR=2 timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_bench_config.py -x -q -m gpu 2>&1 | tail -8 ) > "$OUT/pytest_v2.txt"
: > "$OUT/layer_rate_v2.txt"
for shape in "64 128" "128 128" "128 256" "256 338"; do
  for var in 0 2; do
    set -- $shape
    echo "== cin $1 cout $2 TH_WINO_B3VAR=$var" >> "$OUT/layer_rate_v2.txt"
    TH_WINO_B3VAR=$var timeout 300 python tools/bench_layer.py 5 $1 $2 3 8192 2>&1 | grep gemm | python -c "import sys,json; [print({k:v for k,v in json.loads(l).items() if k in ('ms_per_4096','tflops_algo')}) for l in sys.stdin]" >> "$OUT/layer_rate_v2.txt"
  done
done
for dbg in 1 2 3 7; do
    echo "== 128->256 B3VAR=2 TH_WINO_DBG=$dbg" >> "$OUT/layer_rate_v2.txt"
    TH_WINO_DBG=$dbg TH_WINO_B3VAR=2 timeout 300 python tools/bench_layer.py 5 128 256 3 8192 2>&1 | grep gemm | python -c "import sys,json; [print({k:v for k,v in json.loads(l).items() if k in ('ms_per_4096','tflops_algo')}) for l in sys.stdin]" >> "$OUT/layer_rate_v2.txt"
done
cat "$OUT/pytest_v2.txt" "$OUT/layer_rate_v2.txt"
