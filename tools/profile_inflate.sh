#!/bin/bash
# Kernel trace of the GPU ingest path (th_h5_decode_device: k_inflate_tokens / k_lz_resolve) at predict.py's batch size, on the GPU box:
#   gpurun --timeout 600 -- 'bash tools/profile_inflate.sh' ; the summary lands in gpurun_out/inflate_rocprof.txt
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_inflate
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_inflate -o o -- python "$ROOT/tools/bench_h5_decode.py" 16384 4096 > "$ROOT/gpurun_out/inflate_bench.json" 2> /tmp/prof_inflate.log
f=$(find /tmp/prof_inflate -name "*.db" | head -1)
if [ -n "$f" ]; then (cd "$ROOT" && python tools/rocpd_summary.py --kernel-trace "$f" < /dev/null > "$ROOT/gpurun_out/inflate_rocprof.txt"); head -12 "$ROOT/gpurun_out/inflate_rocprof.txt" | cut -c1-200; else tail -5 /tmp/prof_inflate.log; fi
tail -1 "$ROOT/gpurun_out/inflate_bench.json"
