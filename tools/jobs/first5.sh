# conv_first5: parity tests, then the layer inside ProDCoNN-synth (+ knock-outs named on the command line)
#   gpurun --timeout 900 -- 'bash tools/jobs/first5.sh [TH_FIRST_DBG=1 ...]'
timeout 300 python -m pytest tests/test_gpu_conv_first5.py -q -x 2>&1 | tail -3
for v in "" "$@"; do
  echo "VAR=$v"
  env TH_GUARD=0 $v timeout 200 python tools/plan_report.py --measure prodconn 2>/dev/null | grep -E "measured:|conv3d|dense"
done
