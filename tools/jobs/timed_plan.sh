# TIMED-synth's measured plan (optionally with knobs):  gpurun -- 'bash tools/jobs/timed_plan.sh [VAR=val ...]'
timeout 300 python -m pytest tests/test_gpu_cnn.py -q -x -m gpu 2>&1 | tail -2
for v in "" "$@"; do
  echo "VAR=$v"
  env $v timeout 300 python tools/plan_report.py --measure timed 2>/dev/null | grep -E "measured:|conv3d"
done
