# conv_gl: parity tests, ProDCoNN-synth's plan, then every benchmark model with every eligible layer on it (TH_CONV_GL=2) beside the default
timeout 400 python -m pytest tests/test_gpu_conv_gl.py tests/test_gpu_dense_gemm.py -q -x 2>&1 | tail -15
timeout 200 python tools/plan_report.py --measure prodconn 2>/dev/null | grep -E "measured:|conv3d|dense"
for mdl in timed densecpd rotamer; do
  for v in TH_CONV_GL=1 TH_CONV_GL=2; do
    echo "== $mdl $v"
    env TH_GUARD=0 $v timeout 300 python tools/plan_report.py --measure $mdl 2>/dev/null | grep -E "measured:|k_conv_gl"
  done
done
