# conv_gl with the input prologue: tests, then DenseCPD-synth's plan with and without it
timeout 400 python -m pytest tests/test_gpu_conv_gl.py tests/test_gpu_cnn.py -q -x -m gpu 2>&1 | tail -4
for v in TH_CONV_GL=1 TH_CONV_GL=0; do
  echo "== $v"
  env $v timeout 300 python tools/plan_report.py --measure densecpd 2>/dev/null | grep -E "measured:|k_conv_gl|conv3d_2[0-6] "
done
