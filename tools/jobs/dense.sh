# dense_gemm: parity tests, then ProDCoNN-synth's plan
timeout 300 python -m pytest tests/test_gpu_dense_gemm.py -q -x 2>&1 | tail -15
timeout 200 python tools/plan_report.py --measure prodconn 2>/dev/null | grep -E "measured:|conv3d|dense"
