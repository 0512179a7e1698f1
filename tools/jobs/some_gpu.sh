# a subset of the gpu suite:  gpurun -- 'bash tools/jobs/some_gpu.sh tests/test_gpu_x.py ...'
timeout 1200 python -m pytest "$@" -m gpu -q 2>&1 | tail -8
