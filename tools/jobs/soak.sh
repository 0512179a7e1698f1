# seeded random-graph soak over many more seeds than the test suite runs, and which kernels they reach:  gpurun -- 'bash tools/jobs/soak.sh [lo hi]'
mkdir -p gpurun_out/soak
timeout 1500 python tools/fuzz_soak.py ${1:-48} ${2:-700} 2>&1 | tail -3 | tee gpurun_out/soak/soak.txt
timeout 300 python tools/fuzz_coverage.py 2>&1 | tail -30 | tee gpurun_out/soak/coverage.txt
