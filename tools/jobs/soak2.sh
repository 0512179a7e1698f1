# a longer soak: random graphs (seeds 700..2500) and random fused 10^3 blocks:  gpurun -- 'bash tools/jobs/soak2.sh'
mkdir -p gpurun_out/soak
timeout 2400 python tools/fuzz_soak.py 700 2500 2>&1 | tail -2 | tee gpurun_out/soak/soak2.txt
timeout 900 python tools/fuzz_wfused.py 2>&1 | tail -2 | tee gpurun_out/soak/wfused.txt
