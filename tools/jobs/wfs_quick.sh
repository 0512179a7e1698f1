# parity + rate of the split 10^3 layer:  gpurun -- 'bash tools/jobs/wfs_quick.sh'
mkdir -p gpurun_out/wfs
timeout 600 python -m pytest tests/test_gpu_conv_wfsplit.py -x -q 2>&1 | tail -5
for v in "TH_WF_SPLIT=0" "TH_WF_SPLIT=1" "TH_WF_SPLIT=1" "TH_WF_DBG=1" "TH_WF_DBG=2" "TH_WF_DBG=4" "TH_WF_DBG=16" "TH_WF_DBG=20" "TH_WF_DBG=21" "TH_WF_DBG=32"; do
  echo "== $v"; env $v TH_GUARD=0 timeout 120 python tools/bench_layer.py 10 32 64 3 8192 1 2>&1 | grep -o '"ms_per_4096": [0-9.]*' | tr '\n' ' '; echo
done 2>&1 | tee gpurun_out/wfs/layer_rate_quick.txt
bash tools/jobs/wfs_pmc.sh 2>&1 | tee gpurun_out/wfs/pmc_quick.txt
