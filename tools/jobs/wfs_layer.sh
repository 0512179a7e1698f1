mkdir -p gpurun_out/wfs
timeout 900 python -m pytest tests/test_gpu_conv_wfsplit.py -x -q 2>&1 | tail -25
for v in "TH_WF_SPLIT=0" "TH_WF_SPLIT=1" "TH_WF_DBG=1" "TH_WF_DBG=2" "TH_WF_DBG=4" "TH_WF_DBG=8" "TH_WF_DBG=16" "TH_WF_DBG=3" "TH_WF_DBG=11" "TH_WF_DBG=31"; do
  echo "== $v"; env $v TH_GUARD=0 timeout 120 python tools/bench_layer.py 10 32 64 3 8192 1 2>&1 | grep -o '"label": "[^"]\{0,40\}\|"ms_per_4096": [0-9.]*' | tr '\n' ' '; echo
done 2>&1 | tee gpurun_out/wfs/layer_rate_v1.txt
