# rate of the split 10^3 layer under the knobs given as arguments:  gpurun -- 'bash tools/jobs/wfs_var.sh TH_WF_DBG=64 TH_WF_DBG=128'
mkdir -p gpurun_out/wfs
for v in "TH_WF_SPLIT=1" "$@"; do
  echo "== $v"; env $v TH_GUARD=0 timeout 120 python tools/bench_layer.py 10 32 64 3 8192 1 2>&1 | grep -o '"ms_per_4096": [0-9.]*' | tr '\n' ' '; echo
done 2>&1 | tee -a gpurun_out/wfs/layer_rate_var.txt
