# predict.py end to end from a float32 pack, dense vs sparse transport:  gpurun -- 'bash tools/jobs/e2e_sparse.sh [frames]'
mkdir -p gpurun_out/e2e
N=${1:-100000}
TIMED_PIPELINE_TRACE=1 timeout 800 python - "$N" <<'PY' 2>&1 | tee gpurun_out/e2e/sparse_vs_dense.txt
import json, os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "timed-design_amd")
import bench_legs
from timed_hip import synth
cfg, w = synth.timed_synth(20)
n = int(sys.argv[1])
r = bench_legs.predict_py_e2e(cfg, w, n_pack=n, n_hdf5=0, n_rotamer=0)
print(json.dumps({k: v for k, v in r.items() if "framepack" in k or "sparsify" in k}, indent=1))
PY
