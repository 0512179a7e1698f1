# kernel-level profile of predict.py from a sparse pack:  gpurun -- 'bash tools/jobs/e2e_sparse_prof.sh'
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/e2e; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/run_sparse.py <<'PY'
import os, sys, time, warnings
from pathlib import Path
ROOT = sys.argv[1]
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
import bench_legs, predict
from timed_hip import synth, pack, framepack
cfg, w = synth.timed_synth(20)
td = "/tmp/sp"; os.makedirs(td, exist_ok=True)
mp = Path(td) / "TIMED.pack"; mp.write_bytes(pack.keras_to_pack(cfg, w))
stem = os.path.join(td, "synth_f32")
if not os.path.exists(stem + ".sparse.rank.npy"):
    bench_legs.make_frame_pack(stem, 30000, gaussian=True); framepack.sparsify(stem)
warnings.simplefilter("ignore")
for k in range(2):
    out = Path(td) / f"o{k}"; out.mkdir(exist_ok=True)
    for f in out.iterdir(): f.unlink()
    t0 = time.perf_counter()
    predict.load_dataset_and_predict([mp], stem + ".framepack", batch_size=500, dataset_map_path=out / "datasetmap.txt", path_to_output=out)
    print("run", k, 30000 / (time.perf_counter() - t0), "frames/s")
PY
cd /tmp
python /tmp/run_sparse.py $ROOT 2>&1 | grep -v Predicting
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o sp -- python /tmp/run_sparse.py $ROOT > $OUT/prof.log 2>&1
F=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); head -20 $F | cut -c1-160
F2=$(find $OUT/prof -name '*memory_copy_stats.csv' | head -1); head -8 $F2 | cut -c1-160
rm -rf $OUT/prof/*.db $OUT/prof/*/*.db
