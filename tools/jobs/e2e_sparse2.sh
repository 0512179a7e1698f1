# predict.py from a sparse pack: cold and warm calls, two group sizes:  gpurun -- 'bash tools/jobs/e2e_sparse2.sh'
mkdir -p gpurun_out/e2e
TIMED_PIPELINE_TRACE=1 timeout 600 python - <<'PY' 2>&1 | grep -v Predicting | tee gpurun_out/e2e/sparse_warm.txt
import os, sys, time, warnings
from pathlib import Path
sys.path.insert(0, "tools"); sys.path.insert(0, "timed-design_amd")
import bench_legs, predict
from timed_hip import synth, pack, framepack
cfg, w = synth.timed_synth(20)
td = "/tmp/sp"; os.makedirs(td, exist_ok=True)
mp = Path(td) / "TIMED.pack"; mp.write_bytes(pack.keras_to_pack(cfg, w))
stem = os.path.join(td, "synth_f32")
N = 100000
bench_legs.make_frame_pack(stem, N, gaussian=True); framepack.sparsify(stem)
warnings.simplefilter("ignore")
for fpc in (None, None, 4096, 4096):
    out = Path(td) / "o"; out.mkdir(exist_ok=True)
    for f in out.iterdir(): f.unlink()
    t0 = time.perf_counter()
    kw = {"frames_per_call": fpc} if fpc else {}
    predict.load_dataset_and_predict([mp], stem + ".framepack", batch_size=500, dataset_map_path=out / "datasetmap.txt", path_to_output=out, **kw)
    print("== frames_per_call", fpc, N / (time.perf_counter() - t0), "frames/s")
PY
