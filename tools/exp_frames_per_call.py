"""predict.py on a uint8 frame pack at several --frames_per_call (frames handed to the GPU per th_predict_async call): the default
for host-resident datasets is 1024 (the first piece's copy cannot overlap anything); larger calls run the kernels on
larger chunks.    python tools/exp_frames_per_call.py [frames] [rotamer: 0|1]"""
import os, sys, time, tempfile, warnings
from pathlib import Path
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "timed-design_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from timed_hip import synth, pack
import bench_legs as b
import predict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rot = len(sys.argv) > 2 and sys.argv[2] == "1"
cfg, w = synth.TOPOLOGIES["timed_rotamer" if rot else "timed"]()
with tempfile.TemporaryDirectory() as td:
    mp = Path(td) / "M.pack"
    mp.write_bytes(pack.keras_to_pack(cfg, w))
    stem = os.path.join(td, "u8")
    b.make_frame_pack(stem, n, gaussian=False)
    for rep in range(2):
        for fpc in (1024, 2048, 4096):
            out = Path(td) / f"out_{fpc}_{rep}"
            out.mkdir()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                t0 = time.perf_counter()
                predict.load_dataset_and_predict([mp], stem + ".framepack", batch_size=500, dataset_map_path=out / "datasetmap.txt",
                                                 predict_rotamers=rot, path_to_output=out, frames_per_call=fpc)
                dt = time.perf_counter() - t0
            print(f"rep {rep} frames_per_call {fpc}: {dt:.3f} s = {n / dt / 1e3:.1f} k frames/s", flush=True)
            for f in out.iterdir():
                f.unlink()
