#!/usr/bin/env python
"""predict.py end to end on a gzip float64 .hdf5 written by real h5py (40 000 frames), three times in ONE process: the first run
pays HIP start-up, the scratch allocations of the GPU inflater and the first parse of the file; the others are the warm rate.
    python tools/bench_h5_e2e.py        (TIMED_PIPELINE_TRACE=1 adds the loader / writer waits)"""
import os, sys, time, subprocess, tempfile, warnings
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
import predict
from timed_hip import pack, synth
cfg, w = synth.timed_synth(20)
td = tempfile.mkdtemp()
mp = Path(td) / "TIMED.pack"; mp.write_bytes(pack.keras_to_pack(cfg, w))
h5 = os.path.join(td, "frames.hdf5")
subprocess.run(["/opt/conda/bin/python3.9", os.path.join(ROOT, "tools", "make_synthetic_hdf5.py"), h5, "400", "100"], check=True, capture_output=True)
for k in range(3):
    out = Path(td) / f"out{k}"; out.mkdir()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        predict.load_dataset_and_predict([mp], h5, batch_size=500, dataset_map_path=out / "datasetmap.txt", path_to_output=out)
        dt = time.perf_counter() - t0
    print(f"run {k}: {dt:.3f} s  {40000/dt:.0f} fps", flush=True)
