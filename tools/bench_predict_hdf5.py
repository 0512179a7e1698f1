#!/usr/bin/env python
"""predict.py end to end from an aposteriori-style .hdf5 (per-residue gzip datasets, the reference's input format):
load_batch (h5lite + native th_h5_read_chunked) -> th_predict -> writers.  Needs tools/make_synthetic_hdf5.py's output."""
import json, os, subprocess, sys, tempfile, time, warnings
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import pack, synth
from design_utils import utils
import predict
n_pdb, n_res, bs = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 40), (2, 100), (3, 1000)))
with tempfile.TemporaryDirectory() as td:
    h5 = os.path.join(td, "frames.hdf5")
    subprocess.run(["/opt/conda/bin/python3.9", os.path.join(ROOT, "tools", "make_synthetic_hdf5.py"), h5, str(n_pdb), str(n_res)], check=True)
    cfg, w = synth.timed_synth(20); mp = Path(td) / "TIMED.pack"; mp.write_bytes(pack.keras_to_pack(cfg, w))
    out = Path(td) / "out"; out.mkdir()
    warnings.simplefilter("ignore")
    flat, _ = utils.create_flat_dataset_map(h5)
    t0 = time.perf_counter(); X, y = utils.load_batch(h5, flat[:bs]); t_load = time.perf_counter() - t0
    t0 = time.perf_counter()
    predict.load_dataset_and_predict([mp], h5, batch_size=bs, dataset_map_path=out / "datasetmap.txt", path_to_output=out)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(frames=len(flat), batch_size=bs, host_cores=os.cpu_count(), load_batch_frames_per_s=bs / t_load,
                          predict_py_frames_per_s=len(flat) / dt, file_MB=os.path.getsize(h5) / 1e6)))
