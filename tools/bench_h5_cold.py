#!/usr/bin/env python
"""predict.py on a gzip float64 .hdf5 as the FIRST thing a process does (what every `python predict.py ...` invocation is), then
once more in the same process:   python tools/bench_h5_cold.py frames.hdf5 <n_frames> [batch_size]
Prints one JSON line: the cold call (HIP start-up and code-object load, the model load with its guard, the decoder's scratch, the
first parse of the file's group tables — all inside the timed region), the warm call, and the import time in front of them."""
import json, os, sys, tempfile, time, warnings
t_imp = time.perf_counter()
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
import predict
from timed_hip import pack, synth
t_imp = time.perf_counter() - t_imp
h5, n = sys.argv[1], int(sys.argv[2])
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 500
cfg, w = synth.timed_synth(20)
out = {"import_s": t_imp, "frames": n}
with tempfile.TemporaryDirectory() as td:
    mp = Path(td) / "TIMED.pack"
    mp.write_bytes(pack.keras_to_pack(cfg, w))
    for tag in ("cold", "warm"):
        o = Path(td) / tag
        o.mkdir()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            predict.load_dataset_and_predict([mp], h5, batch_size=bs, dataset_map_path=o / "datasetmap.txt", path_to_output=o)
            dt = time.perf_counter() - t0
        assert sum(1 for _ in open(o / "TIMED.csv")) == n
        out[tag + "_s"] = dt
        out[tag + "_fps"] = n / dt
print(json.dumps(out))
