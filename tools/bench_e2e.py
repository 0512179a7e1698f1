#!/usr/bin/env python
"""predict.py end to end at chosen sizes (the `e2e` leg of bench.py, tools/bench_legs.py predict_py_e2e, standalone):
    python tools/bench_e2e.py [n_pack_frames] [n_hdf5_frames] [batch_size]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "timed-design_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_legs
from timed_hip import synth
n_pack, n_h5, bs = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 100000), (2, 10000), (3, 500)))
cfg, w = synth.timed_synth(20)
res = bench_legs.predict_py_e2e(cfg, w, n_pack=n_pack, n_hdf5=n_h5, batch_size=bs)
res["host_cores"] = os.cpu_count()
print(json.dumps(res))
