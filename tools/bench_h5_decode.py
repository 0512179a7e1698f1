#!/usr/bin/env python
"""Time the two ingest paths of a gzip .hdf5 frame dataset, group by group (no prediction):
    python tools/bench_h5_decode.py [n_frames=4096] [group=1024]
host: design_utils.utils.load_batch (native resolver + zlib on host threads); device: load_batch_device (th_h5_decode_device)."""
import json, os, subprocess, sys, tempfile, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
import numpy as np
from design_utils import utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
grp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
td = tempfile.mkdtemp()
h5 = os.path.join(td, "f.hdf5")
r = subprocess.run(["/opt/conda/bin/python3.9", os.path.join(ROOT, "tools", "make_synthetic_hdf5.py"), h5, str(max(1, n // 100)), "100"],
                   capture_output=True, text=True)
assert r.returncode == 0, r.stderr
warnings.simplefilter("ignore")
fmap = np.array(utils.create_flat_dataset_map(h5)[0])
n = len(fmap)
res = {"frames": n, "group": grp, "file_MB": os.path.getsize(h5) / 1e6}
for name, fn in (("device", lambda rows: utils.load_batch_device(h5, rows, device=0)), ("host", lambda rows: utils.load_batch(h5, rows, dtype=np.float32))):
    fn(fmap[:grp])                                  # warm up (library, pools)
    t0 = time.perf_counter()
    for lo in range(0, n, grp):
        out = fn(fmap[lo:lo + grp])
        assert out is not None
        del out
    dt = time.perf_counter() - t0
    res[name + "_fps"] = n / dt
    res[name + "_ms_per_group"] = dt / ((n + grp - 1) // grp) * 1e3
print(json.dumps(res))
