#!/usr/bin/env python
"""cProfile of predict.load_dataset_and_predict on a synthetic frame pack: where the host time goes."""
import cProfile, os, pstats, sys, tempfile, io
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from timed_hip import pack, synth
import predict
import bench_legs as b
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
with tempfile.TemporaryDirectory() as td:
    stem = os.path.join(td, "synth"); b.make_frame_pack(stem, n, gaussian=(os.environ.get("E2E_BOOL") is None))
    cfg, w = synth.timed_synth(20); mp = Path(td) / "TIMED.pack"; mp.write_bytes(pack.keras_to_pack(cfg, w))
    out = Path(td) / "out"; out.mkdir()
    pr = cProfile.Profile(); pr.enable()
    predict.load_dataset_and_predict([mp], stem + ".framepack", batch_size=bs, dataset_map_path=out / "datasetmap.txt", path_to_output=out)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
