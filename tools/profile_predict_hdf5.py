#!/usr/bin/env python
"""cProfile of predict.py end to end from an aposteriori-style .hdf5 (per-residue gzip datasets)."""
import cProfile, io, os, pstats, subprocess, sys, tempfile, warnings
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import pack, synth
import predict
n_pdb, n_res, bs = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 40), (2, 100), (3, 1000)))
with tempfile.TemporaryDirectory() as td:
    h5 = os.path.join(td, "frames.hdf5")
    subprocess.run(["/opt/conda/bin/python3.9", os.path.join(ROOT, "tools", "make_synthetic_hdf5.py"), h5, str(n_pdb), str(n_res)], check=True)
    cfg, w = synth.timed_synth(20); mp = Path(td) / "TIMED.pack"; mp.write_bytes(pack.keras_to_pack(cfg, w))
    out = Path(td) / "out"; out.mkdir()
    warnings.simplefilter("ignore")
    pr = cProfile.Profile(); pr.enable()
    predict.load_dataset_and_predict([mp], h5, batch_size=bs, dataset_map_path=out / "datasetmap.txt", path_to_output=out)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(24); print(s.getvalue()[:5200])
