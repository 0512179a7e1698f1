#!/usr/bin/env python
"""cProfile of predict.py end to end from an aposteriori-style .hdf5 (per-residue gzip datasets): a first (cold) call, then the
profiled warm call — what a call costs besides the per-frame work.   python tools/profile_predict_hdf5.py [n_pdb] [n_res] [batch]"""
import cProfile, io, os, pstats, subprocess, sys, tempfile, time, warnings
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import pack, synth
import predict
n_pdb, n_res, bs = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 40), (2, 100), (3, 500)))
with tempfile.TemporaryDirectory() as td:
    h5 = os.path.join(td, "frames.hdf5")
    subprocess.run(["/opt/conda/bin/python3.9", os.path.join(ROOT, "tools", "make_synthetic_hdf5.py"), h5, str(n_pdb), str(n_res)], check=True)
    cfg, w = synth.timed_synth(20); mp = Path(td) / "TIMED.pack"; mp.write_bytes(pack.keras_to_pack(cfg, w))
    warnings.simplefilter("ignore")
    for k in range(3):
        out = Path(td) / f"out{k}"; out.mkdir()
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        if k == 2:
            pr.enable()
        predict.load_dataset_and_predict([mp], h5, batch_size=bs, dataset_map_path=out / "datasetmap.txt", path_to_output=out)
        if k == 2:
            pr.disable()
        print(f"call {k}: {n_pdb * n_res} frames in {time.perf_counter() - t0:.3f} s", flush=True)
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
