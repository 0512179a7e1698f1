#!/usr/bin/env python
"""End-to-end predict.py throughput on a synthetic frame pack (host path: th_predict with host
buffers, the reference's batch loop and text writers).  Complements bench.py, which measures the
device-resident kernel path."""
import argparse, json, os, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import engine, pack, synth  # noqa: E402
import predict  # noqa: E402


def make_pack(stem, n, gaussian):
    side, c = 21, 6
    base = synth.synthetic_frames(256, seed=1, gaussian=gaussian)
    frames = np.lib.format.open_memmap(stem + ".frames.npy", mode="w+", dtype=base.dtype, shape=(n, side, side, side, c))
    for lo in range(0, n, 256):
        frames[lo:lo + 256] = base[: min(256, n - lo)]
    frames.flush(); del frames
    labels = np.zeros((n, 20), np.uint8); labels[np.arange(n), np.arange(n) % 20] = 1
    np.save(stem + ".labels.npy", labels)
    three = ["ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU", "MET", "ASN", "PRO", "GLN", "ARG", "SER", "THR", "VAL", "TRP", "TYR"]
    rows = [(f"p{i // 300:04d}", "A", str(i % 300 + 1), three[i % 20]) for i in range(n)]
    np.savetxt(stem + ".map.txt", np.array(rows), delimiter=",", fmt="%s")
    json.dump(dict(frame_dims=[side, side, side, c], voxels_as_gaussian=gaussian, n_frames=n, source="synthetic", make_frame_dataset_ver=""),
              open(stem + ".meta.json", "w"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20000)
    ap.add_argument("--batch_size", type=int, default=500)
    ap.add_argument("--bool", action="store_true")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        stem = os.path.join(td, "synth")
        make_pack(stem, a.frames, not a.bool)
        cfg, w = synth.timed_synth(20)
        mp = Path(td) / "TIMED.pack"
        mp.write_bytes(pack.keras_to_pack(cfg, w))
        out = Path(td) / "out"; out.mkdir()
        t0 = time.perf_counter()
        predict.load_dataset_and_predict([mp], stem + ".framepack", batch_size=a.batch_size, dataset_map_path=out / "datasetmap.txt",
                                         path_to_output=out)
        dt = time.perf_counter() - t0
        # forward-only through the host entry point for comparison
        m = engine.HipFrameModel.load(mp)
        X = np.load(stem + ".frames.npy", mmap_mode="r")
        t1 = time.perf_counter()
        for lo in range(0, a.frames, a.batch_size):
            m.predict(X[lo:lo + a.batch_size])
        dt2 = time.perf_counter() - t1
        print(json.dumps(dict(frames=a.frames, dtype=str(X.dtype), batch_size=a.batch_size, predict_py_frames_per_s=a.frames / dt,
                              th_predict_host_frames_per_s=a.frames / dt2, predict_py_s=dt, forward_only_s=dt2)))
