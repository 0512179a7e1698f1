#!/usr/bin/env python
"""Convert an aposteriori HDF5 frame dataset into a frame pack (timed_hip/framepack.py):

    python tools/pack_frames.py data.hdf5 data            # writes data.frames.npy, data.labels.npy, data.map.txt, data.meta.json
    python timed-design_amd/predict.py --path_to_dataset data.framepack --path_to_model TIMED.h5 ...
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import framepack  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("hdf5"); ap.add_argument("out_stem")
    a = ap.parse_args()
    t0 = time.time()
    fp = framepack.pack_dataset(a.hdf5, a.out_stem, progress_every=10000)
    print(f"{len(fp)} frames {fp.frames.dtype} {fp.frames.shape[1:]} -> {a.out_stem}.frames.npy in {time.time() - t0:.1f}s")
