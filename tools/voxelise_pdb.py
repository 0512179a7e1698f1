#!/usr/bin/env python
"""PDB structure(s) -> frame pack that predict.py accepts wherever it accepts an aposteriori .hdf5:

    python tools/voxelise_pdb.py structure.pdb[.gz] [more.pdb ...] --out data [--boolean] [--all-states]
    python timed-design_amd/predict.py --path_to_dataset data.framepack --path_to_model TIMED.h5 --path_to_output out

Stands where the reference runs `make-frame-dataset ... --voxels-per-side 21 --frame-edge-length 21 -g True -cb True
-ae CNOCBCA` (README.md:83-97).  PARITY UNPINNED against aposteriori (see timed_hip/voxeliser.py)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import voxeliser  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("structures", nargs="+")
    ap.add_argument("--out", required=True, help="stem of the frame pack to write")
    ap.add_argument("--hdf5", action="store_true", help="also write <out>.hdf5 in aposteriori's layout (gzip), readable by h5py and the reference")
    ap.add_argument("--boolean", action="store_true", help="voxels_as_gaussian=False (uint8 frames)")
    ap.add_argument("--all-states", action="store_true", help="voxelise every MODEL as <code>_<k>")
    ap.add_argument("--voxels-per-side", type=int, default=21)
    ap.add_argument("--frame-edge-length", type=float, default=21.0)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    t0 = time.perf_counter()
    Xs, Ls, flat = [], [], []
    for path in a.structures:
        X, labels, rows = voxeliser.voxelise_pdb(path, a.voxels_per_side, a.frame_edge_length, gaussian=not a.boolean,
                                                 all_states=a.all_states, device=a.device)
        Xs.append(X); Ls.append(labels); flat += rows
    X, L = np.concatenate(Xs), np.concatenate(Ls)
    voxeliser.write_frame_pack(a.out, X, L, flat, gaussian=not a.boolean, source=",".join(os.path.basename(p) for p in a.structures))
    if a.hdf5:
        voxeliser.write_hdf5(a.out + ".hdf5", X, L, flat, gaussian=not a.boolean, frame_edge_length=a.frame_edge_length)
    print(f"{len(flat)} residue frames {X.shape[1:]} {X.dtype} from {len(a.structures)} structure(s) -> {a.out}.framepack "
          f"in {time.perf_counter() - t0:.2f} s")
