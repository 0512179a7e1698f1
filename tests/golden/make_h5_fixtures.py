#!/opt/conda/bin/python3.9
"""Write the small HDF5 fixtures under tests/golden/ with REAL h5py (only /opt/conda's python3.9 has
it in the build container):

  keras_tiny.h5        a Keras-legacy-format model file (model_config attr + model_weights groups)
                       holding a tiny TIMED-style net from timed_hip.synth
  frames_tiny.hdf5     an aposteriori-style frame dataset (layout documented at reference
  frames_tiny_bool.hdf5  design_utils/utils.py:238-251): pdb/chain/residue datasets, gzip, attrs

They pin timed_hip/h5lite.py (pure-Python reader) against files produced by the real library.
Usage:  /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py
"""
import json
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import synth  # noqa: E402

THREE = ["ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU", "MET", "ASN", "PRO", "GLN", "ARG", "SER",
         "THR", "VAL", "TRP", "TYR"]


def write_keras(path):
    cfg, weights = synth.timed_synth(20, widths=(8, 16), side=7, in_channels=5, seed=11, bias_std=0.1)
    with h5py.File(path, "w") as f:
        f.attrs["keras_version"] = b"2.13.1"
        f.attrs["backend"] = b"tensorflow"
        f.attrs["model_config"] = json.dumps(cfg).encode("utf8")
        f.attrs["training_config"] = json.dumps({"loss": "categorical_crossentropy", "metrics": ["top_3_cat_acc"]}).encode()
        g = f.create_group("model_weights")
        names = [l["name"] for l in cfg["config"]["layers"]]
        g.attrs["layer_names"] = [n.encode() for n in names]
        g.attrs["backend"] = b"tensorflow"
        suffix = {"Conv3D": ["kernel:0", "bias:0"], "BatchNormalization": ["gamma:0", "beta:0", "moving_mean:0", "moving_variance:0"]}
        for l in cfg["config"]["layers"]:
            lg = g.create_group(l["name"])
            ws = weights.get(l["name"], [])
            wn = [f"{l['name']}/{s}" for s in suffix.get(l["class_name"], [])[:len(ws)]]
            lg.attrs["weight_names"] = [w.encode() for w in wn]
            for w, a in zip(wn, ws):
                lg.create_dataset(w, data=a)  # nested group "<layer>/<weight>" like Keras


def write_frames(path, gaussian):
    rng = np.random.default_rng(5 if gaussian else 6)
    dims = (7, 7, 7, 5)
    with h5py.File(path, "w") as f:
        f.attrs["make_frame_dataset_ver"] = "2.4.0"
        f.attrs["frame_dims"] = dims
        f.attrs["atom_encoder"] = ["C", "N", "O", "CA", "CB"]
        f.attrs["encode_cb"] = True
        f.attrs["atom_filter_fn"] = "keep_sidechain_cb_atoms"
        f.attrs["residue_encoder"] = list("ACDEFGHIKLMNPQRSTVWY")
        f.attrs["frame_edge_length"] = 7.0
        f.attrs["voxels_as_gaussian"] = gaussian
        for pdb, chains in (("1ubq", {"A": 12}), ("2xyz_0", {"A": 3, "B": 11})):
            pg = f.create_group(pdb)
            for chain, n in chains.items():
                cg = pg.create_group(chain)
                # residue ids deliberately include 2-digit numbers: string order != numeric order
                for rid in range(2, 2 + n):
                    if gaussian:
                        frame = rng.random(dims, dtype=np.float32) * (rng.random(dims) < 0.1)
                        frame = frame.astype(np.float32)
                    else:
                        frame = rng.random(dims) < 0.05
                    ds = cg.create_dataset(str(rid), data=frame, compression="gzip" if rid % 2 else None,
                                           shuffle=bool(rid % 3 == 0) if rid % 2 else False)
                    label = THREE[int(rng.integers(0, 20))]
                    if pdb == "2xyz_0" and rid == 4:
                        label = "MSE"  # uncommon residue -> MET
                        enc = np.eye(20)[THREE.index("MET")]
                    else:
                        enc = np.eye(20)[THREE.index(label)]
                    ds.attrs["label"] = label
                    ds.attrs["encoded_residue"] = enc


def main():
    write_keras(os.path.join(HERE, "keras_tiny.h5"))
    write_frames(os.path.join(HERE, "frames_tiny.hdf5"), True)
    write_frames(os.path.join(HERE, "frames_tiny_bool.hdf5"), False)
    for n in ("keras_tiny.h5", "frames_tiny.hdf5", "frames_tiny_bool.hdf5"):
        print(n, os.path.getsize(os.path.join(HERE, n)), "bytes")
    # also dump what h5py itself reads, for the reader tests (numpy arrays only)
    out = {}
    with h5py.File(os.path.join(HERE, "frames_tiny.hdf5"), "r") as f:
        out["g_1ubq_A_5"] = f["1ubq"]["A"]["5"][()]
        out["g_2xyz_B_12"] = f["2xyz_0"]["B"]["12"][()]
        out["g_enc"] = f["2xyz_0"]["A"]["4"].attrs["encoded_residue"]
    with h5py.File(os.path.join(HERE, "frames_tiny_bool.hdf5"), "r") as f:
        out["b_1ubq_A_5"] = f["1ubq"]["A"]["5"][()]
        out["b_1ubq_A_13"] = f["1ubq"]["A"]["13"][()]
    np.savez_compressed(os.path.join(HERE, "h5_expected.npz"), **out)


def write_fill_values(path, expect_path):
    """datasets whose never-written elements must read as a user-defined NON-ZERO fill value: a chunked
    dataset with only one chunk written, a never-written chunked one, a never-written contiguous one, and a zero-fill
    control.  Expectations are what h5py itself reads back."""
    with h5py.File(path, "w") as f:
        d = f.create_dataset("partial", shape=(6, 8), dtype="f8", chunks=(3, 4), fillvalue=2.5, compression="gzip")
        d[0:3, 0:4] = np.arange(12.0).reshape(3, 4)
        f.create_dataset("never_chunked", shape=(4, 4), dtype="f4", chunks=(2, 2), fillvalue=-1.25)
        f.create_dataset("never_contiguous", shape=(5,), dtype="i4", fillvalue=7)
        z = f.create_dataset("zero_fill", shape=(6, 8), dtype="f8", chunks=(3, 4), compression="gzip")
        z[3:6, 4:8] = 1.0
    with h5py.File(path, "r") as f:
        np.savez(expect_path, **{k: f[k][()] for k in f})


if __name__ == "__main__":
    if "--fill-only" not in sys.argv:       # the other fixtures are only rewritten on request: their bytes are pinned in tests
        main()
    write_fill_values(os.path.join(HERE, "fillvalue_tiny.hdf5"), os.path.join(HERE, "fillvalue_expected.npz"))
    print("wrote fillvalue_tiny.hdf5")
