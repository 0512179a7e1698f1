#!/opt/conda/bin/python3.9
"""Write tests/golden/frames_midchunk.hdf5 + frames_midchunk_expected.npz with REAL h5py (only /opt/conda's python3.9 has it in
the build container): residue datasets (21, 21, 21, 6) float64 stored in (7, 11, 11, 6) chunks = 40 656 bytes each — longer than
the 32 KB DEFLATE window but short enough for the GPU decoder's whole-stream LDS window (<= 60 KB), the geometry where
k_lz_resolve once flushed mid-stream into a 16-byte buffer (inflate.hip).  Residues 3, 4 are gzip only, 5, 6 are
shuffle + gzip; residue 4 and 6 carry dense noise in one chunk so that literals dominate there.
Usage:  /opt/conda/bin/python3.9 tests/golden/make_midchunk_fixture.py"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
THREE = ["ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU", "MET", "ASN", "PRO", "GLN", "ARG", "SER", "THR", "VAL",
         "TRP", "TYR"]
rng = np.random.default_rng(77)
n = 4
path = os.path.join(HERE, "frames_midchunk.hdf5")
frames = (rng.random((n, 21, 21, 21, 6)) * (rng.random((n, 21, 21, 21, 6)) < 0.02)).astype(np.float64)
frames[1, 7:14, 0:11, 0:11, :] = np.round(rng.random((7, 11, 11, 6)), 2)        # one whole chunk of 2-decimal noise
frames[3, 14:21, 11:21, 11:21, :] = np.round(rng.random((7, 10, 10, 6)), 1)     # an edge chunk (partly outside the dataset)
frames[2, 20, 20, 20, 5] = -1.5
with h5py.File(path, "w") as f:
    f.attrs["make_frame_dataset_ver"] = "2.4.0"; f.attrs["frame_dims"] = (21, 21, 21, 6)
    f.attrs["atom_encoder"] = list("CNOQP") + ["CA"]; f.attrs["encode_cb"] = True; f.attrs["atom_filter_fn"] = "keep_sidechain_cb"
    f.attrs["residue_encoder"] = THREE; f.attrs["frame_edge_length"] = 21.0; f.attrs["voxels_as_gaussian"] = True
    c = f.create_group("1abc").create_group("A")
    for r in range(n):
        d = c.create_dataset(str(r + 3), data=frames[r], dtype=float, compression="gzip", chunks=(7, 11, 11, 6), shuffle=(r >= 2))
        assert d.chunks == (7, 11, 11, 6) and np.prod(d.chunks) * 8 == 40656
        d.attrs["label"] = THREE[(3 * r) % 20]
        e = np.zeros(20); e[(3 * r) % 20] = 1
        d.attrs["encoded_residue"] = e
with h5py.File(path, "r") as f:
    back = np.stack([f["1abc"]["A"][str(r + 3)][()] for r in range(n)])
assert np.array_equal(back, frames)
np.savez_compressed(os.path.join(HERE, "frames_midchunk_expected.npz"), frames32=back.astype(np.float32))
print(os.path.getsize(path), "bytes")
