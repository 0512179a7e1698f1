#!/opt/conda/bin/python3.9
"""Write tests/golden/frames_chunked.hdf5 + frames_chunked_expected.npz with REAL h5py (only /opt/conda's python3.9 has it in
the build container): an aposteriori-style dataset whose residue datasets are what h5py writes for the reference's real data —
(21, 21, 21, 6) float64, compression="gzip" with h5py's AUTOMATIC chunking ((6, 11, 11, 3): 32 chunks per frame, the ones at the
upper edges only partly inside the dataset) — plus a boolean chain.  It pins the multi-chunk path of the readers (host:
th_h5_read_chunked_as; device: th_h5_decode_device, whose LZ77 kernel places every chunk straight from LDS) against h5py's own
read of the same file.
Usage:  /opt/conda/bin/python3.9 tests/golden/make_chunked_fixture.py"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
THREE = ["ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU", "MET", "ASN", "PRO", "GLN", "ARG", "SER", "THR", "VAL",
         "TRP", "TYR"]
rng = np.random.default_rng(2024)
n = 5
path = os.path.join(HERE, "frames_chunked.hdf5")
frames = (rng.random((n, 21, 21, 21, 6)) * (rng.random((n, 21, 21, 21, 6)) < 0.03)).astype(np.float64)
frames[1, :, :, 20, :] = 7.25            # a plane on the upper edge of the last-but-one dimension: lives in partly-outside chunks
frames[2, 20, 20, 20, 5] = -1.5          # the very last voxel
with h5py.File(path, "w") as f:
    f.attrs["make_frame_dataset_ver"] = "2.4.0"; f.attrs["frame_dims"] = (21, 21, 21, 6)
    f.attrs["atom_encoder"] = list("CNOQP") + ["CA"]; f.attrs["encode_cb"] = True; f.attrs["atom_filter_fn"] = "keep_sidechain_cb"
    f.attrs["residue_encoder"] = THREE; f.attrs["frame_edge_length"] = 21.0; f.attrs["voxels_as_gaussian"] = True
    c = f.create_group("1abc").create_group("A")
    for r in range(n):
        d = c.create_dataset(str(r + 3), data=frames[r], dtype=float, compression="gzip")
        assert d.chunks == (6, 11, 11, 3), d.chunks
        d.attrs["label"] = THREE[(3 * r) % 20]
        e = np.zeros(20); e[(3 * r) % 20] = 1
        d.attrs["encoded_residue"] = e
with h5py.File(path, "r") as f:
    back = np.stack([f["1abc"]["A"][str(r + 3)][()] for r in range(n)])
assert np.array_equal(back, frames)
np.savez_compressed(os.path.join(HERE, "frames_chunked_expected.npz"), frames32=back.astype(np.float32),
                    labels=np.array([THREE[(3 * r) % 20] for r in range(n)]), residues=np.array([str(r + 3) for r in range(n)]))
print(os.path.getsize(path), "bytes")
