#!/opt/conda/bin/python3.9
"""Write tests/golden/frames_fletcher.hdf5 with REAL h5py: an aposteriori-layout file whose residue datasets carry the
fletcher32 filter (alone, and on top of shuffle + gzip).  It pins the Fletcher-32 verification of the readers (HDF5 checks the
checksum on every read and fails the read on a mismatch) against checksums that the HDF5 library itself computed.
Usage:  /opt/conda/bin/python3.9 tests/golden/make_fletcher_fixture.py"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
THREE = ["ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU"]
rng = np.random.default_rng(11)
frames = (rng.random((4, 7, 7, 7, 5)) * (rng.random((4, 7, 7, 7, 5)) < 0.2)).astype(np.float32)
path = os.path.join(HERE, "frames_fletcher.hdf5")
with h5py.File(path, "w") as f:
    f.attrs["make_frame_dataset_ver"] = "2.4.0"; f.attrs["frame_dims"] = (7, 7, 7, 5)
    f.attrs["atom_encoder"] = list("CNOQP"); f.attrs["encode_cb"] = True; f.attrs["atom_filter_fn"] = "keep_sidechain_cb"
    f.attrs["residue_encoder"] = THREE; f.attrs["frame_edge_length"] = 7.0; f.attrs["voxels_as_gaussian"] = True
    c = f.create_group("1abc").create_group("A")
    kinds = [dict(fletcher32=True), dict(fletcher32=True, compression="gzip"), dict(fletcher32=True, compression="gzip", shuffle=True),
             dict(fletcher32=True, chunks=(4, 4, 4, 5))]          # the last one: several chunks, odd byte counts inside the dataset edge
    for r, kw in enumerate(kinds):
        d = c.create_dataset(str(r + 1), data=frames[r], **kw)
        d.attrs["label"] = THREE[r]
        e = np.zeros(20); e[r] = 1
        d.attrs["encoded_residue"] = e
np.savez_compressed(os.path.join(HERE, "frames_fletcher_expected.npz"), frames=frames)
print(os.path.getsize(path), "bytes")
