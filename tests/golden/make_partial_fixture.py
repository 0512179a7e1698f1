#!/opt/conda/bin/python3.9
"""Write tests/golden/frames_partial.hdf5 with REAL h5py: two (21,21,21,6) float64 gzip residue datasets with h5py's automatic
chunking, one of them written only in part — its other chunks were never allocated and read as zeros (HDF5's default fill value).
Pins the "missing chunk" path of the readers (host and GPU: the batch is zeroed first when chunks are missing).
Usage:  /opt/conda/bin/python3.9 tests/golden/make_partial_fixture.py"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(3)
full = (rng.random((21, 21, 21, 6)) * (rng.random((21, 21, 21, 6)) < 0.03)).astype(np.float64)
part = np.zeros((21, 21, 21, 6))
part[:6, :11] = (rng.random((6, 11, 21, 6)) * (rng.random((6, 11, 21, 6)) < 0.05))
path = os.path.join(HERE, "frames_partial.hdf5")
with h5py.File(path, "w") as f:
    f.attrs["make_frame_dataset_ver"] = "2.4.0"; f.attrs["frame_dims"] = (21, 21, 21, 6)
    f.attrs["atom_encoder"] = list("CNOQP") + ["CA"]; f.attrs["encode_cb"] = True; f.attrs["atom_filter_fn"] = "keep_sidechain_cb"
    f.attrs["residue_encoder"] = ["ALA"]; f.attrs["frame_edge_length"] = 21.0; f.attrs["voxels_as_gaussian"] = True
    c = f.create_group("1abc").create_group("A")
    d1 = c.create_dataset("1", data=full, dtype=float, compression="gzip")
    d2 = c.create_dataset("2", shape=(21, 21, 21, 6), dtype=float, compression="gzip", chunks=d1.chunks)
    d2[:6, :11] = part[:6, :11]                     # only the chunks under this block are ever allocated
    assert d2.id.get_num_chunks() < d1.id.get_num_chunks(), (d2.id.get_num_chunks(), d1.id.get_num_chunks())
    for k, d in enumerate((d1, d2)):
        d.attrs["label"] = "ALA"
        e = np.zeros(20); e[0] = 1
        d.attrs["encoded_residue"] = e
with h5py.File(path, "r") as f:
    back = np.stack([f["1abc"]["A"]["1"][()], f["1abc"]["A"]["2"][()]])
assert np.array_equal(back, np.stack([full, part]))
np.savez_compressed(os.path.join(HERE, "frames_partial_expected.npz"), frames32=back.astype(np.float32))
print(os.path.getsize(path), "bytes")
