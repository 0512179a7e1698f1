"""timed_hip.h5write: the HDF5 writer behind the voxeliser's aposteriori-layout output (reference
design_utils/utils.py:238-251).  Files are read back with this repo's h5lite and — where the image's conda interpreter
with REAL h5py exists — with the HDF5 library itself, the arbiter of whether the bytes are valid HDF5."""
import json
import os
import subprocess

import numpy as np
import pytest

from timed_hip import h5lite, h5write

CONDA = "/opt/conda/bin/python3.9"
READER = r'''
import h5py, json, sys, numpy as np
out = {}
def walk(g, prefix):
    for k in g:
        o = g[k]
        p = prefix + "/" + k
        if isinstance(o, h5py.Group):
            out[p] = {"group": len(o), "attrs": {a: (v.tolist() if hasattr(v, "tolist") else v) for a, v in o.attrs.items()}}
            walk(o, p)
        else:
            a = o[()]
            out[p] = {"shape": list(o.shape), "dtype": str(o.dtype), "sum": float(np.asarray(a, dtype=np.float64).sum()),
                      "compression": o.compression, "attrs": {k2: (v.tolist() if hasattr(v, "tolist") else v) for k2, v in o.attrs.items()}}
with h5py.File(sys.argv[1], "r") as f:
    out["/"] = {"attrs": {a: (v.tolist() if hasattr(v, "tolist") else v) for a, v in f.attrs.items()}}
    walk(f, "")
print(json.dumps(out))
'''


def _write(path, n_res=300):
    rng = np.random.default_rng(0)
    data = {}
    with h5write.File(path) as f:
        f.attrs["make_frame_dataset_ver"] = "2.4.0"
        f.attrs["frame_dims"] = np.array([5, 5, 5, 3], dtype=np.int64)
        f.attrs["atom_encoder"] = ["C", "N", "O"]
        f.attrs["encode_cb"] = True
        f.attrs["frame_edge_length"] = 21.0
        f.attrs["voxels_as_gaussian"] = False
        for pdb in ("1abc", "2xyz_0", "9zzz"):
            g = f.create_group(pdb)
            for chain in ("A", "B"):
                c = g.create_group(chain)
                for r in range(1, (n_res if (pdb, chain) == ("1abc", "A") else 7) + 1):
                    kind = r % 4
                    if kind == 0:
                        a = rng.random((5, 5, 5, 3))
                    elif kind == 1:
                        a = rng.random((5, 5, 5, 3)).astype(np.float32)
                    elif kind == 2:
                        a = rng.random((5, 5, 5, 3)) < 0.1
                    else:
                        a = (rng.random((5, 5, 5, 3)) * 255).astype(np.uint8)
                    c.create_dataset(str(r), a, compression="gzip" if r % 3 else None,
                                     attrs={"label": ["MET", "GLY", "TRP"][r % 3], "encoded_residue": np.eye(20)[r % 20]})
                    data[f"/{pdb}/{chain}/{r}"] = a
    return data


def test_h5lite_reads_what_h5write_writes(tmp_path):
    path = tmp_path / "t.hdf5"
    data = _write(path)
    with h5lite.File(path) as f:
        assert f.attrs["make_frame_dataset_ver"] == "2.4.0" and bool(f.attrs["encode_cb"]) is True
        assert bool(f.attrs["voxels_as_gaussian"]) is False and float(f.attrs["frame_edge_length"]) == 21.0
        assert list(f.attrs["atom_encoder"]) == ["C", "N", "O"] and list(f.attrs["frame_dims"]) == [5, 5, 5, 3]
        assert f.keys() == ["1abc", "2xyz_0", "9zzz"] and len(f["1abc"]["A"]) == 300          # 38 symbol-table nodes, 2 B-tree levels
        for key, want in data.items():
            _, pdb, chain, r = key.split("/")
            ds = f[pdb][chain][r]
            got = np.asarray(ds[()])
            assert got.dtype == want.dtype and np.array_equal(got, want), key
            assert ds.attrs["label"] == ["MET", "GLY", "TRP"][int(r) % 3]
            assert np.array_equal(ds.attrs["encoded_residue"], np.eye(20)[int(r) % 20])


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no interpreter with real h5py in this image")
def test_the_hdf5_library_reads_what_h5write_writes(tmp_path):
    path = tmp_path / "t.hdf5"
    data = _write(path)
    r = subprocess.run([CONDA, "-c", READER, str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout)
    assert got["/"]["attrs"] == {"atom_encoder": ["C", "N", "O"], "encode_cb": True, "frame_dims": [5, 5, 5, 3], "frame_edge_length": 21.0,
                                 "make_frame_dataset_ver": "2.4.0", "voxels_as_gaussian": False}
    assert got["/1abc/A"]["group"] == 300 and got["/2xyz_0/B"]["group"] == 7
    for key, want in data.items():
        e = got[key]
        assert e["shape"] == [5, 5, 5, 3] and e["dtype"] == str(want.dtype), (key, e["dtype"])
        assert abs(e["sum"] - float(np.asarray(want, dtype=np.float64).sum())) < 1e-9
        r_ = int(key.rsplit("/", 1)[1])
        assert e["compression"] == ("gzip" if r_ % 3 else None)
        assert e["attrs"]["label"] == ["MET", "GLY", "TRP"][r_ % 3]                     # a str, as aposteriori's files give
        assert e["attrs"]["encoded_residue"] == np.eye(20)[r_ % 20].tolist()


def test_dataset_map_and_load_batch_on_a_written_file(tmp_path):
    """the reference-named readers on a file in aposteriori's layout written by h5write"""
    import warnings
    from design_utils import utils
    path = tmp_path / "d.hdf5"
    rng = np.random.default_rng(1)
    frames = rng.random((9, 5, 5, 5, 3)).astype(np.float32)
    with h5write.File(path) as f:
        f.attrs["frame_dims"] = np.array([5, 5, 5, 3], dtype=np.int64)
        f.attrs["voxels_as_gaussian"] = True
        c = f.create_group("1ubq").create_group("A")
        for i in range(9):
            c.create_dataset(str(10 - i), frames[i].astype(np.float64), compression="gzip",
                             attrs={"label": "MSE" if i == 4 else "ALA", "encoded_residue": np.eye(20)[i]})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flat, pdbs = utils.create_flat_dataset_map(path)
    assert [r[2] for r in flat] == [str(k) for k in range(2, 11)] and flat[4][3] == "MET" and pdbs == {"1ubq"}   # numeric order, MSE -> MET
    X, y = utils.load_batch(path, flat)
    assert X.dtype == np.float64 and np.array_equal(X, frames[::-1].astype(np.float64)) and np.array_equal(y, np.eye(20)[8::-1])
