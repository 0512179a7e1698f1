"""th_h5_read_chunked / h5lite.read_many_direct (host code, no GPU): the native bulk reader of chunked gzip frame
datasets gives exactly what the per-dataset pure-Python path (pinned to real-h5py fixtures in test_host_utils.py)
gives — SURVEY.md §8 row f-1, reference load_batch design_utils/utils.py:514-529."""
import os
import warnings

import numpy as np
import pytest

from design_utils import utils
from timed_hip import h5lite

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["frames_tiny.hdf5", "frames_tiny_bool.hdf5"])
def test_bulk_reader_equals_generic_path(name):
    path = os.path.join(G, name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flat, _ = utils.create_flat_dataset_map(path)
    with h5lite.File(path) as f:
        dss = [f[p][c][r] for p, c, r, _ in flat]
        want = [np.asarray(d[()]) for d in dss]
        dests = [np.empty(w.shape, w.dtype) for w in want]
        done = h5lite.read_many_direct(dss, dests)
        for ok, d, w, ds in zip(done, dests, want, dss):
            if ds.chunked_geometry() is not None:
                assert ok
            if ok:
                assert np.array_equal(d, w)
        # the pure-Python direct path agrees too
        d2 = np.empty(want[0].shape, want[0].dtype)
        if dss[0].read_direct(d2):
            assert np.array_equal(d2, want[0])
    X, y = utils.load_batch(path, flat)             # end to end through load_batch
    assert np.array_equal(X, np.stack(want)) and y.shape == (len(flat), 20)


def test_bulk_reader_rejects_what_it_cannot_place():
    path = os.path.join(G, "frames_tiny.hdf5")
    with h5lite.File(path) as f:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            flat, _ = utils.create_flat_dataset_map(path)
        ds = f[flat[0][0]][flat[0][1]][flat[0][2]]
        w = np.asarray(ds[()])
        wrong_shape = np.empty(w.shape[:-1] + (w.shape[-1] + 1,), w.dtype)
        assert h5lite.read_many_direct([ds], [wrong_shape]) == [False]
        wrong_size = np.empty(w.shape, np.float32 if w.dtype.itemsize == 8 else np.float64)
        assert h5lite.read_many_direct([ds], [wrong_size]) == [False]


def test_native_reader_reports_corrupt_btree_address():
    import ctypes as C
    from timed_hip import _lib
    lib = _lib.load()
    buf = np.zeros(4096, np.uint8)
    dest = np.zeros((4, 4), np.float64)
    rc = lib.th_h5_read_chunked(buf.ctypes.data_as(C.c_void_p), buf.size, 0, 1, (C.c_int64 * 1)(128),
                                (C.c_void_p * 1)(dest.ctypes.data), 2, (C.c_int64 * 2)(4, 4), (C.c_int64 * 2)(2, 2), 8, 1,
                                (C.c_int * 1)(1), 1)
    assert rc != 0 and b"B-tree" in lib.th_last_error()
    rc = lib.th_h5_read_chunked(buf.ctypes.data_as(C.c_void_p), buf.size, 0, 1, (C.c_int64 * 1)(128),
                                (C.c_void_p * 1)(dest.ctypes.data), 2, (C.c_int64 * 2)(4, 4), (C.c_int64 * 2)(2, 2), 8, 1,
                                (C.c_int * 1)(32000), 1)
    assert rc != 0     # unknown filter id
