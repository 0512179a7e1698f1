"""Sparse transport of float32 frames (row f-1; timed_hip/framepack.py, engine.SparseFrames, csrc/sparse_frames.hip): the host side
without a GPU — bit-exact round trips, the THSPF001 blob layout of include/timed_hip.h, the pack files — and, on the GPU, frames
that travel sparse give the SAME BITS as frames that travel dense (reference seam: design_utils/utils.py:487-530 -> predict.py:142)."""
import json
import os
import struct

import numpy as np
import pytest

from timed_hip import engine, framepack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _gaussianish(n, shape, seed, fill=0.08):
    rng = np.random.default_rng(seed)
    x = rng.random((n, *shape), dtype=np.float32) * (rng.random((n, *shape)) < fill)
    return x.astype(np.float32)


def _nasty(shape, seed):
    """frames with every value class the transport must not touch: -0.0, NaN with a payload, infinities, denormals, an empty
    frame, a full frame"""
    x = _gaussianish(6, shape, seed)
    flat = x.reshape(6, -1)
    flat[0, :7] = np.array([-0.0, np.nan, np.inf, -np.inf, 1e-42, -1e-45, 3.0], np.float32)
    flat[0, 7:8].view(np.uint32)[:] = 0x7fc12345                     # a NaN payload
    flat[1] = 0.0                                                     # nothing stored
    flat[2] = np.random.default_rng(seed + 1).standard_normal(flat.shape[1]).astype(np.float32)    # everything stored
    flat[3, -1] = 5.0                                                 # the frame's last element
    flat[4, 0] = -0.0
    return x


@pytest.mark.parametrize("shape", [(21, 21, 21, 6), (21, 21, 21, 5), (7, 7, 7, 5), (3, 3, 3, 1)])
def test_round_trip_is_bit_exact_and_the_blob_matches_the_header(shape):
    x = _nasty(shape, 3)
    sf = engine.SparseFrames.from_dense(x)
    assert sf.shape == x.shape and sf.dense().tobytes() == x.tobytes()
    E = int(np.prod(shape))
    W = sf.bits.shape[1]
    assert W % 4 == 0 and W * 32 >= E
    stored = np.count_nonzero(x.reshape(6, -1).view(np.uint32), axis=1)
    assert np.array_equal(np.diff(sf.rank.astype(np.int64)), stored) and stored[1] == 0 and stored[2] == E
    blob = sf.blob()
    assert blob.ctypes.data % 16 == 0 and bytes(blob[:8]) == b"THSPF001"
    n, e, w, esz = struct.unpack("<4I", bytes(blob[8:24]))
    nv = struct.unpack("<Q", bytes(blob[24:32]))[0]
    assert (n, e, w, esz, nv) == (6, E, W, 4, int(stored.sum()))
    o_rank, o_bits, o_val, total = engine.sparse_blob_layout(6, W, nv)
    assert blob.nbytes == total and o_bits % 16 == 0
    assert np.array_equal(blob[o_rank:o_rank + 7 * 8].view(np.uint64), sf.rank)
    assert np.array_equal(blob[o_bits:o_val].view(np.uint32).reshape(6, W), sf.bits)
    assert blob[o_val:].view(np.float32).tobytes() == x.reshape(6, -1)[x.reshape(6, -1).view(np.uint32) != 0].tobytes()
    # a slice of a batch is a batch: ranks keep their base
    part = engine.SparseFrames(sf.bits[2:5], sf.rank[2:6], sf.values[int(sf.rank[2]):int(sf.rank[5])], shape)
    assert part.dense().tobytes() == x[2:5].tobytes()


def test_pack_files_sparsify_slices_and_a_pack_without_dense_frames(tmp_path):
    shape = (7, 7, 7, 5)
    x = _nasty(shape, 9)
    stem = str(tmp_path / "p")
    np.save(stem + ".frames.npy", x)
    np.save(stem + ".labels.npy", np.eye(20, dtype=np.uint8)[np.arange(6) % 20])
    np.savetxt(stem + ".map.txt", np.array([["1abc", "A", str(i), "ALA"] for i in range(6)]), delimiter=",", fmt="%s")
    json.dump(dict(frame_dims=list(shape), voxels_as_gaussian=True, n_frames=6, source="x", make_frame_dataset_ver=""), open(stem + ".meta.json", "w"))
    dense_b, sparse_b = framepack.sparsify(stem)
    assert dense_b == x.nbytes and sparse_b < dense_b
    fp = framepack.FramePack(stem)
    assert fp.sparse is not None and len(fp) == 6
    assert fp.sparse_batch(1, 5).dense().tobytes() == x[1:5].tobytes()
    rows = fp.flat_map[2:5]
    assert fp.contiguous_rows(rows) == (2, 5) and fp.contiguous_rows(fp.flat_map[[0, 2]]) is None
    for s in framepack.SPARSE_SUFFIXES:                       # every file of the pack names the pack
        assert framepack.pack_stem(stem + s) == stem
    os.remove(stem + ".frames.npy")                           # the sparse files stand in for the dense frames
    assert framepack.is_pack(stem)
    fp2 = framepack.FramePack(stem)
    X, y = fp2.load_batch(fp2.flat_map[1:4])
    assert X.tobytes() == x[1:4].tobytes() and y.shape == (3, 20)
    X2, _ = fp2.load_batch(fp2.flat_map[[4, 0]])
    assert X2.tobytes() == x[[4, 0]].tobytes()
    with pytest.raises(ValueError):
        np.save(stem + ".frames.npy", x.astype(np.uint8))
        framepack.sparsify(stem)


@pytest.mark.gpu
@pytest.mark.parametrize("cin", [6, 5])
def test_frames_that_travel_sparse_give_the_same_bits(gpu, cin):
    """th_predict_sparse_async == th_predict_async on the same frames, bit for bit: several pieces per call (chunk 3), frames whose
    first element is not 16-byte aligned in the dense ring (E = 46 305 with 5 channels), empty and full frames, NaN / -0.0 / denormals"""
    from timed_hip import synth
    cfg, weights = synth.timed_synth(20, widths=(8, 16), in_channels=cin, seed=4, bias_std=0.1)
    m = engine.HipFrameModel.from_keras(cfg, weights, device=gpu)
    x = np.concatenate([_nasty((21, 21, 21, cin), 5)[1:], _gaussianish(11, (21, 21, 21, cin), 6)])     # (the NaN frame apart, below)
    want = m.predict(x)
    sf = engine.SparseFrames.from_dense(x)
    assert sf.blob_bytes < x.nbytes / 3
    got = m.predict(sf)
    assert got.tobytes() == want.tobytes()
    m.set_chunk(3)
    assert m.predict(sf).tobytes() == want.tobytes()
    a, b = m.predict_async(sf), m.predict_async(engine.SparseFrames.from_dense(x[::-1]))                 # two tickets in flight
    assert b.result().tobytes() == want[::-1].tobytes() and a.result().tobytes() == want.tobytes()
    nan = _nasty((21, 21, 21, cin), 5)[:1]
    assert m.predict(engine.SparseFrames.from_dense(nan), logits=True).tobytes() == m.predict(nan, logits=True).tobytes()
    with pytest.raises(ValueError):
        m.predict(engine.SparseFrames.from_dense(_gaussianish(2, (7, 7, 7, cin), 1)))
    m.close()


@pytest.mark.gpu
def test_bad_sparse_blobs_are_refused_with_error_codes(gpu):
    import ctypes as C
    from timed_hip import _lib, synth
    cfg, weights = synth.timed_synth(20, widths=(8, 16), seed=4)
    m = engine.HipFrameModel.from_keras(cfg, weights, device=gpu)
    lib = _lib.load()
    sf = engine.SparseFrames.from_dense(_gaussianish(3, (21, 21, 21, 6), 2))
    good = sf.blob().copy()
    out = np.empty((3, 20), np.float32)
    t = C.c_int(-1)

    def call(buf, nbytes=None):
        buf = np.require(buf, requirements="A")
        return lib.th_predict_sparse_async(m._h, buf.ctypes.data, buf.nbytes if nbytes is None else nbytes, out.ctypes.data, 0, C.byref(t))

    bad = good.copy(); bad[:8] = 0
    assert call(bad) == _lib.TH_EINVAL                                   # magic
    assert call(good, good.nbytes - 8) == _lib.TH_EINVAL                 # truncated
    bad = good.copy(); bad[12:16].view(np.uint32)[0] -= 1
    assert call(bad) == _lib.TH_EINVAL                                   # another frame size than the model's
    bad = good.copy(); bad[32 + 8:32 + 16].view(np.uint64)[0] = 10 ** 9
    assert call(bad) == _lib.TH_EINVAL                                   # an impossible rank table
    shifted = np.empty(good.nbytes + 32, np.uint8)
    off = (-shifted.ctypes.data) % 16 + 8                                # 8 bytes off a 16-byte boundary
    shifted[off:off + good.nbytes] = good
    assert lib.th_predict_sparse_async(m._h, shifted.ctypes.data + off, good.nbytes, out.ctypes.data, 0, C.byref(t)) == _lib.TH_EINVAL
    assert call(good) == _lib.TH_OK
    _lib.check(lib.th_predict_wait(m._h, t.value))
    assert out.tobytes() == m.predict(sf.dense()).tobytes()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("drop_dense", [False, True])
def test_predict_py_from_a_sparse_pack_writes_the_same_bytes(gpu, tmp_path, monkeypatch, drop_dense):
    """predict.py takes the sparse transport by itself when the pack has the files (ragged last group, groups of one reference batch
    and groups of several); every output file equals the dense-pack run's and the .hdf5 run's; TIMED_SPARSE=0 keeps the dense rows;
    the pipeline really submitted SparseFrames batches"""
    import warnings
    from pathlib import Path
    import predict
    model_path = Path(os.path.join(G, "keras_tiny.h5"))
    src = os.path.join(G, "frames_tiny.hdf5")
    seen = []
    real = engine.HipFrameModel._predict_async_sparse

    def spy(self, X, d_out, logits=False):
        seen.append(len(X))
        return real(self, X, d_out, logits=logits)
    monkeypatch.setattr(engine.HipFrameModel, "_predict_async_sparse", spy)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fp = framepack.pack_dataset(src, tmp_path / "tiny", sparse=True)
        assert fp.sparse is not None and fp.frames.dtype == np.float32
        if drop_dense:
            del fp
            from design_utils import utils as du
            du._PACKS.clear()
            os.remove(tmp_path / "tiny.frames.npy")
        outs = {}
        for tag, data, env, fpc in (("h5", src, None, None), ("sparse", str(tmp_path / "tiny.framepack"), None, None),
                                    ("sparse_small", str(tmp_path / "tiny.framepack"), None, 9), ("dense", str(tmp_path / "tiny.framepack"), "0", None)):
            if tag == "dense" and drop_dense:
                continue
            d = tmp_path / tag
            d.mkdir()
            if env is not None:
                monkeypatch.setenv("TIMED_SPARSE", env)
            n0 = len(seen)
            predict.load_dataset_and_predict([model_path], data, batch_size=9, dataset_map_path=d / "datasetmap.txt", path_to_output=d,
                                             **({"frames_per_call": fpc} if fpc else {}))
            outs[tag] = (d, len(seen) - n0)
            monkeypatch.delenv("TIMED_SPARSE", raising=False)
    assert outs["h5"][1] == 0 and outs["sparse"][1] >= 1 and outs["sparse_small"][1] == 3 and outs.get("dense", (None, 0))[1] == 0, outs
    ref = outs["h5"][0]
    for tag, (d, _n) in outs.items():
        for fn in ("keras_tiny.csv", "keras_tiny.fasta", "keras_tiny.txt", "dataset.fasta", "datasetmap.txt", "encoded_labels.csv"):
            assert (ref / fn).read_bytes() == (d / fn).read_bytes(), (tag, fn)
