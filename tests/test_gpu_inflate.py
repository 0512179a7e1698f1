"""DEFLATE on the GPU (csrc/inflate.hip: one lane per stream) against zlib itself — the decoder behind th_h5_decode_device,
which replaces the host-side inflate of load_batch's per-residue reads (reference design_utils/utils.py:514-529).  Bit-exact or
a status code; corrupt streams must end in a status, never in a hang or a wild write."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

from timed_hip import _lib

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _inflate(lib, gpu, streams, sizes, wrapped=1, expect_ok=True):
    comp = b"".join(streams)
    src_off = np.cumsum([0] + [len(s) for s in streams[:-1]]).astype(np.int64)
    src_len = np.array([len(s) for s in streams], dtype=np.int64)
    dst_len = np.array(sizes, dtype=np.int64)
    dst_off = np.cumsum([0] + [(n + 7) // 8 * 8 for n in sizes[:-1]]).astype(np.int64)
    total = int(dst_off[-1] + (sizes[-1] + 7) // 8 * 8)
    out = np.full(total + 8, 0xAB, dtype=np.uint8)
    status = np.zeros(len(streams), dtype=np.int32)
    cbuf = np.frombuffer(comp + b"\0", dtype=np.uint8)
    rc = lib.th_inflate_many(gpu, cbuf.ctypes.data_as(C.c_void_p), len(comp), len(streams), src_off.ctypes.data_as(C.POINTER(C.c_int64)),
                             src_len.ctypes.data_as(C.POINTER(C.c_int64)), dst_off.ctypes.data_as(C.POINTER(C.c_int64)),
                             dst_len.ctypes.data_as(C.POINTER(C.c_int64)), out.ctypes.data_as(C.c_void_p), total, wrapped,
                             status.ctypes.data_as(C.POINTER(C.c_int)))
    if expect_ok:
        assert rc == 0, (rc, lib.th_last_error(), status[status != 0][:10])
    return rc, status, [bytes(out[o:o + n]) for o, n in zip(dst_off, sizes)], out


def _payloads(rng):
    """what a frame dataset holds, and what stresses a decoder"""
    out = []
    g = np.zeros((21, 21, 21, 6))                                   # sparse Gaussian frame (float64, mostly zeros)
    idx = rng.integers(0, g.size, 900)
    g.ravel()[idx] = rng.random(900)
    out.append(g.tobytes())
    out.append((rng.random((11, 11, 11, 3)) < 0.02).tobytes())       # boolean chunk
    out.append(rng.integers(0, 256, 70000, dtype=np.uint8).tobytes())  # incompressible: stored blocks at level 0, literals else
    out.append(b"")                                                   # empty
    out.append(b"a")
    out.append(b"abc" * 30000)                                        # long matches at distance 3
    out.append(bytes(range(256)) * 300)                               # distance 256
    out.append(np.repeat(rng.integers(0, 256, 400, dtype=np.uint8), rng.integers(1, 600, 400)).tobytes())   # runs: distance 1
    text = (b"the quick brown fox jumps over the lazy dog. " * 50) + rng.integers(97, 123, 5000, dtype=np.uint8).tobytes()
    out.append(text * 7)                                              # distances up to 32 K
    out.append(np.arange(40000, dtype=np.float64).tobytes())
    return out


def test_streams_of_every_block_type_decode_bit_exactly(gpu, lib):
    rng = np.random.default_rng(0)
    payloads = _payloads(rng)
    streams, sizes, want = [], [], []
    for level in (0, 1, 6, 9):                                        # 0: stored blocks; 1: fixed + dynamic; 6/9: dynamic
        for p in payloads:
            streams.append(zlib.compress(p, level)); sizes.append(len(p)); want.append(p)
    for p in payloads[:6]:                                            # Z_FIXED strategy: fixed-Huffman blocks only
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        streams.append(c.compress(p) + c.flush()); sizes.append(len(p)); want.append(p)
    rc, status, got, raw = _inflate(lib, gpu, streams, sizes)
    assert not status.any()
    for k, (a, b) in enumerate(zip(got, want)):
        assert a == b, f"stream {k} differs"
    # nothing was written outside the declared outputs (the 0xAB fill survives in the padding and the trailer)
    assert raw[-8:].tolist() == [0xAB] * 8


def test_raw_deflate_and_many_streams(gpu, lib):
    rng = np.random.default_rng(1)
    streams, sizes, want = [], [], []
    for k in range(700):                                              # more streams than one wavefront, ragged sizes
        n = int(rng.integers(0, 9000))
        p = (rng.integers(0, 4, n, dtype=np.uint8) * rng.integers(0, 2, n, dtype=np.uint8)).tobytes()
        c = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15)
        streams.append(c.compress(p) + c.flush()); sizes.append(n); want.append(p)
    rc, status, got, _ = _inflate(lib, gpu, streams, sizes, wrapped=0)
    assert got == want


def test_corrupt_streams_end_in_a_status(gpu, lib):
    rng = np.random.default_rng(2)
    good = zlib.compress(_payloads(rng)[0], 6)
    n = len(_payloads(np.random.default_rng(2))[0])
    streams, sizes = [good], [n]
    for k in range(255):
        b = bytearray(good)
        kind = k % 4
        if kind == 0:
            b = b[: int(rng.integers(2, len(b) - 1))]                 # truncated
        elif kind == 1:
            for pos in rng.integers(2, len(b), 3):
                b[pos] ^= 1 << int(rng.integers(0, 8))                # bit flips
        elif kind == 2:
            b[0] = int(rng.integers(0, 256))                          # header
        else:
            b = bytearray(rng.integers(0, 256, int(rng.integers(6, 400)), dtype=np.uint8).tobytes())   # garbage
        streams.append(bytes(b)); sizes.append(n)
    rc, status, got, raw = _inflate(lib, gpu, streams, sizes, expect_ok=False)
    assert status[0] == 0 and got[0] == zlib.decompress(good)
    for k in range(1, len(streams)):
        try:
            ref = zlib.decompress(streams[k])
        except zlib.error:
            ref = None
        if status[k] == 0:             # accepted here => accepted by zlib, with the same bytes (the Adler-32 trailer is checked)
            assert ref is not None and got[k] == ref
        elif ref is not None:          # rejected here but fine for zlib: only if it does not hold exactly the n declared bytes
            assert len(ref) != n
    assert rc == -2 and np.count_nonzero(status) > 200                # TH_EIO: most of them are rejected
    assert raw[-8:].tolist() == [0xAB] * 8
    assert lib.th_inflate_many(gpu, None, 0, 1, None, None, None, None, None, 0, 1, None) == -1


def _write_dataset(path, n_res=40, gaussian=True, seed=0):
    """an aposteriori-layout .hdf5 through timed_hip.h5write (gzip, one chunk per residue dataset)"""
    from timed_hip import voxeliser
    rng = np.random.default_rng(seed)
    if gaussian:
        frames = (rng.random((n_res, 21, 21, 21, 6)) * (rng.random((n_res, 21, 21, 21, 6)) < 0.08)).astype(np.float32)
    else:
        frames = (rng.random((n_res, 21, 21, 21, 6)) < 0.03).astype(np.float32)
    three = ["ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU", "MET", "ASN", "PRO", "GLN", "ARG", "SER", "THR", "VAL",
             "TRP", "TYR"]
    flat = [("1abc" if i < n_res // 2 else "2xyz", "A", str(i + 1), three[i % 20]) for i in range(n_res)]
    labels = np.zeros((n_res, 20), np.uint8)
    labels[np.arange(n_res), np.arange(n_res) % 20] = 1
    voxeliser.write_hdf5(path, frames, labels, flat, gaussian=gaussian, atom_encoder=list("CNOQP") + ["CA"])
    return frames, labels, flat


@pytest.mark.parametrize("gaussian", [True, False])
def test_hdf5_frames_decoded_on_the_gpu_equal_the_host_reader(gpu, tmp_path, gaussian):
    """th_h5_decode_device (B-trees on the host, DEFLATE + placement + float64 -> float32 on the GPU) against load_batch"""
    import warnings
    from design_utils import utils
    p = tmp_path / "d.hdf5"
    frames, labels, flat = _write_dataset(p, gaussian=gaussian)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap, _ = utils.create_flat_dataset_map(p)
        X, y = utils.load_batch(p, fmap, dtype=np.float32)
        got = utils.load_batch_device(p, np.array(fmap), device=gpu)
    assert got is not None, "the deflate-only dataset must take the device path"
    dev, yd = got
    assert dev.shape == X.shape and np.array_equal(yd, y)
    host = dev.buffer.download(dev.shape, dev.dtype)
    assert np.array_equal(host.astype(np.float32), X.astype(np.float32))
    sub = np.array(fmap)[[5, 3, 30, 7]]                         # an arbitrary selection of residues, not a contiguous range
    d2, y2 = utils.load_batch_device(p, sub, device=gpu)
    X2, yy2 = utils.load_batch(p, sub, dtype=np.float32)
    assert np.array_equal(d2.buffer.download(d2.shape, d2.dtype).astype(np.float32), X2.astype(np.float32)) and np.array_equal(y2, yy2)


@pytest.mark.parametrize("tok_kb", [64, 1000])
def test_decode_in_pieces_of_a_bounded_token_arena(gpu, tmp_path, monkeypatch, tok_kb):
    """the decoder's token arena is bounded (2 GB; the kernels run over pieces of the chunk list that fit it): with the bound forced
    down to one chunk per piece (64 KB) and to a ragged few (1000 KB: 15 chunks of 16.5 KB x 4 bytes per byte) the frames are the
    ones the host reader returns, whole batch and arbitrary selection"""
    import warnings
    from design_utils import utils
    monkeypatch.setenv("TH_INFLATE_TOK_KB", str(tok_kb))
    p = tmp_path / "d.hdf5"
    frames, labels, flat = _write_dataset(p, gaussian=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap, _ = utils.create_flat_dataset_map(p)
        X, y = utils.load_batch(p, fmap, dtype=np.float32)
        dev, yd = utils.load_batch_device(p, np.array(fmap), device=gpu)
        assert np.array_equal(dev.buffer.download(dev.shape, dev.dtype).astype(np.float32), X.astype(np.float32)) and np.array_equal(yd, y)
        sub = np.array(fmap)[[9, 2, 31, 4, 17]]
        d2, y2 = utils.load_batch_device(p, sub, device=gpu)
        X2, yy2 = utils.load_batch(p, sub, dtype=np.float32)
    assert np.array_equal(d2.buffer.download(d2.shape, d2.dtype).astype(np.float32), X2.astype(np.float32)) and np.array_equal(y2, yy2)
    utils.release_device_memory()


def test_predict_py_with_gpu_inflate_writes_the_same_files(gpu, tmp_path, monkeypatch):
    import warnings
    import predict
    from timed_hip import pack, synth
    p = tmp_path / "d.hdf5"
    _write_dataset(p, n_res=50)
    cfg, weights = synth.timed_synth(20, widths=(8, 16), side=21, in_channels=6, seed=3)
    mp = tmp_path / "M.pack"
    mp.write_bytes(pack.keras_to_pack(cfg, weights))
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    calls = []
    from design_utils import utils
    real = utils.load_batch_device
    monkeypatch.setattr(utils, "load_batch_device", lambda *x, **k: (calls.append(1), real(*x, **k))[1])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        predict.load_dataset_and_predict([mp], p, batch_size=7, dataset_map_path=a / "datasetmap.txt", path_to_output=a, frames_per_call=16)
        assert len(calls) == 4                                   # 50 frames in groups of 14 (2 reference batches): every group on the GPU
        monkeypatch.setenv("TIMED_GPU_INFLATE", "0")
        predict.load_dataset_and_predict([mp], p, batch_size=7, dataset_map_path=b / "datasetmap.txt", path_to_output=b, frames_per_call=16)
        assert len(calls) == 4
    for fn in sorted(x.name for x in a.iterdir()):
        assert (a / fn).read_bytes() == (b / fn).read_bytes(), fn


def test_multi_chunk_h5py_file_decoded_on_the_gpu_equals_h5py(gpu):
    """tests/golden/frames_chunked.hdf5 (real h5py: float64, gzip, automatic (6,11,11,3) chunking, 32 chunks per frame with
    partly-outside edge chunks): every chunk fits the LDS window, so k_lz_resolve places it straight into the frame — against
    h5py's own read of the file, for the whole map and for a permuted selection"""
    import os
    import warnings
    from design_utils import utils
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(G, "frames_chunked_expected.npz"))
    path = os.path.join(G, "frames_chunked.hdf5")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap = np.array(utils.create_flat_dataset_map(path)[0])
        got = utils.load_batch_device(path, fmap, device=gpu)
        assert got is not None, "a deflate-only float64 dataset must take the device path"
        dev, y = got
        assert dev.dtype == np.float32 and dev.shape == (5, 21, 21, 21, 6)
        assert np.array_equal(dev.buffer.download(dev.shape, dev.dtype), z["frames32"])
        assert np.array_equal(y.argmax(1), [(3 * r) % 20 for r in range(5)])
        order = [4, 0, 2]
        d2, _ = utils.load_batch_device(path, fmap[order], device=gpu)
        assert np.array_equal(d2.buffer.download(d2.shape, d2.dtype), z["frames32"][order])


def test_chunks_longer_than_the_deflate_window_but_whole_in_lds(gpu):
    """tests/golden/frames_midchunk.hdf5 (real h5py): 40 656-byte chunks, gzip alone (residues 3, 4) and shuffle + gzip (5, 6).
    Such a chunk is kept whole in LDS and placed from there (the fused path, d_raw is 16 bytes): the mid-stream flush of the
    ring mode must not run (it once wrote the stream's first bytes far outside d_raw).  Decoded into a poisoned neighbour-rich
    pool, against h5py's own read; also raw streams of 33-60 KB through th_inflate_many."""
    import os
    import warnings
    from design_utils import utils
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(G, "frames_midchunk_expected.npz"))["frames32"]
    path = os.path.join(G, "frames_midchunk.hdf5")
    utils._H5_KEEP.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap = np.array(utils.create_flat_dataset_map(path)[0])
        for order in ([0, 1], [1, 0, 1], [2, 3], [3, 2, 2]):        # one filter pipeline per batch (a mixed batch goes to the host reader)
            got = utils.load_batch_device(path, fmap[order], device=gpu)
            assert got is not None, "deflate / shuffle + deflate float64 residues must take the device path"
            dev, _y = got
            assert np.array_equal(dev.buffer.download(dev.shape, dev.dtype), z[order])
    utils._H5_KEEP.clear()
    lib = _lib.load()
    rng = np.random.default_rng(5)
    raws = []
    for n in (32769, 32776, 32777, 40000, 49152, 61440):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        a[rng.random(n) < 0.7] = 0                                   # long zero runs between literals
        raws.append(a.tobytes())
        raws.append(bytes(rng.integers(0, 4, n, dtype=np.uint8)))    # short alphabet: matches everywhere
    streams = [zlib.compress(r, 6) for r in raws]
    rc, status, got, _raw = _inflate(lib, gpu, streams, [len(r) for r in raws])
    assert rc == 0 and not status.any()
    assert all(g == r for g, r in zip(got, raws))


def test_boolean_h5py_fixture_decoded_on_the_gpu(gpu):
    """the deflate-only residues of tests/golden/frames_tiny_bool.hdf5 (real h5py, 1-byte elements: the non-converting placement)"""
    import os
    import warnings
    from design_utils import utils
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    path = os.path.join(G, "frames_tiny_bool.hdf5")
    rows = np.array([("1ubq", "A", r, "ALA") for r in ("11", "13", "5", "7")] + [("2xyz_0", "B", r, "ALA") for r in ("11", "5", "7")])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X, y = utils.load_batch(path, rows)
        got = utils.load_batch_device(path, rows, device=gpu)
    assert got is not None
    dev, yd = got
    assert np.array_equal(dev.buffer.download(dev.shape, dev.dtype).astype(X.dtype), X) and np.array_equal(yd, y)


def test_gpu_decode_is_taken_when_h5py_is_installed(gpu, monkeypatch):
    """`import h5py` working (the reference's normal environment) must not disable the device path: a stand-in h5py that refuses
    to open files is installed, load_batch_device still decodes on the GPU through the kept h5lite handle"""
    import os
    import sys
    import types
    import warnings
    from design_utils import utils
    fake = types.ModuleType("h5py")

    def _refuse(*a, **k):
        raise RuntimeError("h5py.File was called")
    fake.File = _refuse
    monkeypatch.setitem(sys.modules, "h5py", fake)
    utils._H5_KEEP.clear()
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(G, "frames_chunked_expected.npz"))
    path = os.path.join(G, "frames_chunked.hdf5")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap = np.array(utils.create_flat_dataset_map(path)[0])
        got = utils.load_batch_device(path, fmap, device=gpu)
    assert got is not None
    assert np.array_equal(got[0].buffer.download(got[0].shape, got[0].dtype), z["frames32"])
    utils._H5_KEEP.clear()


@pytest.mark.parametrize("wrapped", [1, 0])
def test_random_streams_of_every_strategy_match_zlib(gpu, lib, wrapped):
    """600 seeded random payloads (runs, periodic data of random period, noise, sparse doubles, mixtures; 0 .. 70 000 bytes)
    compressed at random levels with every zlib strategy (default, filtered, Huffman-only, RLE, fixed) and window size: the
    decoder's output is zlib's input, byte for byte"""
    rng = np.random.default_rng(20240 + wrapped)

    def payload():
        kind = int(rng.integers(0, 7))
        n = int(rng.choice([0, 1, 2, 3, 17, 255, 256, 257, 258, 259, 1000, 4096, 33000, 70000])) if rng.random() < 0.3 else int(rng.integers(0, 6000))
        if kind == 0:
            return bytes(n)
        if kind == 1:
            return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        if kind == 2:
            period = int(rng.integers(1, 300))
            return (rng.integers(0, 256, period, dtype=np.uint8).tobytes() * (n // period + 1))[:n]
        if kind == 3:
            return np.repeat(rng.integers(0, 4, max(1, n // 20), dtype=np.uint8), rng.integers(1, 40, max(1, n // 20))).tobytes()[:n]
        if kind == 4:
            g = np.zeros(max(1, n // 8))
            k = max(1, g.size // 10)
            g[rng.integers(0, g.size, k)] = rng.random(k)
            return g.tobytes()[:n]
        if kind == 5:
            return rng.integers(97, 101, n, dtype=np.uint8).tobytes()          # 4-letter alphabet: short codes, many matches
        a = rng.integers(0, 256, max(1, n // 3), dtype=np.uint8).tobytes()
        return (a + bytes(n // 3) + a)[:n]                                       # a far match across a run of zeros

    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]
    streams, sizes, want = [], [], []
    for _ in range(600):
        p = payload()
        wbits = int(rng.integers(9, 16))
        c = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, wbits if wrapped else -wbits, int(rng.integers(1, 10)),
                             strategies[int(rng.integers(0, 5))])
        half = len(p) // 2
        data = c.compress(p[:half])
        if rng.random() < 0.3:
            data += c.flush(zlib.Z_FULL_FLUSH)                       # an empty stored block in the middle of the stream
        data += c.compress(p[half:]) + c.flush()
        streams.append(data); sizes.append(len(p)); want.append(p)
    rc, status, got, _ = _inflate(lib, gpu, streams, sizes, wrapped=wrapped)
    assert rc == 0 and not status.any(), f"statuses {np.unique(status)}"
    bad = [k for k, (a, b) in enumerate(zip(got, want)) if a != b]
    assert not bad, f"streams {bad[:5]} differ"


def test_adler32_trailer_is_verified_like_zlib_does(gpu, lib):
    """zlib (and with it h5py's deflate filter) refuses a stream whose Adler-32 does not match its data; so does the GPU path:
    a flipped trailer byte, and a flipped payload byte inside a STORED block (which every Huffman-level check lets through),
    end in status 9 — for a stream kept whole in LDS and for one that goes round the 64 KB ring; raw deflate has no trailer"""
    rng = np.random.default_rng(5)
    small = rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    big = (rng.integers(0, 256, 3000, dtype=np.uint8).tobytes() * 40)[:100003]           # > 60 KB: the ring path, ragged length
    streams, sizes, expect = [], [], []
    for p in (small, big, b"", b"x"):
        for level in (0, 6):
            good = zlib.compress(p, level)
            streams.append(good); sizes.append(len(p)); expect.append(0)
            bad = bytearray(good); bad[-2] ^= 0x10                                        # trailer
            streams.append(bytes(bad)); sizes.append(len(p)); expect.append(9)
            if level == 0 and len(p) > 100:
                bad = bytearray(good); bad[len(good) // 2] ^= 0x01                        # payload of a stored block
                assert zlib.decompressobj().decompress(bytes(bad[:-4])) != p
                streams.append(bytes(bad)); sizes.append(len(p)); expect.append(9)
    rc, status, got, _ = _inflate(lib, gpu, streams, sizes, expect_ok=False)
    assert status.tolist() == expect, status.tolist()
    for k, e in enumerate(expect):
        if e == 0:
            assert got[k] == zlib.decompress(streams[k])
    # raw deflate: nothing to compare against, the payload flip in a stored block goes through (as it does in zlib)
    c = zlib.compressobj(0, zlib.DEFLATED, -15)
    raw = bytearray(c.compress(small) + c.flush()); raw[len(raw) // 2] ^= 1
    rc, status, got, _ = _inflate(lib, gpu, [bytes(raw)], [len(small)], wrapped=0)
    assert rc == 0 and status[0] == 0 and got[0] == zlib.decompressobj(-15).decompress(bytes(raw))


def test_corrupt_chunk_in_an_hdf5_file_is_an_error_not_a_frame(gpu, tmp_path):
    """a byte of one gzip chunk of the real-h5py fixture is flipped — once in the Adler-32 trailer, once in the middle of the
    compressed data: the host reader (zlib) raises, and so does the GPU path (th_h5_decode_device -> TH_EIO), instead of handing
    a wrong frame to the model"""
    import os
    import struct
    import warnings
    from design_utils import utils
    from timed_hip import h5lite
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    data = bytearray(open(os.path.join(G, "frames_chunked.hdf5"), "rb").read())
    with h5lite.File(os.path.join(G, "frames_chunked.hdf5")) as f:
        btree, _shape, chunk, _esz, _filters = f["1abc"]["A"]["5"].chunked_geometry()
        a = f._base + btree
        assert bytes(f._m[a:a + 4]) == b"TREE" and f._m[a + 5] == 0            # a leaf node
        csize = struct.unpack_from("<I", f._m, a + 24)[0]
        child = struct.unpack_from("<Q", f._m, a + 24 + 8 + 8 * (len(chunk) + 1))[0] + f._base
    for k, pos in enumerate((child + csize - 1, child + csize // 2)):
        bad = bytearray(data)
        bad[pos] ^= 0x20
        p = tmp_path / f"bad{k}.hdf5"
        p.write_bytes(bytes(bad))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fmap = np.array(utils.create_flat_dataset_map(p)[0])
            with pytest.raises(Exception):
                utils.load_batch(p, fmap, dtype=np.float32)
            with pytest.raises(RuntimeError, match="did not inflate"):
                utils.load_batch_device(p, fmap, device=gpu)
            good_rows = fmap[[0, 1, 3, 4]]                                      # the other residues are intact
            assert utils.load_batch_device(p, good_rows, device=gpu) is not None
    utils._H5_KEEP.clear()


@pytest.mark.parametrize("lpw", [8, 16, 32, 64])
def test_every_streams_per_wavefront_instantiation(gpu, lib, monkeypatch, lpw):
    """k_inflate_tokens<8 | 16 | 32 | 64> (the batch size picks one; TH_INFLATE_LPW forces it): the same 150 streams of every
    block type, partly filling the last wavefront, through each instantiation"""
    monkeypatch.setenv("TH_INFLATE_LPW", str(lpw))
    rng = np.random.default_rng(77)
    payloads = _payloads(rng)
    streams, sizes, want = [], [], []
    for k in range(150):
        p = payloads[k % len(payloads)]
        c = zlib.compressobj([0, 1, 6, 9][k % 4], zlib.DEFLATED, 15, 8, zlib.Z_FIXED if k % 7 == 0 else zlib.Z_DEFAULT_STRATEGY)
        streams.append(c.compress(p) + c.flush()); sizes.append(len(p)); want.append(p)
    rc, status, got, _ = _inflate(lib, gpu, streams, sizes)
    assert rc == 0 and not status.any()
    assert all(a == b for a, b in zip(got, want))


@pytest.mark.parametrize("name", ["frames_tiny.hdf5", "frames_tiny_bool.hdf5"])
def test_shuffle_plus_deflate_residues_of_the_h5py_fixtures_decode_on_the_gpu(gpu, name):
    """the real-h5py fixtures hold residues written with shuffle=True + gzip (filter pipeline 2, 1) and with gzip alone: both
    pipelines are decoded on the device (the unshuffle happens while the chunk is placed from LDS) and equal the host reader,
    which test_host_utils.py pins to h5py's own read; float32 Gaussian frames are placed as they are"""
    import os
    import warnings
    from design_utils import utils
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    path = os.path.join(G, name)
    shuffled = np.array([("1ubq", "A", "3", "ALA"), ("1ubq", "A", "9", "ALA"), ("2xyz_0", "A", "3", "ALA"), ("2xyz_0", "B", "3", "ALA"),
                         ("2xyz_0", "B", "9", "ALA")])
    plain = np.array([("1ubq", "A", "11", "ALA"), ("1ubq", "A", "13", "ALA"), ("2xyz_0", "B", "5", "ALA"), ("2xyz_0", "B", "7", "ALA")])
    utils._H5_KEEP.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for rows in (shuffled, plain):
            X, y = utils.load_batch(path, rows, dtype=np.float32)
            got = utils.load_batch_device(path, rows, device=gpu)
            assert got is not None, "deflate and shuffle + deflate residues must take the device path"
            dev, yd = got
            assert dev.shape == X.shape and np.array_equal(yd, y)
            assert np.array_equal(dev.buffer.download(dev.shape, dev.dtype).astype(X.dtype), X)
    utils._H5_KEEP.clear()


def test_never_allocated_chunks_are_zeros_on_the_gpu_too(gpu):
    """tests/golden/frames_partial.hdf5: a residue whose dataset was written only in part (its other chunks were never
    allocated).  The device path zeroes the batch when chunks are missing — decoded into a dirty, reused device buffer here —
    and skips the memset when every chunk is there (the first residue alone)"""
    import os
    import warnings
    from design_utils import utils
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(G, "frames_partial_expected.npz"))["frames32"]
    path = os.path.join(G, "frames_partial.hdf5")
    utils._H5_KEEP.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap = np.array(utils.create_flat_dataset_map(path)[0])
        for rows, want in ((fmap, z), (fmap[:1], z[:1]), (fmap[1:], z[1:]), (fmap, z)):
            got = utils.load_batch_device(path, rows, device=gpu)
            assert got is not None
            dev = got[0]
            assert np.array_equal(dev.buffer.download(dev.shape, dev.dtype), want)
            dev.buffer.upload(np.full(dev.shape, 7.0, np.float32))        # leave the pooled buffer dirty for the next round
            del dev, got
    utils._H5_KEEP.clear()


def test_duplicated_chunk_record_does_not_leave_stale_data(gpu, tmp_path):
    """a malformed chunk B-tree that lists chunk 1 twice and chunk 0 not at all has the full NUMBER of records: the count alone
    must not switch the zero-fill off — the region of the missing chunk reads as zeros (as for a never-allocated chunk), not as
    what the pooled device buffer held before.  A chunk offset that is not a multiple of the chunk size is refused."""
    import os
    import struct
    import warnings
    from design_utils import utils
    from timed_hip import h5lite
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    src = os.path.join(G, "frames_chunked.hdf5")
    z = np.load(os.path.join(G, "frames_chunked_expected.npz"))["frames32"]
    data = bytearray(open(src, "rb").read())
    with h5lite.File(src) as f:
        btree, _shape, chunk, _esz, _filters = f["1abc"]["A"]["5"].chunked_geometry()
        a = f._base + btree
        assert bytes(f._m[a:a + 4]) == b"TREE" and f._m[a + 5] == 0 and struct.unpack_from("<H", f._m, a + 6)[0] == 32
    rank1 = len(chunk) + 1
    ksz = 8 + 8 * rank1 + 8                                     # key (size, mask, offsets) + child pointer
    key0, key1 = a + 24, a + 24 + ksz
    offs0 = struct.unpack_from(f"<{rank1}Q", data, key0 + 8)
    offs1 = struct.unpack_from(f"<{rank1}Q", data, key1 + 8)
    assert offs0[:4] == (0, 0, 0, 0) and offs1[:4] == (0, 0, 0, 3)
    dup = bytearray(data)
    struct.pack_into(f"<{rank1}Q", dup, key0 + 8, *offs1)       # record 0 now claims to be chunk (0, 0, 0, 3) as well
    p = tmp_path / "dup.hdf5"
    p.write_bytes(bytes(dup))
    utils._H5_KEEP.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap = np.array(utils.create_flat_dataset_map(src)[0])
        row = fmap[[2]]                                          # residue "5"
        got = utils.load_batch_device(src, row, device=gpu)      # the intact file: leaves the right values in the pooled buffer
        assert np.array_equal(got[0].buffer.download(got[0].shape, got[0].dtype), z[[2]])
        assert np.count_nonzero(z[2, 0:6, 0:11, 0:11, 0:3]) > 10
        del got
        got = utils.load_batch_device(p, row, device=gpu)
        assert got is not None
        frame = got[0].buffer.download(got[0].shape, got[0].dtype)[0]
        assert not frame[0:6, 0:11, 0:11, 0:3].any(), "the chunk that is missing from the tree must read as zeros"
        assert np.array_equal(frame[6:], z[2, 6:])               # the chunks that are there once are placed as usual
        del got
        mis = bytearray(data)
        struct.pack_into(f"<{rank1}Q", mis, key1 + 8, 0, 0, 0, 2, 0)     # offset 2 along a dimension chunked by 3
        q = tmp_path / "misaligned.hdf5"
        q.write_bytes(bytes(mis))
        with pytest.raises(RuntimeError, match="multiple of the chunk size"):
            utils.load_batch_device(q, row, device=gpu)
    utils._H5_KEEP.clear()


def test_decoder_scratch_is_released_and_comes_back(gpu):
    """th_h5_release_scratch / design_utils.utils.release_device_memory: the decoder's device scratch and the pooled batch buffers are
    given back (free device memory grows), and the next decode allocates again and is still right"""
    import ctypes as C
    import os
    import warnings
    from design_utils import utils
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    path = os.path.join(G, "frames_chunked.hdf5")
    z = np.load(os.path.join(G, "frames_chunked_expected.npz"))["frames32"]
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        fr, tot = C.c_size_t(), C.c_size_t()
        assert hip.hipSetDevice(gpu) == 0 and hip.hipMemGetInfo(C.byref(fr), C.byref(tot)) == 0
        return fr.value
    utils._H5_KEEP.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap = np.array(utils.create_flat_dataset_map(path)[0])
        got = utils.load_batch_device(path, fmap, device=gpu)
        assert np.array_equal(got[0].buffer.download(got[0].shape, got[0].dtype), z)
        del got
        held = free_bytes()
        utils.release_device_memory([gpu])
        assert free_bytes() > held
        assert _lib.load().th_h5_release_scratch(gpu) == 0       # nothing left to free: still fine
        assert _lib.load().th_h5_release_scratch(99) == -1
        got = utils.load_batch_device(path, fmap, device=gpu)
        assert np.array_equal(got[0].buffer.download(got[0].shape, got[0].dtype), z)
    utils._H5_KEEP.clear()


def test_sharded_rccl_path_with_gpu_inflated_frames(gpu, tmp_path, monkeypatch):
    """the one-process-per-GPU path of predict.py (here: a 1-rank RCCL communicator) on a gzip .hdf5: the frames of every group
    are inflated on the GPU (DeviceFrames), predicted into the device shard buffer (TH_PREDICT_IN_DEVICE | TH_PREDICT_OUT_DEVICE),
    gathered with th_comm_gather_rows — and every file equals the plain host-reader run byte for byte"""
    import os
    import warnings
    import predict
    from design_utils import utils
    from timed_hip import distributed as td, pack, synth
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    data = os.path.join(G, "frames_chunked.hdf5")                 # real h5py, 32 chunks per frame
    cfg, weights = synth.timed_synth(20, widths=(8, 16), side=21, in_channels=6, seed=5)
    mp = tmp_path / "M.pack"
    mp.write_bytes(pack.keras_to_pack(cfg, weights))
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    calls = []
    real = utils.load_batch_device
    monkeypatch.setattr(utils, "load_batch_device", lambda *x, **k: (calls.append(1), real(*x, **k))[1])
    comm = td.RcclGather(td.RcclGather.new_unique_id(), 1, 0, gpu)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        predict.load_dataset_and_predict([mp], data, batch_size=2, dataset_map_path=b / "datasetmap.txt", path_to_output=b,
                                         frames_per_call=2, gather=comm)
        assert len(calls) == 3                                    # 5 frames in groups of 2: every group decoded on the GPU
        monkeypatch.setenv("TIMED_GPU_INFLATE", "0")
        predict.load_dataset_and_predict([mp], data, batch_size=2, dataset_map_path=a / "datasetmap.txt", path_to_output=a)
        assert len(calls) == 3
    comm.close()
    for fn in sorted(x.name for x in a.iterdir()):
        assert (a / fn).read_bytes() == (b / fn).read_bytes(), fn
    utils._H5_KEEP.clear()
