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


@pytest.mark.parametrize("name", ["frames_tiny.hdf5", "frames_tiny_bool.hdf5"])
def test_native_header_resolution_equals_python_reader(name):
    """th_h5_resolve (object headers, `encoded_residue`, `label` for a whole batch in one call) against h5lite's own
    per-dataset parse, which is pinned to real-h5py fixtures in test_host_utils.py"""
    path = os.path.join(G, name)
    with h5lite.File(path) as f:
        addrs, want_label, want_enc, want_geo = [], [], [], []
        for pdb in f:
            for chain in f[pdb].keys():
                grp = f[pdb][chain]
                links = grp._load()
                for res in grp.keys():
                    ds = grp[res]
                    addrs.append(links[res])
                    want_label.append(utils._as_str(ds.attrs["label"]))
                    want_enc.append(np.asarray(ds.attrs["encoded_residue"], dtype=float))
                    want_geo.append(ds.chunked_geometry())
        r = h5lite.resolve_many(f, addrs, num_attr="encoded_residue", num_len=20, str_attr="label", str_len=16)
        assert r is not None and np.all((r["status"] & 6) == 6)
        # bit 1 = "stored exactly like the first dataset" (the fixtures mix contiguous and gzip-chunked residues on purpose;
        # a real aposteriori file is homogeneous): the others are left to the general reader
        same_as_first = np.array([(geo is None) == (want_geo[0] is None) and (geo is None or geo[1:] == want_geo[0][1:])
                                  for geo in want_geo])
        assert np.array_equal((r["status"] & 1).astype(bool), same_as_first)
        assert r["strs"] == want_label
        assert np.array_equal(r["num"], np.stack(want_enc))
        g = r["geom"]
        rank = int(g[0])
        for i, geo in enumerate(want_geo):
            if not same_as_first[i]:
                assert int(r["btree"][i]) == -1
                continue
            if geo is None:                      # contiguous storage: the address slot carries the data address
                assert int(g[27]) == 1
                continue
            btree, shape, chunk, esz, filters = geo
            assert int(g[27]) == 2 and int(r["btree"][i]) == btree
            assert tuple(g[1:1 + rank]) == shape and tuple(g[8:8 + rank]) == chunk and int(g[15]) == esz
            assert tuple(int(x) for x in g[19:19 + int(g[18])]) == filters
        # an attribute that is not there / a wrong length leaves the bit clear instead of guessing
        r2 = h5lite.resolve_many(f, addrs[:3], num_attr="encoded_residue", num_len=19, str_attr="nope", str_len=8)
        assert np.all((r2["status"] & 6) == 0) and np.array_equal((r2["status"] & 1).astype(bool), same_as_first[:3])
        # a group's header is not a dataset: no geometry bit, nothing crashes
        root_links = f._root._load()
        r3 = h5lite.resolve_many(f, [next(iter(root_links.values()))], num_attr="encoded_residue", num_len=20)
        assert int(r3["status"][0]) == 0 and int(r3["btree"][0]) == -1
        # garbage addresses are refused, not dereferenced
        r4 = h5lite.resolve_many(f, [addrs[0], 10 ** 12, 3], num_attr="encoded_residue", num_len=20)
        assert int(r4["status"][0]) == 3 and int(r4["status"][1]) == 0 and (int(r4["status"][2]) & 1) == 0
        # a batch whose first dataset is of the OTHER storage kind resolves that kind natively instead
        other = int(np.nonzero(~same_as_first)[0][0]) if not same_as_first.all() else None
        if other is not None:
            r5 = h5lite.resolve_many(f, [addrs[other]] + addrs, num_attr="encoded_residue", num_len=20)
            got = (r5["status"][1:] & 1).astype(bool)
            assert (r5["status"][0] & 1) and got[other] and not np.any(got & same_as_first)


def test_load_batch_float32_option_is_the_keras_cast():
    """dtype=np.float32: the float64 -> float32 rounding happens while the chunks are placed and equals NumPy's cast of the
    reference-dtype batch bit for bit; labels unchanged; boolean datasets ignore the option"""
    path = os.path.join(G, "frames_tiny.hdf5")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flat, _ = utils.create_flat_dataset_map(path)
    X64, y64 = utils.load_batch(path, flat)
    X32, y32 = utils.load_batch(path, flat, dtype=np.float32)
    assert X64.dtype == np.float64 and X32.dtype == np.float32
    assert np.array_equal(X32, X64.astype(np.float32)) and np.array_equal(y32, y64)
    # arbitrary row order and repeated rows
    pick = [flat[i] for i in (5, 0, 5, 25, 12)]
    Xp, yp = utils.load_batch(path, pick, dtype=np.float32)
    assert np.array_equal(Xp, X32[[5, 0, 5, 25, 12]]) and np.array_equal(yp, y64[[5, 0, 5, 25, 12]])
    pb = os.path.join(G, "frames_tiny_bool.hdf5")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fb, _ = utils.create_flat_dataset_map(pb)
    Xb, _ = utils.load_batch(pb, fb, dtype=np.float32)
    assert Xb.dtype == bool


def test_load_batch_falls_back_per_dataset(monkeypatch):
    """rows the native pass declines go through the general reader, the others stay native"""
    path = os.path.join(G, "frames_tiny.hdf5")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flat, _ = utils.create_flat_dataset_map(path)
    want_X, want_y = utils.load_batch(path, flat)
    real = h5lite.resolve_many

    def flaky(f, addrs, **kw):
        r = real(f, addrs, **kw)
        r["status"][::3] = 0            # pretend every third header was unusual
        return r
    monkeypatch.setattr(h5lite, "resolve_many", flaky)
    X, y = utils.load_batch(path, flat)
    assert np.array_equal(X, want_X) and np.array_equal(y, want_y)
    monkeypatch.setattr(h5lite, "resolve_many", lambda *a, **k: None)      # no native library at all
    X, y = utils.load_batch(path, flat)
    assert np.array_equal(X, want_X) and np.array_equal(y, want_y)


def test_user_defined_fill_values_are_honoured_not_zeroed():
    """never-written elements read as the dataset's fill value (fixture written and read back by real h5py);
    the native bulk paths, which zero-fill, decline such datasets instead of guessing"""
    path = os.path.join(G, "fillvalue_tiny.hdf5")
    want = np.load(os.path.join(G, "fillvalue_expected.npz"))
    with h5lite.File(path) as f:
        for name in ("partial", "never_chunked", "never_contiguous", "zero_fill"):
            got = np.asarray(f[name][()])
            assert got.dtype == want[name].dtype and np.array_equal(got, want[name]), name
        assert want["partial"][5, 7] == 2.5 and want["never_contiguous"][0] == 7
        assert f["partial"].chunked_geometry() is None and f["never_chunked"].chunked_geometry() is None
        assert f["zero_fill"].chunked_geometry() is not None
        dest = np.empty((6, 8))
        assert f["partial"].read_direct(dest) is False
        assert h5lite.read_many_direct([f["partial"], f["zero_fill"]], [np.empty((6, 8)), dest]) == [False, True]
        assert np.array_equal(dest, want["zero_fill"])
        links = f._root._load()
        r = h5lite.resolve_many(f, [links["zero_fill"], links["partial"]])
        assert (int(r["status"][0]) & 1) == 1 and (int(r["status"][1]) & 1) == 0
        r = h5lite.resolve_many(f, [links["partial"]])
        assert (int(r["status"][0]) & 1) == 0


def test_bulk_reader_requires_matching_dtype_kind():
    """a 1-byte integer dataset is not silently byte-copied into a bool array (or uint8 into int8 ...)"""
    path = os.path.join(G, "frames_tiny_bool.hdf5")
    with h5lite.File(path) as f:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            flat, _ = utils.create_flat_dataset_map(path)
        dss = [f[p][c][r] for p, c, r, _ in flat]
        ds = next(d for d in dss if d.chunked_geometry() is not None)
        w = np.asarray(ds[()])
        assert w.dtype == bool
        assert h5lite.read_many_direct([ds], [np.empty(w.shape, np.int8)]) == [False]
        ok = np.empty(w.shape, bool)
        assert h5lite.read_many_direct([ds], [ok]) == [True] and np.array_equal(ok, w)


def test_resolver_survives_truncated_and_corrupted_files():
    """th_h5_resolve parses untrusted bytes.  The valid fixture is truncated at every 97th byte and corrupted with
    random byte flips inside the object headers; the buffer handed to the library is EXACTLY file-sized and sits at the end of
    a page-aligned allocation followed by a PROT_NONE guard page, so a read past the end is a SIGSEGV, not luck.  Every call
    must return (status bits clear where it cannot parse) — the caller then falls back to the general reader."""
    import ctypes as C
    import mmap
    from timed_hip import _lib, h5lite
    lib = _lib.load()
    path = os.path.join(G, "frames_tiny.hdf5")
    data = open(path, "rb").read()
    with h5lite.File(path) as f:
        addrs = []
        for pdb in f:
            for chain in f[pdb]:
                links = f[pdb][chain]._load()
                addrs += [links[k] for k in links]
        base = f._base
    addrs = np.asarray(addrs, dtype=np.int64)
    n = len(addrs)
    libc = C.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    page = mmap.PAGESIZE

    def run(buf: bytes):
        size = len(buf)
        if size == 0:
            return None
        npages = (size + page - 1) // page + 1
        m = mmap.mmap(-1, npages * page)
        start = (npages - 1) * page - size                   # the data ends exactly where the guard page begins
        m[start:start + size] = buf
        addr = C.addressof(C.c_char.from_buffer(m))
        assert libc.mprotect(addr + (npages - 1) * page, page, 0) == 0
        status = np.zeros(n, np.int32); btree = np.zeros(n, np.int64); geom = np.zeros(40, np.int64)
        num = np.zeros((n, 20)); sbuf = np.zeros((n, 16), np.uint8)
        rc = lib.th_h5_resolve(C.c_void_p(addr + start), size, base, n, addrs.ctypes.data_as(C.POINTER(C.c_int64)), b"encoded_residue",
                               num.ctypes.data, 20, b"label", sbuf.ctypes.data, 16, btree.ctypes.data_as(C.POINTER(C.c_int64)),
                               geom.ctypes.data_as(C.POINTER(C.c_int64)), status.ctypes.data_as(C.POINTER(C.c_int)), 1)
        libc.mprotect(addr + (npages - 1) * page, page, 3)
        del addr
        m.close()
        return rc, status

    rc, status = run(data)
    # the intact file resolves completely (bit 1 = "same geometry as the first dataset": the fixture mixes two dtypes)
    assert rc == 0 and np.all(status & 2) and np.all(status & 4) and np.count_nonzero(status & 1) >= n // 2
    for cut in range(len(data) - 1, 64, -97):                # truncations: headers, heaps, chunk data cut anywhere
        rc, status = run(data[:cut])
        assert rc == 0
    rng = np.random.default_rng(0)
    lo, hi = int(addrs.min()) + base, min(len(data), int(addrs.max()) + base + 2048)
    for _ in range(300):                                     # random corruption inside the object-header region
        b = bytearray(data)
        for pos in rng.integers(lo, hi, size=int(rng.integers(1, 12))):
            b[pos] = int(rng.integers(0, 256))
        rc, status = run(bytes(b))
        assert rc == 0


def test_load_batch_on_corrupted_files_raises_or_returns(tmp_path):
    """the whole ingest path (h5lite walk -> th_h5_resolve -> th_h5_read_chunked_as, general reader as fallback) on files with
    random byte flips anywhere: a Python exception or a result, never a crash of the process"""
    path = os.path.join(G, "frames_tiny.hdf5")
    data = open(path, "rb").read()
    rng = np.random.default_rng(1)
    outcomes = {"ok": 0, "raised": 0}
    for k in range(60):
        b = bytearray(data)
        for pos in rng.integers(8, len(b), size=int(rng.integers(1, 6))):
            b[pos] = int(rng.integers(0, 256))
        p = tmp_path / f"c{k}.hdf5"
        p.write_bytes(bytes(b))
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                flat, _ = utils.create_flat_dataset_map(p)
                X, y = utils.load_batch(p, flat)
            assert X.shape[0] == len(flat)
            outcomes["ok"] += 1
        except Exception:
            outcomes["raised"] += 1
        p.unlink()
    assert outcomes["ok"] + outcomes["raised"] == 60 and outcomes["ok"] > 0


def _group_tables(f):
    """(btree, heap data offset, heap size, python-walked links) of every old-style group of an h5lite file"""
    import struct
    out = []
    todo, seen = [f._root], set()
    while todo:
        g = todo.pop()
        if g._addr in seen:
            continue
        seen.add(g._addr)
        for mtype, _fl, d in g._messages():
            if mtype == 0x11:
                btree, heap = struct.unpack_from("<QQ", d, 0)
                hd = g._local_heap(heap)
                links = {}
                g._walk_btree(btree, hd, links)
                out.append((btree, hd[0], hd[1], links))
        for name in g.keys():
            child = g[name]
            if isinstance(child, h5lite.Group):
                todo.append(child)
    return out


@pytest.mark.parametrize("name", ["frames_tiny.hdf5", "frames_tiny_bool.hdf5"])
def test_native_group_listing_equals_interpreter_walk(name):
    """th_h5_group_links (what Group._load now calls) against h5lite's own symbol-table walk, which the real-h5py fixtures of
    test_host_utils.py pin: same names, same object-header addresses, same (B-tree) order; a capacity one short is refused"""
    import ctypes as C
    from timed_hip import _lib
    lib = _lib.load()
    with h5lite.File(os.path.join(G, name)) as f:
        tables = _group_tables(f)
        assert len(tables) >= 3
        whole = np.frombuffer(f._m, dtype=np.uint8)
        for btree, hoff, hsize, links in tables:
            names = np.zeros(hsize + 8, np.uint8)
            addrs = np.zeros(hsize + 8, np.int64)
            n, used = C.c_int64(), C.c_int64()
            rc = lib.th_h5_group_links(whole.ctypes.data_as(C.c_void_p), whole.size, f._base, btree, hoff, hsize,
                                       names.ctypes.data_as(C.c_void_p), names.size, addrs.ctypes.data_as(C.POINTER(C.c_int64)), addrs.size,
                                       C.byref(n), C.byref(used))
            assert rc == 0 and n.value == len(links)
            got = names[:used.value].tobytes().decode().split("\0")[:-1]
            assert got == list(links.keys()) and addrs[:n.value].tolist() == list(links.values())
            if n.value:
                rc = lib.th_h5_group_links(whole.ctypes.data_as(C.c_void_p), whole.size, f._base, btree, hoff, hsize,
                                           names.ctypes.data_as(C.c_void_p), names.size, addrs.ctypes.data_as(C.POINTER(C.c_int64)),
                                           n.value - 1, C.byref(n), C.byref(used))
                assert rc != 0
                rc = lib.th_h5_group_links(whole.ctypes.data_as(C.c_void_p), whole.size, f._base, btree, hoff, hsize,
                                           names.ctypes.data_as(C.c_void_p), used.value - 1, addrs.ctypes.data_as(C.POINTER(C.c_int64)),
                                           addrs.size, C.byref(n), C.byref(used))
                assert rc != 0
        del whole
        # and the product path: Group._load through the native call == through the interpreter walk
        g = next(iter(f[next(iter(f))].keys()))
        grp = f[next(iter(f))][g]
        native = dict(grp._load())
        grp._links = None
        grp._links_native = lambda *a: False
        assert grp._load() == native and list(grp._load().keys()) == list(native.keys())


def test_group_listing_survives_truncated_and_corrupted_files():
    """th_h5_group_links parses untrusted bytes: truncations and byte flips inside the B-tree / SNOD / heap region, the buffer
    ending exactly at a PROT_NONE guard page — every call returns (0 or an error code), none reads past the buffer"""
    import ctypes as C
    import mmap
    from timed_hip import _lib
    lib = _lib.load()
    path = os.path.join(G, "frames_tiny.hdf5")
    data = open(path, "rb").read()
    with h5lite.File(path) as f:
        tables = [(b, o, s) for b, o, s, _ in _group_tables(f)]
        base = f._base
    libc = C.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    page = mmap.PAGESIZE
    names = np.zeros(1 << 16, np.uint8)
    addrs = np.zeros(1 << 13, np.int64)

    def run(buf: bytes):
        size = len(buf)
        npages = (size + page - 1) // page + 1
        m = mmap.mmap(-1, npages * page)
        start = (npages - 1) * page - size
        m[start:start + size] = buf
        addr = C.addressof(C.c_char.from_buffer(m))
        assert libc.mprotect(addr + (npages - 1) * page, page, 0) == 0
        rcs = []
        for btree, hoff, hsize in tables:
            n, used = C.c_int64(), C.c_int64()
            rcs.append(lib.th_h5_group_links(C.c_void_p(addr + start), size, base, btree, hoff, hsize, names.ctypes.data_as(C.c_void_p),
                                             names.size, addrs.ctypes.data_as(C.POINTER(C.c_int64)), addrs.size, C.byref(n), C.byref(used)))
            assert 0 <= n.value <= addrs.size and 0 <= used.value <= names.size
        libc.mprotect(addr + (npages - 1) * page, page, 3)
        del addr
        m.close()
        return rcs

    assert all(rc == 0 for rc in run(data))
    for cut in range(len(data) - 1, 64, -89):
        run(data[:cut])
    rng = np.random.default_rng(1)
    lo = min(min(b + base, o) for b, o, _ in tables)
    hi = min(len(data), max(max(b + base, o + s) for b, o, s in tables) + 4096)
    for _ in range(300):
        b = bytearray(data)
        for pos in rng.integers(lo, hi, size=int(rng.integers(1, 12))):
            b[pos] = int(rng.integers(0, 256))
        run(bytes(b))
    # wild arguments
    n, used = C.c_int64(), C.c_int64()
    buf = np.frombuffer(data, np.uint8)
    for btree, hoff, hsize in [(10 ** 15, 0, 8), (-5, 0, 8), (tables[0][0], len(data) - 4, 64), (tables[0][0], -1, 8), (tables[0][0], 0, -3)]:
        rc = lib.th_h5_group_links(buf.ctypes.data_as(C.c_void_p), buf.size, base, btree, hoff, hsize, names.ctypes.data_as(C.c_void_p),
                                   names.size, addrs.ctypes.data_as(C.POINTER(C.c_int64)), addrs.size, C.byref(n), C.byref(used))
        assert rc != 0


def test_multi_chunk_float64_frames_equal_h5py(tmp_path):
    """tests/golden/frames_chunked.hdf5 is what h5py writes for the reference's real data: (21,21,21,6) float64, gzip, h5py's
    automatic (6,11,11,3) chunking — 32 chunks per frame, the upper-edge ones only partly inside the dataset.  The map, the
    labels and load_batch (float64 as stored, and the float32 option) against h5py's own read of the file (the .npz)."""
    z = np.load(os.path.join(G, "frames_chunked_expected.npz"))
    path = os.path.join(G, "frames_chunked.hdf5")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap, pdbs = utils.create_flat_dataset_map(path)
        assert [r[2] for r in fmap] == list(z["residues"]) and [r[3] for r in fmap] == list(z["labels"]) and pdbs == {"1abc"}
        X, y = utils.load_batch(path, fmap)
        X32, _ = utils.load_batch(path, fmap, dtype=np.float32)
    assert X.dtype == np.float64 and np.array_equal(X.astype(np.float32), z["frames32"])
    assert X32.dtype == np.float32 and np.array_equal(X32, z["frames32"])
    assert y.shape == (5, 20) and np.array_equal(y.argmax(1), [(3 * r) % 20 for r in range(5)])
    with h5lite.File(path) as f:
        geo = f["1abc"]["A"]["3"].chunked_geometry()
    assert geo[2] == (6, 11, 11, 3) and geo[4] == (1,)


def test_mid_size_chunks_equal_h5py():
    """tests/golden/frames_midchunk.hdf5 (real h5py): 40 656-byte chunks — longer than the DEFLATE window, shorter than the GPU
    decoder's whole-stream limit — gzip alone and shuffle + gzip, on the host reader (the GPU twin is in test_gpu_inflate.py)"""
    z = np.load(os.path.join(G, "frames_midchunk_expected.npz"))["frames32"]
    path = os.path.join(G, "frames_midchunk.hdf5")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap, _ = utils.create_flat_dataset_map(path)
        X32, _y = utils.load_batch(path, fmap, dtype=np.float32)
    assert np.array_equal(X32, z)
    with h5lite.File(path) as f:
        assert f["1abc"]["A"]["3"].chunked_geometry()[2] == (7, 11, 11, 6)
        assert f["1abc"]["A"]["3"].chunked_geometry()[4] == (1,) and f["1abc"]["A"]["5"].chunked_geometry()[4] == (2, 1)


def test_native_map_path_does_not_depend_on_h5py_being_absent(monkeypatch):
    """the reference's users HAVE h5py installed: the native map path (and with it load_batch_device's GPU decode, which shares
    the kept h5lite handle) must not switch itself off because `import h5py` works.  A stand-in h5py whose File refuses to open
    anything proves that the map is built without it; a file h5lite cannot read falls through to h5py (here: the refusal)."""
    import sys
    import types
    fake = types.ModuleType("h5py")

    def _refuse(*a, **k):
        raise RuntimeError("h5py.File was called")
    fake.File = _refuse
    monkeypatch.setitem(sys.modules, "h5py", fake)
    utils._H5_KEEP.clear()
    path = os.path.join(G, "frames_chunked.hdf5")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap, _ = utils.create_flat_dataset_map(path)
    assert len(fmap) == 5 and utils._kept_h5lite(path) is not None
    X32, y = utils.load_batch(path, fmap, dtype=np.float32)           # the native host reader, too
    assert np.array_equal(X32, np.load(os.path.join(G, "frames_chunked_expected.npz"))["frames32"]) and y.shape == (5, 20)
    bogus = os.path.join(G, "h5_expected.npz")                       # not an HDF5 file
    assert utils._kept_h5lite(bogus) is None
    with pytest.raises(RuntimeError, match="h5py.File was called"):
        utils.create_flat_dataset_map(bogus)
    utils._H5_KEEP.clear()


def test_fletcher32_is_verified_by_both_host_readers(tmp_path):
    """tests/golden/frames_fletcher.hdf5 (real h5py: fletcher32 alone, over gzip, over shuffle + gzip, and on a multi-chunk
    dataset): HDF5 checks the checksum on every read and fails the read on a mismatch.  Both readers here reproduce the
    library's checksums (the intact file reads back exactly) and refuse a chunk with one flipped data byte."""
    import struct
    z = np.load(os.path.join(G, "frames_fletcher_expected.npz"))["frames"]
    path = os.path.join(G, "frames_fletcher.hdf5")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap, _ = utils.create_flat_dataset_map(path)
        X, y = utils.load_batch(path, fmap, dtype=np.float32)                      # the native threaded reader
    assert np.array_equal(X, z) and [r[2] for r in fmap] == ["1", "2", "3", "4"]
    with h5lite.File(path) as f:                                                   # the pure-Python reader
        for r in range(4):
            assert np.array_equal(np.asarray(f["1abc"]["A"][str(r + 1)][()]), z[r])
        btree, _shape, chunk, _esz, filters = f["1abc"]["A"]["1"].chunked_geometry()
        assert filters == (3,)
        a = f._base + btree
        csize = struct.unpack_from("<I", f._m, a + 24)[0]
        child = struct.unpack_from("<Q", f._m, a + 24 + 8 + 8 * (len(chunk) + 1))[0] + f._base
    data = bytearray(open(path, "rb").read())
    data[child + csize // 2] ^= 0x04
    bad = tmp_path / "bad.hdf5"
    bad.write_bytes(bytes(data))
    with h5lite.File(str(bad)) as f:
        with pytest.raises(h5lite.H5FormatError, match="fletcher32"):
            f["1abc"]["A"]["1"][()]
        assert np.array_equal(np.asarray(f["1abc"]["A"]["2"][()]), z[1])          # the other datasets are intact
    from timed_hip import _lib
    lib = _lib.load()
    with h5lite.File(str(bad)) as f:                                               # the native reader declines the chunk ...
        ds = f["1abc"]["A"]["1"]
        geo = ds.chunked_geometry()
        dest = np.empty(ds.shape, np.float32)
        import ctypes as C
        whole = np.frombuffer(f._m, dtype=np.uint8)
        rc = lib.th_h5_read_chunked(whole.ctypes.data_as(C.c_void_p), whole.size, f._base, 1, (C.c_int64 * 1)(geo[0]),
                                    (C.c_void_p * 1)(dest.ctypes.data), 4, (C.c_int64 * 4)(*geo[1]), (C.c_int64 * 4)(*geo[2]), 4, 1,
                                    (C.c_int * 1)(3), 1)
        del whole
        assert rc != 0 and b"fletcher32" in lib.th_last_error()
    with warnings.catch_warnings():                                                # ... and load_batch as a whole raises
        warnings.simplefilter("ignore")
        with pytest.raises(h5lite.H5FormatError, match="fletcher32"):
            utils.load_batch(str(bad), fmap, dtype=np.float32)
    utils._H5_KEEP.clear()


def test_never_allocated_chunks_read_as_zeros():
    """tests/golden/frames_partial.hdf5 (real h5py): one residue was written only in part, its other chunks do not exist in the
    file — they read as zeros, as h5py returns them (the .npz)"""
    z = np.load(os.path.join(G, "frames_partial_expected.npz"))["frames32"]
    path = os.path.join(G, "frames_partial.hdf5")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap, _ = utils.create_flat_dataset_map(path)
        out = np.full((2, 21, 21, 21, 6), 7.0, np.float32)            # a reused, dirty batch buffer
        X, _y = utils.load_batch(path, fmap, dtype=np.float32, out=out)
    assert np.array_equal(X, z) and np.count_nonzero(z[1][6:]) == 0
    utils._H5_KEEP.clear()


def test_decode_device_refuses_bad_arguments_before_touching_a_gpu():
    """th_h5_decode_device validates its arguments (null pointers, rank, element size, filter pipeline, conversion) before any
    HIP call: the refusals are testable without a GPU, and an unsupported pipeline is TH_EUNSUP (-4: "use the host reader")"""
    import ctypes as C
    from timed_hip import _lib
    lib = _lib.load()
    buf = np.zeros(4096, np.uint8)
    shape, chunk = (C.c_int64 * 2)(4, 4), (C.c_int64 * 2)(2, 2)
    one = (C.c_int64 * 1)(128)

    def call(n=1, addrs=one, rank=2, esz=8, nf=1, filt=(1,), conv=0, out=1):
        return lib.th_h5_decode_device(buf.ctypes.data_as(C.c_void_p), buf.size, 0, n, addrs, rank, shape, chunk, esz, nf,
                                       (C.c_int * max(1, len(filt)))(*filt), conv, 0, C.c_void_p(out))
    assert call(filt=(32000,)) == -4 and call(nf=2, filt=(1, 2)) == -4 and call(nf=3, filt=(2, 1, 3)) == -4     # TH_EUNSUP
    assert call(rank=0) == -4 and call(rank=9) == -4 and call(esz=0) == -4
    assert call(conv=1, esz=4) == -1 and call(conv=7) == -1                                                      # TH_EINVAL
    assert call(addrs=None) == -1 and call(out=0) == -1
    assert call(n=0, addrs=None, out=0) == 0                                                                     # nothing to do


def test_gpu_decode_retries_in_pieces_when_the_device_is_out_of_memory(monkeypatch):
    """h5lite.decode_resolved_device: TH_ENOMEM from th_h5_decode_device (its token arena did not fit) splits the batch in halves
    down to MIN_DECODE datasets; every piece lands at its own offset of the output; below that the error surfaces and
    load_batch_device turns it into "use the host reader" (None + a warning) instead of aborting predict.py.  The native call is
    replaced by a recorder: no GPU needed."""
    from timed_hip import _lib
    path = os.path.join(G, "frames_chunked.hdf5")
    real = _lib.load()
    calls = []

    class Fake:
        limit = 1

        def __getattr__(self, name):
            return getattr(real, name)

        def th_h5_decode_device(self, file, file_len, base, n, addrs, rank, shape, chunk, esz, nf, filters, conv, device, d_out):
            calls.append((int(n), int(d_out.value)))
            return _lib.TH_ENOMEM if n > self.limit else 0
    fake = Fake()
    monkeypatch.setattr(_lib, "load", lambda: fake)
    monkeypatch.setattr(h5lite, "MIN_DECODE", 1)
    utils._H5_KEEP.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap = np.array(utils.create_flat_dataset_map(path)[0])
    with h5lite.File(path) as f:
        links = f["1abc"]["A"]._load()
        addrs = np.array([links[r[2]] for r in fmap], dtype=np.int64)
        res = h5lite.resolve_many(f, addrs, num_attr="encoded_residue", num_len=20)
        assert h5lite.decode_resolved_device(f, res, 1 << 20, 0, as_float32=True)
        frame_bytes = 21 * 21 * 21 * 6 * 4
        done = sorted((off - (1 << 20)) // frame_bytes for n, off in calls if n == 1)
        assert done == [0, 1, 2, 3, 4] and calls[0] == (5, 1 << 20)
        assert all((off - (1 << 20)) % frame_bytes == 0 for _n, off in calls)
        calls.clear()
        monkeypatch.setattr(h5lite, "MIN_DECODE", 64)             # 5 datasets cannot be split: the error surfaces ...
        with pytest.raises(_lib.TimedHipError) as e:
            h5lite.decode_resolved_device(f, res, 1 << 20, 0, as_float32=True)
        assert e.value.code == _lib.TH_ENOMEM and calls == [(5, 1 << 20)]
    # ... and load_batch_device hands the batch to the host reader
    class FakeBuf:
        ptr = 1 << 20
        device = 0
        nbytes = 1 << 40

        def free(self):
            pass
    monkeypatch.setattr(utils._DEVICE_POOL, "acquire", lambda nbytes, device: FakeBuf())
    monkeypatch.setattr(utils._DEVICE_POOL, "release", lambda buf: None)
    utils._H5_KEEP.clear()
    with pytest.warns(UserWarning, match="host reader"):
        assert utils.load_batch_device(path, fmap, device=0) is None
    utils._H5_KEEP.clear()
