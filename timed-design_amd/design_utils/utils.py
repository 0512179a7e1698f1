"""design_utils.utils — the dataset / codec / writer functions either side of the CNN kernels.

Same names, argument meaning and output formats as the reference's ``design_utils/utils.py`` for the
functions on the hot path; each function cites the lines it mirrors.
Written from scratch: HDF5 access goes through h5py when importable and the bundled pure-Python
reader otherwise, amino-acid tables are local (design_utils/amino_acids.py), and the per-residue
Python loops of the reference are vectorised where that cannot change the output.

Out of scope here (UI helpers of the reference module): PDB property editing, BLOSUM lookup,
alphanumeric map codes, rm_tree.
"""
from __future__ import annotations

import contextlib
import os
import threading
import typing as t
import warnings
from collections.abc import Mapping
from itertools import product
from pathlib import Path

import numpy as np

from timed_hip import textio

from .amino_acids import UNCOMMON_RESIDUE_DICT, side_chain_dihedrals, standard_amino_acids


# ---- dataset access -----------------------------------------------------------------------------------
def open_frame_dataset(path):
    """h5py.File when h5py is importable, timed_hip.h5lite.File otherwise (same read surface)."""
    try:
        import h5py  # type: ignore
        return h5py.File(str(path), "r")
    except ImportError:
        from timed_hip import h5lite
        return h5lite.File(str(path))


def _have_h5py() -> bool:
    try:
        import h5py  # type: ignore  # noqa: F401
        return True
    except ImportError:
        return False


def _as_str(v) -> str:
    if isinstance(v, bytes):
        return v.decode()
    if isinstance(v, np.ndarray) and v.shape == ():
        return _as_str(v[()])
    return str(v)


# ---- dataset map ----------------------------------------------------------------------------------------
# dataset-map flavours: (column delimiter, header lines to skip)
_MAP_FORMATS = {True: (",", 0),     # old maps: csv rows (pdb, chain, residue_id, label)
                False: (" ", 3)}    # PDBench maps: three header lines, then "<pdb> <count>"


def load_datasetmap(path_to_datasetmap: Path, is_old: bool = False) -> np.ndarray:
    """reference utils.py:190-227: rows of the map as strings.  A map with a single row parses to a 1-D array; it is
    wrapped in a list so that callers can still iterate rows (reference :223-225)."""
    path_to_datasetmap = Path(path_to_datasetmap)
    assert (
        path_to_datasetmap.suffix == ".txt"
    ), f"Expected Path {path_to_datasetmap} to be a .txt file but got {path_to_datasetmap.suffix}."
    delimiter, header_lines = _MAP_FORMATS[bool(is_old)]
    rows = np.asarray(np.genfromtxt(path_to_datasetmap, delimiter=delimiter, dtype=str, skip_header=header_lines))
    return [rows] if rows.ndim == 1 else rows


def get_pdb_keys_to_filter(pdb_key_path: Path, file_extension: str = ".txt") -> t.List[str]:
    """reference utils.py:284-315: first four characters of every key in every list file."""
    pdb_key_files = list(Path(pdb_key_path).glob(f"**/*{file_extension}"))
    assert len(pdb_key_files) >= 1, "Expected at least 1 pdb key file."
    keys: t.List[str] = []
    for fpath in pdb_key_files:
        for pdb in np.atleast_1d(np.genfromtxt(fpath, dtype=str)):
            keys.append(str(pdb)[:4])
    return keys


def create_flat_dataset_map(
    frame_dataset: Path,
    filter_list: t.List[str] = [],
    remove_blacklist_silently: bool = False,
    uncommon_residue_dict: t.Optional[dict] = None,
) -> (t.List[t.Tuple[str, str, str, str]], t.Set[str]):
    """reference utils.py:318-407.  Flattens pdb/chain/residue into the deterministic frame order the
    whole pipeline relies on: pdb groups in HDF5 name order, chains in name order, residues sorted
    NUMERICALLY (the reference's ``np.int`` at :368 only runs on NumPy < 1.24; plain ``int`` here).
    Uncommon residue labels are mapped through aposteriori's table (:381-385)."""
    from timed_hip import framepack
    if framepack.is_pack(frame_dataset):  # packed dataset: the map was fixed when the pack was written
        fmap = [tuple(r) for r in _frame_pack(frame_dataset).flat_map.tolist()]     # the cached pack load_batch will use
        return fmap, {r[0] for r in fmap}
    if framepack.is_structure(frame_dataset):   # a PDB file: voxelised on the GPU (row f-4), one frame per residue
        fmap = [tuple(str(x) for x in r) for r in _structure_pack(frame_dataset).flat_map if r[0][:4] not in filter_list]
        return fmap, {r[0] for r in fmap}
    uncommon = UNCOMMON_RESIDUE_DICT if uncommon_residue_dict is None else uncommon_residue_dict
    kept = _kept_h5lite(frame_dataset)      # stays mapped, link tables parsed once: load_batch_device reads the same handle
    if kept is not None:
        from timed_hip import h5lite
        try:
            return _flat_map_of(kept, filter_list, remove_blacklist_silently, uncommon)
        except (h5lite.H5Unsupported, h5lite.H5FormatError):
            if not _have_h5py():         # a file feature h5lite does not read: h5py's job, when it is there
                raise
    with open_frame_dataset(frame_dataset) as dataset_file:
        return _flat_map_of(dataset_file, filter_list, remove_blacklist_silently, uncommon)


def _flat_map_of(dataset_file, filter_list, remove_blacklist_silently, uncommon):
    """create_flat_dataset_map over an open dataset (h5py.File or h5lite.File)"""
    standard_residues = list(standard_amino_acids.values())
    training_set_pdbs = set()
    flat_dataset_map = []
    chains = []
    if True:
        for pdb_code in dataset_file:
            if pdb_code[:4] in filter_list:
                if remove_blacklist_silently:
                    warnings.warn(f"PDB code {pdb_code} was found in benchmark dataset. It was automatically removed.")
                    continue
                raise ValueError(
                    f"PDB code {pdb_code} was found in benchmark dataset. "
                    f"Turn on remove_blacklist_silently=True if you want to ignore these structures for training.")
            pdb_group = dataset_file[pdb_code]
            for chain_id in pdb_group.keys():
                chain_group = pdb_group[chain_id]
                residue_n = np.array(list(chain_group.keys()), dtype=int)
                residue_n.sort()
                chains.append((pdb_code, chain_id, chain_group, residue_n))
        # the `label` attribute of every residue (utils.py:375): h5lite files resolve ALL object headers of the file in one
        # threaded native call; anything else reads attribute by attribute as the reference does
        all_labels = _all_chain_labels(dataset_file, chains)
        at = 0
        for pdb_code, chain_id, chain_group, residue_n in chains:
            labels = all_labels[at:at + len(residue_n)] if all_labels is not None else None
            at += len(residue_n)
            for k, residue_id in enumerate(residue_n.astype(str).tolist()):
                residue_label = labels[k] if labels is not None else _as_str(chain_group[residue_id].attrs["label"])
                if residue_label not in standard_residues:
                    if residue_label in uncommon:
                        warnings.warn(f"{residue_label} is not a standard residue.")
                        residue_label = uncommon[residue_label]
                        warnings.warn(f"Residue converted to {residue_label}.")
                    else:
                        raise AssertionError(f"Expected natural amino acid, but got {residue_label}.")
                flat_dataset_map.append((pdb_code, chain_id, residue_id, residue_label))
                training_set_pdbs.add(pdb_code)
    return flat_dataset_map, training_set_pdbs


def read_flat_dataset_map(dataset_map_path) -> np.ndarray:
    """``np.genfromtxt(dataset_map_path, delimiter=",", dtype="str")`` of reference predict.py:99 as a 2-D array: plain
    files through the native tokenizer (th_csv_shape / th_csv_fill: 100 k rows in milliseconds instead of 0.2 s),
    anything else through NumPy itself."""
    table = textio.read_string_table(dataset_map_path, ",")
    if table is None:
        table = np.atleast_2d(np.genfromtxt(dataset_map_path, delimiter=",", dtype="str"))
    return table


def flat_dataset_map_array(frame_dataset: Path, filter_list: t.Sequence[str] = ()) -> np.ndarray:
    """create_flat_dataset_map as a 2-D string array (what predict.py slices): a frame pack's map is handed out as the
    array the pack was loaded with — no detour through 100 k tuples."""
    from timed_hip import framepack
    if framepack.is_pack(frame_dataset) and not filter_list:
        return _frame_pack(frame_dataset).flat_map
    flat, _ = create_flat_dataset_map(frame_dataset, list(filter_list))
    return np.array(flat)


def _all_chain_labels(dataset_file, chains):
    """`label` attribute of every residue dataset of every chain in ONE native call (h5lite files only; None otherwise:
    the caller then reads attribute by attribute)."""
    if not chains or not all(_is_h5lite_group(c[2]) for c in chains):
        return None
    from timed_hip import h5lite
    addrs = []
    try:
        for _pdb, _chain, chain_group, residue_n in chains:
            links = chain_group._load()
            addrs.extend(links[r] for r in residue_n.astype(str).tolist())
    except KeyError:        # a residue name that is not its own canonical integer ("007"): the general path handles it
        return None
    res = h5lite.resolve_many(dataset_file, addrs, str_attr="label", str_len=16)
    if res is None or not np.all(res["status"] & 4):
        return None
    return res["strs"]


def _is_h5lite_group(obj) -> bool:
    from timed_hip import h5lite
    return isinstance(obj, h5lite.Group)


def _load_batch_native(dataset, data_point_batch, X, y, as_float32: bool):
    """The whole batch through two native calls (th_h5_resolve: object headers + `encoded_residue` rows;
    th_h5_read_chunked_as: chunk B-trees, inflate, placement on host threads).  Returns a boolean mask of the rows
    that were filled; the rest (anything unusual about a dataset) go through the general reader."""
    from timed_hip import h5lite
    n = len(data_point_batch)
    done = np.zeros(n, dtype=bool)
    addrs = np.full(n, -1, dtype=np.int64)
    chain_links = {}
    for i, (pdb_code, chain_id, residue_id, *_rest) in enumerate(data_point_batch):
        key = (str(pdb_code), str(chain_id))
        links = chain_links.get(key)
        if links is None:
            try:
                grp = dataset[key[0]][key[1]]
            except KeyError:
                return done
            links = chain_links[key] = grp._load() if isinstance(grp, h5lite.Group) else {}
        addrs[i] = links.get(str(residue_id), -1)
    if np.any(addrs < 0):
        return done
    res = h5lite.resolve_many(dataset, addrs, num_attr="encoded_residue", num_len=y.shape[1])
    if res is None:
        return done
    ok = (res["status"] & 3) == 3
    g = res["geom"]
    rank = int(g[0])
    want_kind = {0: "iu", 1: "f", 8: "b"}.get(int(g[16]), "")
    if as_float32:
        fits = X.dtype == np.float32 and int(g[16]) == 1 and int(g[15]) == 8
    else:
        fits = X.dtype.itemsize == int(g[15]) and (X.dtype.kind in want_kind or (X.dtype == bool and int(g[16]) in (0, 8)))
    if not (fits and tuple(int(v) for v in g[1:1 + rank]) == tuple(X.shape[1:]) and ok.any()):
        return done
    rows = np.nonzero(ok)[0]
    if h5lite.read_resolved(dataset, res, rows, [X[i] for i in rows], as_float32=as_float32):
        y[rows] = res["num"][rows]
        done[rows] = True
    return done


def load_batch(dataset_path: Path, data_point_batch: t.List[t.Tuple], dtype=None, out=None) -> (np.ndarray, np.ndarray):
    """reference utils.py:487-530: X[batch, *frame_dims] (float64 when voxels_as_gaussian, else bool —
    :518-521) and y[batch, 20] one-hot labels from the ``encoded_residue`` attribute.

    ``dtype=np.float32`` (opt-in, used by predict.py's pipeline) returns Gaussian frames as float32: the float64 ->
    float32 rounding Keras applies to the batch anyway happens while the chunks are placed, and half
    as many bytes cross PCIe.  The default keeps the reference's dtypes.  ``out`` (opt-in): a preallocated array of at
    least ``len(data_point_batch)`` frames of the right shape and dtype — e.g. page-locked memory from
    ``timed_hip.engine.pinned_empty`` — that receives the frames instead of a fresh allocation (a fresh 0.5 GB batch
    costs more in first-touch page faults than its inflation does); ignored when it does not fit."""
    from timed_hip import framepack
    if framepack.is_pack(dataset_path):  # HDF5-free fast path (SURVEY f-1): memory-mapped rows, no per-residue reads
        return _frame_pack(dataset_path).load_batch(data_point_batch)
    if framepack.is_structure(dataset_path):
        return _structure_pack(dataset_path).load_batch(data_point_batch)
    # the native readers (th_h5_resolve + th_h5_read_chunked_as on host threads) work on the kept h5lite mapping of the file,
    # whether or not h5py is installed; what h5lite cannot read at all is h5py's job when it is there
    kept = _kept_h5lite(dataset_path)
    if kept is not None:
        from timed_hip import h5lite
        try:
            return _load_batch_from(kept, data_point_batch, dtype, out)
        except (h5lite.H5Unsupported, h5lite.H5FormatError):
            if not _have_h5py():
                raise
    with open_frame_dataset(dataset_path) as dataset:
        return _load_batch_from(dataset, data_point_batch, dtype, out)


def _load_batch_from(dataset, data_point_batch, dtype, out):
    """load_batch over an open dataset (h5py.File or h5lite.File)"""
    batch_size = len(data_point_batch)
    if True:
        dims = tuple(int(d) for d in np.asarray(dataset.attrs["frame_dims"]).ravel())
        voxels_as_gaussian = bool(dataset.attrs["voxels_as_gaussian"])
        as_f32 = dtype is not None and np.dtype(dtype) == np.float32 and voxels_as_gaussian
        x_dtype = np.dtype((np.float32 if as_f32 else float) if voxels_as_gaussian else bool)
        if (out is not None and out.dtype == x_dtype and out.ndim == 1 + len(dims) and tuple(out.shape[1:]) == dims
                and out.shape[0] >= batch_size and out.flags.c_contiguous):
            X = out[:batch_size]
        else:
            X = np.empty((batch_size, *dims), dtype=x_dtype)
        y = np.zeros((batch_size, 20), dtype=float)
        if _is_h5lite_file(dataset) and batch_size:
            filled = _load_batch_native(dataset, data_point_batch, X, y, as_f32)
            if filled.all():
                return X, y
        else:
            filled = np.zeros(batch_size, dtype=bool)
        todo = np.nonzero(~filled)[0]
        if len(todo) < batch_size:      # the general reader for the few datasets the native pass declined
            for i in todo:
                pdb_code, chain_id, residue_id = (str(v) for v in data_point_batch[i][:3])
                ds = dataset[pdb_code][chain_id][residue_id]
                X[i] = np.asarray(ds[()])
                y[i] = np.asarray(ds.attrs["encoded_residue"])
            return X, y
        frames = []
        for i, (pdb_code, chain_id, residue_id, _) in enumerate(data_point_batch):
            ds = dataset[str(pdb_code)][str(chain_id)][str(residue_id)]
            frames.append(ds)
            y[i] = np.asarray(ds.attrs["encoded_residue"])

        if frames and _is_h5lite(frames[0]):
            # one native call walks every residue's chunk B-tree, gunzips and places the chunks straight into X on host
            # threads (libtimedhip th_h5_read_chunked); anything it cannot place takes the generic path
            from timed_hip import h5lite
            filled = h5lite.read_many_direct(frames, [X[i] for i in range(batch_size)])
        else:
            filled = [False] * batch_size
        for i in range(batch_size):
            if not filled[i]:
                X[i] = np.asarray(frames[i][()])
    return X, y


class _DevicePool:
    """A few reusable device buffers for batches decoded on the GPU (a fresh hipMalloc / hipFree per batch costs more than
    the decode of a small batch)."""

    def __init__(self, keep: int = 6):
        import threading
        self.free: t.List = []
        self.keep = keep
        self.lock = threading.Lock()

    def acquire(self, nbytes: int, device: int):
        from timed_hip import engine
        with self.lock:
            for i, b in enumerate(self.free):
                if b.device == device and b.nbytes >= nbytes:
                    return self.free.pop(i)
        return engine.DeviceBuffer(nbytes, device)

    def release(self, buf):
        with self.lock:
            if len(self.free) < self.keep:
                self.free.append(buf)
                return
        buf.free()


    def drain(self):
        with self.lock:
            bufs, self.free = self.free, []
        for b in bufs:
            b.free()


_DEVICE_POOL = _DevicePool()


# Staging rings (engine.StagingRing: page-locked slots between a mapped frame pack and the GPU) kept between predict calls: building
# one costs ~50 ms (first-touch faults + page-locking 5 x 227 MB), taking it down 45 ms.  A run takes a ring out of the cache and
# puts it back when it is done, so two runs never share slots.
_STAGING_RINGS: dict = {}
_STAGING_LOCK = __import__("threading").Lock()


def take_staging_ring(slots: int, rows: int, frame_shape, dtype, threads: int):
    from timed_hip import engine
    key = (int(slots), int(rows), tuple(int(d) for d in frame_shape), np.dtype(dtype).str, int(threads))
    with _STAGING_LOCK:
        ring = _STAGING_RINGS.pop(key, None)
    if ring is None:
        ring = engine.StagingRing(slots, rows, frame_shape, dtype, threads=threads)
        ring.key = key
    return ring


def give_back_staging_ring(ring) -> None:
    with _STAGING_LOCK:
        if ring.enabled and ring.key not in _STAGING_RINGS:
            _STAGING_RINGS[ring.key] = ring
            return
    ring.close()


def release_device_memory(devices=None) -> None:
    """Give back what load_batch_device keeps between calls so that the next call is fast: the pooled batch buffers (up to six of
    frames_per_call frames each), the page-locked staging ring of frame-pack runs, the GPU decoder's scratch (token arena, ~5 bytes per uncompressed byte of a batch) and the device
    blocks of closed models (weights / arenas kept for the next load).  predict.py's
    CLI calls it when its run ends; a long-lived process (the UI) calls it when it is done predicting.  No reference counterpart
    (the reference holds no device memory)."""
    from timed_hip import _lib
    _DEVICE_POOL.drain()
    with _STAGING_LOCK:
        rings = list(_STAGING_RINGS.values())
        _STAGING_RINGS.clear()
    for ring in rings:
        ring.close()
    lib = _lib.load()
    lib.th_format_csv_device_release()      # the text formatter's stream and buffers (predict.py --predict_rotamers)
    for d in (range(16) if devices is None else devices):
        lib.th_h5_release_scratch(int(d))
        lib.th_dev_trim(int(d))              # device blocks of closed models, kept for the next load
_H5_KEEP: dict = {}      # path -> (mtime, size, h5lite.File): the dataset load_batch_device read last stays mapped


def _kept_h5lite(dataset_path):
    """The reference re-opens the dataset for every batch (utils.py:514); mapping and unmapping a multi-GB file costs ~7 ms
    per call (munmap), so the native paths (dataset map, GPU decode) keep the file they read last open through timed_hip.h5lite
    (re-opened when the file changed on disk) — also when h5py is installed: h5py then still serves the general reader.
    None when h5lite cannot open the file (not HDF5, a superblock version it does not read ...)."""
    from timed_hip import h5lite
    key = os.path.abspath(os.fspath(dataset_path))
    try:
        st = os.stat(key)
    except OSError:
        return None
    kept = _H5_KEEP.get(key)
    if kept is not None and kept[0] == st.st_mtime_ns and kept[1] == st.st_size:
        return kept[2]
    for _k, (_m, _s, f) in list(_H5_KEEP.items()):
        try:
            f.close()
        except Exception:
            pass
    _H5_KEEP.clear()
    try:
        f = h5lite.File(key)
    except (h5lite.H5Unsupported, h5lite.H5FormatError, OSError, ValueError):
        return None
    _H5_KEEP[key] = (st.st_mtime_ns, st.st_size, f)
    return f


def load_batch_device(dataset_path: Path, data_point_batch, device: int = 0):
    """load_batch (reference utils.py:487-530) with the frames left in HBM: (DeviceFrames [n, *frame_dims] float32 — or uint8
    for boolean datasets —, y[n, 20]) — or None when this dataset cannot take the path (not an h5lite-readable .hdf5, a filter
    pipeline other than deflate, mixed geometries ...; the caller then uses load_batch).  The gzip chunks are inflated ON THE
    GPU (libtimedhip th_h5_decode_device, one lane per chunk): only the compressed bytes cross PCIe — 8.7x fewer than the float64
    frames — and no host core spends a millisecond per frame in zlib."""
    from timed_hip import _lib, engine, framepack, h5lite
    if framepack.is_pack(dataset_path) or framepack.is_structure(dataset_path):
        return None
    n = len(data_point_batch)
    if n == 0:
        return None
    dataset = _kept_h5lite(dataset_path)
    if dataset is None:
        return None
    try:
        return _load_batch_device(dataset, data_point_batch, n, device)
    except (h5lite.H5Unsupported, h5lite.H5FormatError, KeyError):      # a file feature h5lite does not read: the host reader's job
        return None
    except _lib.TimedHipError as e:
        # no device memory left for the decoder (TH_ENOMEM, after decode_resolved_device already tried smaller pieces) or a HIP call
        # failed: the host reader does not need the device at all — use it instead of aborting the run.  A corrupt chunk (TH_EIO) or
        # a bad argument stays an error: the host reader would refuse the same file.
        if e.code not in (_lib.TH_ENOMEM, _lib.TH_EHIP):
            raise
        warnings.warn(f"GPU decode of {n} frames failed ({e}); reading this batch through the host reader")
        return None


_LOAD_TRACE = [0.0, 0.0, 0.0, 0.0]       # TIMED_PIPELINE_TRACE: seconds in (group links, resolve_many, buffer, decode) of load_batch_device


def _load_batch_device(dataset, data_point_batch, n, device):
    import time
    from timed_hip import engine, h5lite
    t_0 = time.perf_counter()
    if True:
        dims = tuple(int(d) for d in np.asarray(dataset.attrs["frame_dims"]).ravel())
        gaussian = bool(dataset.attrs["voxels_as_gaussian"])
        addrs = np.full(n, -1, dtype=np.int64)
        chain_links = {}
        for i, (pdb_code, chain_id, residue_id, *_rest) in enumerate(data_point_batch):
            key = (str(pdb_code), str(chain_id))
            links = chain_links.get(key)
            if links is None:
                try:
                    grp = dataset[key[0]][key[1]]
                except KeyError:
                    return None
                links = chain_links[key] = grp._load() if isinstance(grp, h5lite.Group) else {}
            addrs[i] = links.get(str(residue_id), -1)
        if np.any(addrs < 0):
            return None
        t_1 = time.perf_counter()
        res = h5lite.resolve_many(dataset, addrs, num_attr="encoded_residue", num_len=20)
        t_2 = time.perf_counter()
        if res is None or not np.all((res["status"] & 3) == 3):
            return None
        g = res["geom"]
        rank = int(g[0])
        if tuple(int(v) for v in g[1:1 + rank]) != dims:
            return None
        cls, esz = int(g[16]), int(g[15])
        if gaussian and cls == 1 and esz == 8:
            dtype, as_f32 = np.float32, True
        elif gaussian and cls == 1 and esz == 4:          # frames stored as float32: placed as they are
            dtype, as_f32 = np.float32, False
        elif not gaussian and esz == 1 and cls in (0, 8):
            dtype, as_f32 = np.uint8, False
        else:
            return None
        nbytes = n * int(np.prod(dims)) * np.dtype(dtype).itemsize
        t_3 = time.perf_counter()
        buf = _DEVICE_POOL.acquire(nbytes, device)
        t_4 = time.perf_counter()
        ok = False
        try:
            ok = h5lite.decode_resolved_device(dataset, res, buf.ptr, device, as_float32=as_f32)
        finally:
            if not ok:
                _DEVICE_POOL.release(buf)
        t_5 = time.perf_counter()
        for i, dt in enumerate((t_1 - t_0, t_2 - t_1, t_4 - t_3, t_5 - t_4)):
            _LOAD_TRACE[i] += dt
        if not ok:
            return None
        return engine.DeviceFrames(buf, (n, *dims), dtype, on_release=_DEVICE_POOL.release), np.asarray(res["num"], dtype=float)


def _is_h5lite(obj) -> bool:
    from timed_hip import h5lite
    return isinstance(obj, h5lite.Dataset)


def _is_h5lite_file(obj) -> bool:
    from timed_hip import h5lite
    return isinstance(obj, h5lite.File)


_PACKS: dict = {}
_STRUCTURES: dict = {}


def _structure_pack(path):
    from timed_hip import framepack
    key = (os.path.abspath(os.fspath(path)), os.path.getmtime(path))
    if key not in _STRUCTURES:
        _STRUCTURES.clear()            # one structure's frames at a time
        _STRUCTURES[key] = framepack.StructurePack(path)
    return _STRUCTURES[key]



def _frame_pack(path):
    from timed_hip import framepack
    key = framepack.pack_stem(path)
    if key not in _PACKS:
        _PACKS[key] = framepack.FramePack(path)
    return _PACKS[key]


# ---- rotamer codec ------------------------------------------------------------------------------------
def _rotamer_suffixes(n_chi: int) -> t.List[str]:
    """the 3**n_chi rotamer labels of a residue with n_chi side-chain dihedrals, each angle in bin 1, 2 or 3, first angle
    slowest ("11", "12", "13", "21", ...); a residue without side-chain dihedrals has the single label "0" """
    return ["".join(bins) for bins in product("123", repeat=n_chi)] if n_chi else ["0"]


def get_rotamer_codec(return_reduction_guide: bool = False):
    """reference utils.py:410-465.  The 338 rotamer categories: residues in alphabetical one-letter order, every residue
    contributing its rotamer labels ("ARG_1123" ...; "ALA_0" / "GLY_0" for residues without side-chain dihedrals).
    Returns ({rotamer_index: one-hot(20) of its residue}, [338 names]) and optionally the index of each residue's first
    category ([0, 1, 4, 13, 40, ...] — the guide printed at reference :425)."""
    residues = list(standard_amino_acids.values())
    per_residue = [_rotamer_suffixes(len(side_chain_dihedrals.get(res, ()))) for res in residues]
    flat_categories = [f"{res}_{suffix}" for res, suffixes in zip(residues, per_residue) for suffix in suffixes]
    owner = np.repeat(np.arange(len(residues)), [len(sfx) for sfx in per_residue])       # category -> residue index
    identity = np.eye(len(residues), dtype=int)
    rot_to_20res = {k: identity[r].copy() for k, r in enumerate(owner)}
    if return_reduction_guide:
        reduction_guide = np.concatenate([[0], np.cumsum([len(sfx) for sfx in per_residue])[:-1]]).tolist()
        return rot_to_20res, flat_categories, reduction_guide
    return rot_to_20res, flat_categories


def compress_rotamer_predictions_to_20(prediction_matrix: np.ndarray) -> np.ndarray:
    """reference utils.py:468-484: (n, 338) -> (n, 20) by summing each residue's rotamer columns."""
    _, _, reduction_guide = get_rotamer_codec(return_reduction_guide=True)
    return np.add.reduceat(prediction_matrix, reduction_guide, axis=1)


# ---- prediction matrix -> sequences ----------------------------------------------------------------------
def _column_letters(rotamers_categories) -> np.ndarray:
    """one-letter residue code of every probability column: the 20 residues in one-letter order, or — for a rotamer
    matrix — the residue of each rotamer category ("ARG_1123" -> "R"; categories already given as letters pass through)"""
    one_letter = {three: one for one, three in standard_amino_acids.items()}
    if not rotamers_categories:
        return np.array(list(standard_amino_acids.keys()))
    if len(rotamers_categories[0]) == 1:
        return np.array(list(rotamers_categories))
    return np.array([one_letter[name.split("_")[0]] for name in rotamers_categories])


class SequencePlan:
    """Everything extract_sequence_from_pred_matrix derives from the dataset map ALONE: the keys in first-seen order,
    the matrix rows of every key (map order) and the true sequences.  predict.py builds it on a side thread while the
    GPU works, so that only the arg-max and the string joins remain once the probabilities exist."""

    def __init__(self, flat_dataset_map):
        fmap = np.asarray(flat_dataset_map)
        if fmap.ndim != 2:
            fmap = np.atleast_2d(fmap)
        self.old_datasetmap = fmap.shape[1] == 4                     # re-derived from the map width (reference :662)
        self.n_rows = 0
        self.keys: t.List[str] = []
        self.runs: t.Dict[str, t.List[t.Tuple[int, int]]] = {}       # key -> [(lo, hi), ...] matrix row ranges, in order
        self.real: t.Dict[str, str] = {}
        if self.old_datasetmap:
            n = fmap.shape[0]
            self.n_rows = n
            pdb, chain = fmap[:, 0], fmap[:, 1]
            # consecutive rows of one pdb + chain are one run; a key that comes back later continues its entry
            change = np.flatnonzero((pdb[1:] != pdb[:-1]) | (chain[1:] != chain[:-1])) + 1
            starts = np.concatenate([[0], change]).astype(np.int64)
            ends = np.concatenate([change, [n]]).astype(np.int64)
            run_keys = np.char.add(pdb[starts].astype(str), chain[starts].astype(str))
            one_letter = {three: one for one, three in standard_amino_acids.items()}
            uniq, inverse = np.unique(fmap[:, 3], return_inverse=True)
            real_letters = np.array([one_letter[str(u)] for u in uniq], dtype="S1")[inverse] if n else np.empty(0, "S1")
            real_parts: t.Dict[str, t.List[bytes]] = {}
            for key, lo, hi in zip(run_keys.tolist(), starts.tolist(), ends.tolist()):
                self.runs.setdefault(key, []).append((lo, hi))
                real_parts.setdefault(key, []).append(real_letters[lo:hi].tobytes())
            self.keys = list(self.runs)
            self.real = {k: b"".join(v).decode("ascii") for k, v in real_parts.items()}
        else:
            cursor = 0
            for key, count in fmap:
                key, count = str(key), int(count)
                self.runs.setdefault(key, []).append((cursor, cursor + count))
                cursor += count
            self.n_rows = cursor
            self.keys = list(self.runs)
            self.real = {k: "" for k in self.keys}                   # "<pdb> <count>" maps carry no true sequence

    def rows(self, key) -> t.Union[slice, np.ndarray]:
        runs = self.runs[key]
        if len(runs) == 1:
            return slice(runs[0][0], runs[0][1])
        return np.concatenate([np.arange(lo, hi) for lo, hi in runs])


class LazyProbabilities(Mapping):
    """``pdb_to_probability`` of the reference — {key: [list(row), ...]} (utils.py:646,688-690) — with the lists built on
    first access of a key: a 100 k-residue run holds 2 M Python floats that predict.py's caller rarely looks at.
    ``matrix(key)`` gives the same rows as an array without the list round trip (sample.py uses it)."""

    def __init__(self, prediction_matrix: np.ndarray, plan: SequencePlan):
        self._matrix, self._plan, self._lists = prediction_matrix, plan, {}

    def matrix(self, key) -> np.ndarray:
        return self._matrix[self._plan.rows(key)]

    def __getitem__(self, key):
        if key not in self._lists:
            if key not in self._plan.runs:
                raise KeyError(key)
            self._lists[key] = [list(r) for r in self.matrix(key)]
        return self._lists[key]

    def __iter__(self):
        return iter(self._plan.keys)

    def __len__(self):
        return len(self._plan.keys)

    def __contains__(self, key):
        return key in self._plan.runs


def extract_sequence_from_pred_matrix(
    flat_dataset_map: t.List[t.Tuple],
    prediction_matrix: np.ndarray,
    rotamers_categories: t.Optional[t.List[str]],
    old_datasetmap: bool = False,
    is_consensus: bool = False,
    plan: t.Optional[SequencePlan] = None,
) -> (dict, dict, dict, dict, dict):
    """reference utils.py:616-723.  argmax (first maximum) -> one-letter sequence per key; key = pdb+chain for 4-column
    maps, the map's own key for "<pdb> <count>" maps (which carry no true sequence: it stays "").  ``old_datasetmap`` is
    re-derived from the map width exactly as the reference does (:662).  With ``is_consensus`` the states
    "<pdb>_<n>..." of an NMR ensemble are merged by the reference's RUNNING PAIRWISE average (acc+new)/2 (:699-705) —
    not an arithmetic mean.

    The arg-max and the residue letters come from one native pass (th_argmax_letters); ``pdb_to_probability`` is a
    Mapping whose list-of-lists values are built when a key is first read (LazyProbabilities).  ``plan`` (opt-in): a
    SequencePlan of the same map prepared earlier."""
    letters = _column_letters(rotamers_categories)
    prediction_matrix = np.asarray(prediction_matrix)
    if plan is None:
        plan = SequencePlan(flat_dataset_map)
    if prediction_matrix.ndim == 2 and prediction_matrix.dtype in (np.float16, np.float32, np.float64) and \
            all(len(str(c)) == 1 and ord(str(c)) < 128 for c in letters):
        predicted = textio.argmax_letters(prediction_matrix, letters)
        join = lambda rows: predicted[rows].tobytes().decode("ascii")      # noqa: E731
    else:       # multi-character column names or an unusual matrix dtype: NumPy
        chosen = letters[np.argmax(prediction_matrix, axis=1)]
        join = lambda rows: "".join(chosen[rows])                          # noqa: E731
    pdb_to_sequence = {key: join(plan.rows(key)) for key in plan.keys}
    pdb_to_probability = LazyProbabilities(prediction_matrix, plan)
    pdb_to_real_sequence = dict(plan.real)
    if not is_consensus:
        return pdb_to_sequence, pdb_to_probability, pdb_to_real_sequence, None, None
    # consecutive keys with the same prefix before the first "_" are states of one structure
    pdb_to_consensus_prob: dict = {}
    previous = None
    for key in pdb_to_sequence:
        structure = key.split("_")[0]
        state = np.array(pdb_to_probability.matrix(key))
        pdb_to_consensus_prob[structure] = state if structure != previous else (pdb_to_consensus_prob[structure] + state) / 2
        previous = structure
    pdb_to_consensus = {structure: "".join(letters[np.argmax(prob, axis=1)]) for structure, prob in pdb_to_consensus_prob.items()}
    return pdb_to_sequence, pdb_to_probability, pdb_to_real_sequence, pdb_to_consensus, pdb_to_consensus_prob


# ---- writers ---------------------------------------------------------------------------------------------
def save_outputs_to_file(
    y_true: np.ndarray,
    y_pred: t.Dict[int, list],
    flat_dataset_map: t.List[t.Tuple],
    model: int,
    model_name: str,
    path_to_output: Path = Path.cwd(),
):
    """reference utils.py:726-771.  APPENDS: encoded_labels.csv ('%i', first model only), datasetmap.txt
    (written once, only if absent), <model>.csv — probabilities cast to float16 then written with
    np.savetxt's default '%.18e' (so the file holds float16-rounded values)."""
    append_outputs(np.asarray(y_true), np.array(y_pred[model], dtype=np.float16), flat_dataset_map, model, model_name, path_to_output)


def append_outputs(y_true: np.ndarray, predictions_f16: np.ndarray, flat_dataset_map, model: int, model_name: str,
                   path_to_output: Path = Path.cwd(), opener=None, write_map: bool = True):
    """The file appends of save_outputs_to_file on arrays (no list round trip): ``y_true`` [n, 20] labels,
    ``predictions_f16`` [n, n_classes] float16 — what ``np.array(y_pred[model], dtype=np.float16)`` yields there.
    ``opener(path)`` returns the append handle of a file (default: the file itself, binary append); the sharded
    predict path passes an in-memory sink and writes the dataset map on rank 0 only (``write_map=False``)."""
    path_to_output = Path(path_to_output)
    opener = opener or (lambda p: open(p, "ab"))
    if model == 0:
        with opener(path_to_output / "encoded_labels.csv") as f:
            _savetxt_small_ints(f, np.asarray(y_true))
    path_to_datasetmap = path_to_output / "datasetmap.txt"
    if write_map and not path_to_datasetmap.exists():
        with open(path_to_datasetmap, "a") as f:
            _savetxt_strings(f, flat_dataset_map)
    with opener(path_to_output / f"{model_name}.csv") as f:
        textio.savetxt_csv(f, np.asarray(predictions_f16, dtype=np.float16))   # same bytes as np.savetxt(f, predictions, delimiter=",")


def _savetxt_strings(f, rows) -> None:
    """np.savetxt(f, np.asarray(rows), delimiter=",", fmt="%s") for a table of strings (the flat dataset map): '%s' of a
    NumPy string is the string itself, so every row is its fields joined by commas."""
    rows = np.asarray(rows)
    if rows.ndim == 2 and rows.dtype.kind in "US":
        f.write("".join([",".join(r) + "\n" for r in rows.tolist()]))
    else:
        np.savetxt(f, rows, delimiter=",", fmt="%s")


def _savetxt_small_ints(f, a: np.ndarray) -> None:
    """np.savetxt(f, a, delimiter=",", fmt="%i") for the one-hot label matrix: single digits are laid out as bytes
    directly; anything else goes through NumPy."""
    a = np.atleast_2d(np.asarray(a))
    if a.ndim == 2 and a.size and np.all(np.isfinite(a)):
        ai = a.astype(np.int64)          # '%i' truncates towards zero like this cast
        if ai.min() >= 0 and ai.max() <= 9:
            out = np.empty((ai.shape[0], 2 * ai.shape[1]), dtype=np.uint8)
            out[:, 0::2] = ai + ord("0")
            out[:, 1::2] = ord(",")
            out[:, -1] = ord("\n")
            f.write(out.tobytes())
            return
    np.savetxt(f, a, delimiter=",", fmt="%i")


def convert_dataset_map_for_srb(flat_dataset_map: list, model_name: str, path_to_output: Path = Path.cwd()):
    """reference utils.py:533-566: PDBench map "<pdb> <count>" (chain appended to 4-letter codes,
    "_0..." state suffix stripped) after three header lines."""
    count_dict: dict = {}
    fmap = np.asarray(flat_dataset_map)
    if fmap.ndim == 2 and fmap.shape[1] == 4 and fmap.shape[0] and fmap.dtype.kind == "U":
        # the same counts from runs of equal (pdb, chain): one dictionary update per run instead of per residue
        pdb, chain = fmap[:, 0], fmap[:, 1]
        change = np.flatnonzero((pdb[1:] != pdb[:-1]) | (chain[1:] != chain[:-1])) + 1
        starts = np.concatenate([[0], change])
        lengths = np.diff(np.concatenate([starts, [fmap.shape[0]]]))
        runs = zip(pdb[starts].tolist(), chain[starts].tolist(), lengths.tolist())
    else:
        runs = ((pdb, chain, 1) for pdb, chain, _res_idx, _ in flat_dataset_map)
    for pdb, chain, count in runs:
        pdb, chain = str(pdb), str(chain)
        if "_0" in pdb:
            pdb = pdb.split("_0")[0]
        if len(pdb) == 4:
            pdb += chain
        count_dict[pdb] = count_dict.get(pdb, 0) + count
    with open(Path(path_to_output) / f"{model_name}.txt", "w") as d:
        d.write("ignore_uncommon False\ninclude_pdbs\n##########\n")
        for pdb, count in count_dict.items():
            d.write(f"{pdb} {count}\n")


def save_dict_to_fasta(pdb_to_sequence: dict, model_name: str, path_to_output: Path = Path.cwd()):
    """reference utils.py:595-613."""
    with open(Path(path_to_output) / f"{model_name}.fasta", "w") as f:
        for pdb, seq in pdb_to_sequence.items():
            f.write(f">{pdb}\n{seq}\n")


def save_consensus_probs(pdb_to_consensus_prob: dict, model_name: str, path_to_output: Path = Path.cwd()):
    """reference utils.py:569-592.  (The reference writes the .csv to the CWD regardless of
    ``path_to_output`` — Appendix C-8; here both files go to ``path_to_output``.)"""
    path_to_output = Path(path_to_output)
    with open(path_to_output / f"{model_name}_consensus.txt", "w") as d, open(
        path_to_output / f"{model_name}_consensus.csv", "a"
    ) as p:
        d.write("ignore_uncommon False\ninclude_pdbs\n##########\n")
        for pdb, predictions in pdb_to_consensus_prob.items():
            d.write(f"{pdb} {len(predictions)}\n")
            np.savetxt(p, predictions, delimiter=",")
