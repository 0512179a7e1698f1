"""Packed frame datasets — the HDF5-free ingest path.

The reference reads one small HDF5 dataset per residue, re-opening the file for every batch
(reference design_utils/utils.py:514-529): a few hundred frames/s, three orders of magnitude below
what the GPU consumes.  A *frame pack* is the same information laid out for streaming:

    <stem>.frames.npy   [N, D, H, W, C]  uint8 (boolean datasets) or float32 (Gaussian datasets),
                        rows in flat-dataset-map order (create_flat_dataset_map, reference utils.py:362-393)
    <stem>.labels.npy   [N, 20] uint8 one-hot ``encoded_residue`` rows
    <stem>.map.txt      the flat dataset map as the reference writes it (csv: pdb,chain,residue,label)
    <stem>.meta.json    frame_dims, voxels_as_gaussian, source file, aposteriori version string

Sparse transport (round 6, optional, float32 packs): Gaussian frames are ~8 % non-zero, and a float32 pack fed the GPU at the PCIe
rate (222 KB per frame).  ``sparsify(stem)`` / ``pack_dataset(..., sparse=True)`` add

    <stem>.sparse.bits.npy    [N, W] uint32   bit k of word w set <=> element 32 w + k of the frame is stored (W = ceil(E / 32)
                                              rounded up to 4; every element whose BIT PATTERN is not +0.0 is stored: -0.0, NaN
                                              payloads and denormals survive, the round trip is bit-exact)
    <stem>.sparse.rank.npy    [N + 1] uint64  stored elements in front of frame i
    <stem>.sparse.values.npy  [total] float32 the stored elements, frame by frame, in element order

``FramePack.sparse_batch(lo, hi)`` hands a slice of them to the engine as a ``SparseFrames`` batch (``th_predict_sparse_async``:
the bitmap and the values cross PCIe — ~25 KB per frame — and the dense frame is rebuilt on the device in front of the first
layer); ``predict.py`` does so by itself when the files are there (``TIMED_SPARSE=0`` keeps the dense rows).  ``.frames.npy`` may
be deleted from a pack that has the sparse files (``batch`` then expands on the host).

``pack_dataset`` converts an aposteriori HDF5 file once (host side, any HDF5 reader); afterwards
``FramePack.batch(lo, hi)`` is a zero-copy memory-mapped slice that goes straight to ``th_predict``.
``design_utils.utils.load_batch`` / ``predict.load_dataset_and_predict`` accept a pack wherever they
accept a ``.hdf5`` path.  Values are bit-identical to what ``load_batch`` would return (uint8 0/1 vs
bool, float32 vs the float64 up-cast of float32 data — Keras casts to float32 either way).
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

SUFFIXES = (".frames.npy", ".labels.npy", ".map.txt", ".meta.json")
SPARSE_SUFFIXES = (".sparse.bits.npy", ".sparse.rank.npy", ".sparse.values.npy")


def _has_all(stem: str) -> bool:
    """labels, map and meta, plus the dense frames or the sparse triple"""
    if not all(os.path.exists(stem + s) for s in SUFFIXES[1:]):
        return False
    return os.path.exists(stem + SUFFIXES[0]) or all(os.path.exists(stem + s) for s in SPARSE_SUFFIXES)


def is_pack(path) -> bool:
    p = os.fspath(path)
    stem = pack_stem(p)
    return stem is not None and _has_all(stem)


def pack_stem(path) -> Optional[str]:
    p = os.fspath(path)
    for s in SUFFIXES + SPARSE_SUFFIXES:
        if p.endswith(s):
            return p[: -len(s)]
    if p.endswith(".framepack"):
        return p[: -len(".framepack")]
    if _has_all(p):
        return p
    return None


def sparsify(stem, rows_per_pass: int = 2048) -> Tuple[int, int]:
    """Write the sparse transport files of a float32 pack next to its dense frames (or replace older ones).  Returns
    (dense bytes, sparse bytes).  Vectorised NumPy, one pass over the memory-mapped frames."""
    stem = pack_stem(stem) or os.fspath(stem)
    frames = np.load(stem + ".frames.npy", mmap_mode="r")
    if frames.dtype != np.float32:
        raise ValueError(f"{stem}: sparse transport is for float32 (Gaussian) packs, this one holds {frames.dtype}")
    n = frames.shape[0]
    E = int(np.prod(frames.shape[1:]))
    W = ((E + 31) // 32 + 3) // 4 * 4
    counts = np.zeros(n, np.int64)
    for lo in range(0, n, rows_per_pass):            # pass 1: how many elements each frame stores
        x = np.asarray(frames[lo:lo + rows_per_pass]).reshape(-1, E).view(np.uint32)
        counts[lo:lo + len(x)] = np.count_nonzero(x, axis=1)
    rank = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    bits = np.lib.format.open_memmap(stem + ".sparse.bits.npy", mode="w+", dtype=np.uint32, shape=(n, W))
    values = np.lib.format.open_memmap(stem + ".sparse.values.npy", mode="w+", dtype=np.float32, shape=(max(int(rank[-1]), 1),))
    for lo in range(0, n, rows_per_pass):            # pass 2: bitmaps and values
        x = np.asarray(frames[lo:lo + rows_per_pass]).reshape(-1, E)
        mask = x.view(np.uint32) != 0
        padded = np.zeros((len(x), W * 32), bool)
        padded[:, :E] = mask
        bits[lo:lo + len(x)] = np.packbits(padded, axis=1, bitorder="little").view(np.uint32).reshape(len(x), W)
        values[int(rank[lo]):int(rank[lo + len(x)])] = x[mask]
    bits.flush(); values.flush()
    del bits, values
    np.save(stem + ".sparse.rank.npy", rank)
    return n * E * 4, n * W * 4 + (n + 1) * 8 + int(rank[-1]) * 4


def pack_dataset(hdf5_path, out_stem, filter_list: Sequence[str] = (), remove_blacklist_silently: bool = False,
                 progress_every: int = 0, sparse: bool = False) -> "FramePack":
    """Convert an aposteriori frame dataset (HDF5) into a frame pack.  One sequential pass.  ``sparse=True`` also writes the
    sparse transport files of a Gaussian (float32) dataset."""
    from design_utils import utils  # local import: design_utils imports timed_hip lazily too
    out_stem = os.fspath(out_stem)
    flat_map, _ = utils.create_flat_dataset_map(hdf5_path, list(filter_list), remove_blacklist_silently)
    n = len(flat_map)
    with utils.open_frame_dataset(hdf5_path) as ds:
        dims = tuple(int(d) for d in np.asarray(ds.attrs["frame_dims"]).ravel())
        gaussian = bool(ds.attrs["voxels_as_gaussian"])
        ver = str(ds.attrs["make_frame_dataset_ver"]) if "make_frame_dataset_ver" in ds.attrs else ""
        dtype = np.float32 if gaussian else np.uint8
        frames = np.lib.format.open_memmap(out_stem + ".frames.npy", mode="w+", dtype=dtype, shape=(n, *dims))
        labels = np.zeros((n, 20), dtype=np.uint8)
        for i, (pdb, chain, res, _label) in enumerate(flat_map):
            d = ds[pdb][chain][res]
            fr = np.asarray(d[()])
            if gaussian and fr.dtype != np.float32 and not np.array_equal(fr.astype(np.float32).astype(fr.dtype), fr):
                raise ValueError(f"{pdb}/{chain}/{res}: frame values are not exactly representable in float32")
            frames[i] = fr
            labels[i] = np.asarray(d.attrs["encoded_residue"]).astype(np.uint8)
            if progress_every and (i + 1) % progress_every == 0:
                print(f"packed {i + 1}/{n} frames")
        frames.flush()
        del frames
    np.save(out_stem + ".labels.npy", labels)
    np.savetxt(out_stem + ".map.txt", np.asarray(flat_map), delimiter=",", fmt="%s")
    with open(out_stem + ".meta.json", "w") as f:
        json.dump(dict(frame_dims=list(dims), voxels_as_gaussian=gaussian, n_frames=n, source=os.path.basename(os.fspath(hdf5_path)),
                       make_frame_dataset_ver=ver), f)
    if sparse and gaussian:
        sparsify(out_stem)
    return FramePack(out_stem)


def _read_map(path) -> np.ndarray:
    """the pack's own map file (written by pack_dataset with '%s' fields, no quoting, no comments): plain split; anything
    irregular goes through NumPy's general reader"""
    from . import textio
    table = textio.read_string_table(path, ",")          # native tokenizer: milliseconds for 100 k rows
    if table is not None and table.shape[1] == 4:
        return table
    with open(path) as f:
        rows = [line.split(",") for line in f.read().splitlines() if line]
    if rows and all(len(r) == 4 for r in rows):
        return np.array(rows, dtype=str)
    return np.atleast_2d(np.genfromtxt(path, delimiter=",", dtype=str))


class FramePack:
    def __init__(self, path):
        stem = pack_stem(path)
        if stem is None or not _has_all(stem):
            raise FileNotFoundError(f"{path}: not a frame pack (need {', '.join(SUFFIXES)}; the sparse files may stand in for the frames)")
        self.stem = stem
        self.meta = json.load(open(stem + ".meta.json"))
        self.frames = np.load(stem + ".frames.npy", mmap_mode="r") if os.path.exists(stem + ".frames.npy") else None
        self.labels = np.load(stem + ".labels.npy")
        self.flat_map = _read_map(stem + ".map.txt")
        self.sparse = None          # (bits [N, W], rank [N + 1], values): the sparse transport files, when present
        if all(os.path.exists(stem + s) for s in SPARSE_SUFFIXES):
            bits = np.load(stem + ".sparse.bits.npy", mmap_mode="r")
            rank = np.load(stem + ".sparse.rank.npy")
            values = np.load(stem + ".sparse.values.npy", mmap_mode="r")
            E = int(np.prod(self.meta["frame_dims"]))
            if (bits.ndim != 2 or bits.shape[0] != self.labels.shape[0] or bits.shape[1] * 32 < E or bits.shape[1] % 4 or len(rank) != bits.shape[0] + 1
                    or int(rank[-1]) > len(values) or bits.dtype != np.uint32 or values.dtype != np.float32):
                raise ValueError(f"{stem}: inconsistent sparse transport files")
            self.sparse = (bits, rank.astype(np.uint64), values)
        n_frames = self.frames.shape[0] if self.frames is not None else self.sparse[0].shape[0]
        if len(self.flat_map) != n_frames or self.labels.shape[0] != n_frames:
            raise ValueError(f"{stem}: inconsistent pack (map {len(self.flat_map)}, frames {n_frames})")
        self._index: Optional[Dict[Tuple[str, str, str], int]] = None
        self._cursor = 0

    def __len__(self):
        return self.labels.shape[0]

    @property
    def frame_dims(self):
        return tuple(self.meta["frame_dims"])

    def batch(self, lo: int, hi: int) -> Tuple[np.ndarray, np.ndarray]:
        """Rows [lo, hi) of the flat map: (frames view, labels as float like load_batch's y)."""
        if self.frames is None:
            return self.sparse_batch(lo, hi).dense(), self.labels[lo:hi].astype(float)
        return self.frames[lo:hi], self.labels[lo:hi].astype(float)

    def sparse_batch(self, lo: int, hi: int):
        """Rows [lo, hi) in the sparse transport form (engine.SparseFrames): views of the memory-mapped files, nothing is copied"""
        from . import engine
        bits, rank, values = self.sparse
        return engine.SparseFrames(bits[lo:hi], rank[lo:hi + 1], values[int(rank[lo]):int(rank[hi])], self.frame_dims)

    def contiguous_rows(self, data_point_batch) -> Optional[Tuple[int, int]]:
        """(lo, hi) when the batch is a contiguous slice of the map (what predict.py's loader asks for), else None"""
        n = len(data_point_batch)
        if not n:
            return None
        first = tuple(str(x) for x in data_point_batch[0][:3])
        for guess in (self._cursor, 0):
            if guess < len(self.flat_map) and tuple(self.flat_map[guess, :3]) == first:
                break
        else:
            if self._index is None:
                self._index = {(str(p), str(c), str(r)): i for i, (p, c, r, _l) in enumerate(self.flat_map)}
            guess = self._index.get(first)
            if guess is None:
                return None
        if guess + n > len(self.flat_map):
            return None
        rows = np.asarray(data_point_batch)
        if rows.ndim != 2 or not np.array_equal(self.flat_map[guess:guess + n, :3], rows[:, :3].astype(str)):
            return None
        return guess, guess + n

    def rows_of(self, data_point_batch) -> np.ndarray:
        if self._index is None:
            self._index = {(str(p), str(c), str(r)): i for i, (p, c, r, _l) in enumerate(self.flat_map)}
        return np.array([self._index[(str(p), str(c), str(r))] for p, c, r, *_ in data_point_batch], dtype=np.int64)

    def load_batch(self, data_point_batch) -> Tuple[np.ndarray, np.ndarray]:
        """Same contract as design_utils.utils.load_batch for an arbitrary list of map rows."""
        # the usual caller hands over a contiguous slice of the map itself: find it from its first row and confirm with
        # one vectorised comparison instead of a dictionary lookup per residue
        n = len(data_point_batch)
        if n and isinstance(data_point_batch, np.ndarray) and data_point_batch.ndim == 2 and data_point_batch.shape[1] >= 3:
            # a sequential reader continues where the previous batch ended: try that row first (no 100 k-entry
            # dictionary needed), then the start of the map, then the dictionary
            first = tuple(str(x) for x in data_point_batch[0, :3])
            i0 = None
            for guess in (self._cursor, 0):
                if guess < len(self.flat_map) and tuple(self.flat_map[guess, :3]) == first:
                    i0 = guess
                    break
            if i0 is None:
                if self._index is None:
                    self._index = {(str(p), str(c), str(r)): i for i, (p, c, r, _l) in enumerate(self.flat_map)}
                i0 = self._index.get(first)
            if i0 is not None and i0 + n <= len(self.flat_map) and np.array_equal(self.flat_map[i0:i0 + n, :3], data_point_batch[:, :3]):
                self._cursor = i0 + n
                return self.batch(i0, i0 + n)
        rows = self.rows_of(data_point_batch)
        if len(rows) and np.array_equal(rows, np.arange(rows[0], rows[0] + len(rows))):
            return self.batch(int(rows[0]), int(rows[0]) + len(rows))
        if self.frames is None:
            return np.concatenate([self.sparse_batch(int(r), int(r) + 1).dense() for r in rows]), self.labels[rows].astype(float)
        return np.asarray(self.frames[rows]), self.labels[rows].astype(float)


# ---- structures as datasets --------------------------------------------------------------------------------------
STRUCTURE_SUFFIXES = (".pdb", ".pdb.gz", ".pdb1", ".pdb1.gz", ".ent", ".ent.gz")


def is_structure(path) -> bool:
    p = os.fspath(path).lower()
    return p.endswith(STRUCTURE_SUFFIXES) and os.path.isfile(os.fspath(path))


class StructurePack:
    """A PDB file used directly as a dataset: voxelised once on the GPU (timed_hip.voxeliser — row f-4, parity unpinned
    against aposteriori) and kept in memory with the same ``flat_map`` / ``load_batch`` surface as a FramePack, so
    ``predict.py --path_to_dataset structure.pdb.gz`` needs neither aposteriori nor an HDF5 file."""

    def __init__(self, path, device: int = 0, gaussian: bool = True):
        import warnings
        from . import voxeliser
        warnings.warn(f"{os.fspath(path)}: frames are built by {voxeliser.PROVENANCE}; expect small differences from a "
                      "make-frame-dataset .hdf5 of the same structure", stacklevel=2)
        self.frames, labels, flat = voxeliser.voxelise_pdb(path, gaussian=gaussian, device=device)
        self.labels = labels
        self.flat_map = np.asarray(flat, dtype=str).reshape(-1, 4)
        self._index = {(p, c, r): i for i, (p, c, r, _l) in enumerate(self.flat_map)}

    def __len__(self):
        return self.frames.shape[0]

    def load_batch(self, data_point_batch):
        rows = np.array([self._index[(str(p), str(c), str(r))] for p, c, r, *_ in data_point_batch], dtype=np.int64)
        if len(rows) and np.array_equal(rows, np.arange(rows[0], rows[0] + len(rows))):
            return self.frames[rows[0]: rows[0] + len(rows)], self.labels[rows[0]: rows[0] + len(rows)].astype(float)
        return self.frames[rows], self.labels[rows].astype(float)
