"""Packed frame datasets — the HDF5-free ingest path.

The reference reads one small HDF5 dataset per residue, re-opening the file for every batch
(reference design_utils/utils.py:514-529): a few hundred frames/s, three orders of magnitude below
what the GPU consumes.  A *frame pack* is the same information laid out for streaming:

    <stem>.frames.npy   [N, D, H, W, C]  uint8 (boolean datasets) or float32 (Gaussian datasets),
                        rows in flat-dataset-map order (create_flat_dataset_map, reference utils.py:362-393)
    <stem>.labels.npy   [N, 20] uint8 one-hot ``encoded_residue`` rows
    <stem>.map.txt      the flat dataset map as the reference writes it (csv: pdb,chain,residue,label)
    <stem>.meta.json    frame_dims, voxels_as_gaussian, source file, aposteriori version string

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


def is_pack(path) -> bool:
    p = os.fspath(path)
    stem = pack_stem(p)
    return stem is not None and all(os.path.exists(stem + s) for s in SUFFIXES)


def pack_stem(path) -> Optional[str]:
    p = os.fspath(path)
    for s in SUFFIXES:
        if p.endswith(s):
            return p[: -len(s)]
    if p.endswith(".framepack"):
        return p[: -len(".framepack")]
    if all(os.path.exists(p + s) for s in SUFFIXES):
        return p
    return None


def pack_dataset(hdf5_path, out_stem, filter_list: Sequence[str] = (), remove_blacklist_silently: bool = False,
                 progress_every: int = 0) -> "FramePack":
    """Convert an aposteriori frame dataset (HDF5) into a frame pack.  One sequential pass."""
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
        if stem is None or not all(os.path.exists(stem + s) for s in SUFFIXES):
            raise FileNotFoundError(f"{path}: not a frame pack (need {', '.join(SUFFIXES)})")
        self.stem = stem
        self.meta = json.load(open(stem + ".meta.json"))
        self.frames = np.load(stem + ".frames.npy", mmap_mode="r")
        self.labels = np.load(stem + ".labels.npy")
        self.flat_map = _read_map(stem + ".map.txt")
        if len(self.flat_map) != self.frames.shape[0] or self.labels.shape[0] != self.frames.shape[0]:
            raise ValueError(f"{stem}: inconsistent pack (map {len(self.flat_map)}, frames {self.frames.shape[0]})")
        self._index: Optional[Dict[Tuple[str, str, str], int]] = None
        self._cursor = 0

    def __len__(self):
        return self.frames.shape[0]

    @property
    def frame_dims(self):
        return tuple(self.meta["frame_dims"])

    def batch(self, lo: int, hi: int) -> Tuple[np.ndarray, np.ndarray]:
        """Rows [lo, hi) of the flat map: (frames view, labels as float like load_batch's y)."""
        return self.frames[lo:hi], self.labels[lo:hi].astype(float)

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
