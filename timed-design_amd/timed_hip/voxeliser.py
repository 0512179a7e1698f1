"""GPU voxeliser: PDB structure -> per-residue 21x21x21xC frames.

Replaces the producer of the reference's input: ``aposteriori.make_frame_dataset`` as the reference invokes it
(ui.py:73-86: frame_edge_length=21.0, voxels_per_side=21, codec=Codec.CNOCACB(), voxels_as_gaussian=True,
voxelise_all_states=False; README.md:83-97: ``make-frame-dataset ... -cb True -ae CNOCBCA -g True``), and writes what
predict.py consumes (a frame pack, timed_hip/framepack.py, or device-resident frames for th_predict_device) — so a
structure goes PDB -> frames -> probabilities without the HDF5 round trip.

    *** PARITY UNPINNED ***  aposteriori==2.4.0 is an external package whose source is neither in /root/reference nor
    in this image, and the reference has no test at this boundary.  What IS fixed by the reference tree: the dataset
    layout and attributes (design_utils/utils.py:238-251), the frame geometry the UI asks for (ui.py:73-86), the atom
    encoder names, and the idealised C-beta position (-0.741287356, -0.53937931, -1.224287356) quoted at utils.py:247.
    Everything else below is this module's own specification, written from aposteriori's documented behaviour; a model
    trained on aposteriori's frames must be validated on frames from here before the two are mixed.  The oracle
    (oracle/voxel_oracle.py) restates exactly this specification and the GPU kernel is tested against it.

Specification
  1. Atoms: ATOM records (HETATM only for residues of UNCOMMON_RESIDUE_DICT, mapped to their parent amino acid); first
     alternate location; model 1 unless ``all_states``.
  2. Encoded atoms ("keep backbone, add C-beta"): N, CA, C, O of every residue of every chain, plus — with
     ``encode_cb`` — ONE idealised C-beta per residue that has N, CA and C, at CB_LOCAL in that residue's own frame
     (real C-beta and all side-chain atoms are dropped).  Channels follow ``atom_encoder`` (default C, N, O, CA, CB).
  3. Residue frame (aposteriori's align_to_residue_plane): origin at CA; +y along CA->N; C in the xy half-plane x > 0;
     z = x cross y.  local = R (p - CA) with R = rows (e_x, e_y, e_z), evaluated in float32 as
     (R_i0*dx + R_i1*dy) + R_i2*dz with separate multiplies and adds.
  4. Grid: voxel edge a = frame_edge_length / voxels_per_side; index_k = floor(local_k / a + 0.5) + voxels_per_side//2;
     an atom is encoded iff all three indices are inside [0, voxels_per_side).
  5. Boolean frames (voxels_as_gaussian=False): frame[index][channel] = 1 (uint8).
  6. Gaussian frames: the atom is spread over the 3x3x3 block of voxels around its own voxel with weights
     w = exp(-r^2 / (2 sigma^2)), r = distance from the voxel centre to the atom, sigma = sigma_scale * vdW radius
     (C 1.70, N 1.55, O 1.52 Angstrom; sigma_scale 0.5), normalised so that the 27 weights sum to 1 (weight falling
     outside the frame is lost); contributions are added in atom order, float32, no clipping.
  7. One frame per residue that has N, CA, C and a standard (or mapped) name; label = one-hot over
     ``standard_amino_acids`` order; dataset-map rows (pdb_code, chain, residue number, label) in file order with
     residues sorted numerically per chain, as create_flat_dataset_map would list them.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, pdbio

CB_LOCAL = np.array([-0.741287356, -0.53937931, -1.224287356])     # reference design_utils/utils.py:247
DEFAULT_ENCODER = ("C", "N", "O", "CA", "CB")
VDW_RADIUS = {"C": 1.70, "N": 1.55, "O": 1.52}
SIGMA_SCALE = 0.5
BACKBONE = ("N", "CA", "C", "O")


def residue_frame(n: np.ndarray, ca: np.ndarray, c: np.ndarray) -> Optional[np.ndarray]:
    """3x3 rotation (rows e_x, e_y, e_z) of spec item 3, float64; None for degenerate geometry."""
    ey = n - ca
    ny = np.linalg.norm(ey)
    if ny < 1e-6:
        return None
    ey = ey / ny
    cp = c - ca
    cp = cp - np.dot(cp, ey) * ey
    nx = np.linalg.norm(cp)
    if nx < 1e-6:
        return None
    ex = cp / nx
    ez = np.cross(ex, ey)
    return np.stack([ex, ey, ez])


def prepare_structure(model: pdbio.Model, encode_cb: bool = True, atom_encoder: Sequence[str] = DEFAULT_ENCODER,
                      uncommon: Optional[dict] = None):
    """Host-side preparation of one model: (atoms_xyz f32 [n,3], atom_channel i32 [n], atom_sigma f32 [n],
    frames_rt f32 [n_res,12], rows [(chain, number, three-letter label)])."""
    from design_utils.amino_acids import UNCOMMON_RESIDUE_DICT, standard_amino_acids
    uncommon = UNCOMMON_RESIDUE_DICT if uncommon is None else uncommon
    standard = set(standard_amino_acids.values())
    chan = {name: i for i, name in enumerate(atom_encoder)}
    xyz: List[np.ndarray] = []
    ch: List[int] = []
    sg: List[float] = []
    frames: List[np.ndarray] = []
    rows: List[Tuple[str, str, str]] = []
    per_chain: dict = {}
    for res in model.residues:
        label = res.name if res.name in standard else uncommon.get(res.name)
        if label is None or (res.hetero and res.name not in uncommon):
            continue
        R = None
        if all(a in res.atoms for a in ("N", "CA", "C")):
            R = residue_frame(res.atoms["N"], res.atoms["CA"], res.atoms["C"])
        for a in BACKBONE:
            if a in res.atoms and a in chan:
                xyz.append(res.atoms[a]); ch.append(chan[a]); sg.append(SIGMA_SCALE * VDW_RADIUS[a[0]])
        if R is not None:
            if encode_cb and "CB" in chan:
                xyz.append(res.atoms["CA"] + R.T @ CB_LOCAL); ch.append(chan["CB"]); sg.append(SIGMA_SCALE * VDW_RADIUS["C"])
            per_chain.setdefault(res.chain, []).append((res, R, label))
    for chain, items in per_chain.items():
        def key(it):
            digits = "".join(c for c in it[0].number if c.isdigit() or c == "-")
            return (int(digits) if digits not in ("", "-") else 0, it[0].number)
        for res, R, label in sorted(items, key=key):
            frames.append(np.concatenate([R.reshape(9), res.atoms["CA"]]))
            rows.append((chain, res.number, label))
    return (np.asarray(xyz, dtype=np.float32).reshape(-1, 3), np.asarray(ch, dtype=np.int32), np.asarray(sg, dtype=np.float32),
            np.asarray(frames, dtype=np.float32).reshape(-1, 12), rows)


def voxelise(atoms_xyz, atom_channel, atom_sigma, frames_rt, voxels_per_side: int = 21, frame_edge_length: float = 21.0,
             n_channels: int = 5, gaussian: bool = True, device: int = 0, d_out: Optional[int] = None) -> Optional[np.ndarray]:
    """Run the kernel (th_voxelise).  Returns frames [n_res, V, V, V, C] (float32 for Gaussian, uint8 for boolean frames),
    or None when ``d_out`` (a device address with room for them) is given: the frames then stay in HBM for
    th_predict_device."""
    lib = _lib.load()
    xyz = np.ascontiguousarray(atoms_xyz, dtype=np.float32)
    chn = np.ascontiguousarray(atom_channel, dtype=np.int32)
    sig = np.ascontiguousarray(atom_sigma, dtype=np.float32)
    frt = np.ascontiguousarray(frames_rt, dtype=np.float32)
    n_res, V = frt.shape[0], int(voxels_per_side)
    out = None
    if d_out is None:
        out = np.empty((n_res, V, V, V, n_channels), dtype=np.float32 if gaussian else np.uint8)
    if n_res:
        _lib.check(lib.th_voxelise(device, xyz.ctypes.data, chn.ctypes.data, sig.ctypes.data, xyz.shape[0], frt.ctypes.data, n_res,
                                   V, float(frame_edge_length), n_channels, 1 if gaussian else 0,
                                   C.c_void_p(d_out) if d_out is not None else out.ctypes.data, 1 if d_out is not None else 0))
    return out


def voxelise_pdb(path, voxels_per_side: int = 21, frame_edge_length: float = 21.0, encode_cb: bool = True,
                 atom_encoder: Sequence[str] = DEFAULT_ENCODER, gaussian: bool = True, all_states: bool = False, device: int = 0):
    """PDB file -> (frames, labels [n,20] uint8, flat dataset map rows (pdb_code, chain, residue number, label)).
    The pdb code is the file name up to ".pdb"; with ``all_states`` every model is voxelised as "<code>_<k>"
    (aposteriori's voxelise_all_states), otherwise model 1 only."""
    from design_utils.amino_acids import standard_amino_acids
    three = list(standard_amino_acids.values())
    code = os.path.basename(str(path)).split(".pdb")[0]
    models = pdbio.read_pdb(path)
    if not models:
        raise ValueError(f"{path}: no ATOM records")
    if not all_states:
        models = models[:1]
    frames, labels, flat = [], [], []
    for k, model in enumerate(models):
        xyz, ch, sg, frt, rows = prepare_structure(model, encode_cb=encode_cb, atom_encoder=atom_encoder)
        name = f"{code}_{k}" if all_states else code
        frames.append(voxelise(xyz, ch, sg, frt, voxels_per_side, frame_edge_length, len(atom_encoder), gaussian, device))
        for chain, number, label in rows:
            flat.append((name, chain, number, label))
            onehot = np.zeros(20, np.uint8)
            onehot[three.index(label)] = 1
            labels.append(onehot)
    X = np.concatenate(frames) if frames else np.empty((0,) + (voxels_per_side,) * 3 + (len(atom_encoder),), np.float32)
    return X, np.asarray(labels, dtype=np.uint8).reshape(-1, 20), flat


PROVENANCE = "timed_hip.voxeliser (in-house GPU voxeliser; parity with aposteriori 2.4.0 NOT pinned by a golden file)"


def write_frame_pack(stem, frames: np.ndarray, labels: np.ndarray, flat_map, gaussian: bool, source: str = ""):
    """Store voxelised frames as a frame pack that predict.py accepts wherever it accepts an .hdf5 path."""
    import json
    stem = os.fspath(stem)
    np.save(stem + ".frames.npy", frames)
    np.save(stem + ".labels.npy", labels)
    np.savetxt(stem + ".map.txt", np.asarray(flat_map), delimiter=",", fmt="%s")
    with open(stem + ".meta.json", "w") as f:
        json.dump(dict(frame_dims=list(frames.shape[1:]), voxels_as_gaussian=bool(gaussian), n_frames=int(frames.shape[0]),
                       source=source, make_frame_dataset_ver="timed_hip.voxeliser (unpinned vs aposteriori 2.4.0)",
                       frame_provenance=PROVENANCE), f)


def write_hdf5(path, frames: np.ndarray, labels: np.ndarray, flat_map, gaussian: bool, atom_encoder: Sequence[str] = DEFAULT_ENCODER,
               frame_edge_length: float = 21.0, encode_cb: bool = True, compression: Optional[str] = "gzip"):
    """Store voxelised frames in aposteriori's own layout (reference design_utils/utils.py:238-251) — pdb_code / chain /
    residue number datasets with `label` and `encoded_residue` attributes, file attributes as make-frame-dataset writes
    them — with timed_hip.h5write, so that h5py, the reference's predict.py and this repo's readers all open it.  Frames
    are stored as float64 (Gaussian) or bool like the reference's loader expects (utils.py:518-521)."""
    from design_utils.amino_acids import standard_amino_acids
    from . import h5write
    with h5write.File(path) as f:
        f.attrs["make_frame_dataset_ver"] = "2.4.0"        # the layout version the reference's checks accept (utils.py:271-280)
        # ... but these frames did NOT come from aposteriori: a separate attribute lets any downstream tool tell them apart
        f.attrs["frame_provenance"] = PROVENANCE
        f.attrs["frame_dims"] = np.asarray(frames.shape[1:], dtype=np.int64)
        f.attrs["atom_encoder"] = list(atom_encoder)
        f.attrs["encode_cb"] = bool(encode_cb)
        f.attrs["atom_filter_fn"] = "timed_hip.voxeliser: backbone + idealised C-beta (unpinned vs aposteriori)"
        f.attrs["residue_encoder"] = list(standard_amino_acids.keys())
        f.attrs["frame_edge_length"] = float(frame_edge_length)
        f.attrs["voxels_as_gaussian"] = bool(gaussian)
        groups: dict = {}
        for i, (pdb, chain, number, label) in enumerate(flat_map):
            g = groups.get((pdb,))
            if g is None:
                g = groups[(pdb,)] = f.create_group(str(pdb))
            c = groups.get((pdb, chain))
            if c is None:
                c = groups[(pdb, chain)] = g.create_group(str(chain))
            data = frames[i].astype(np.float64) if gaussian else frames[i].astype(bool)
            c.create_dataset(str(number), data, compression=compression,
                             attrs={"label": str(label), "encoded_residue": labels[i].astype(np.float64)})
