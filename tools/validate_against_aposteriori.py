#!/usr/bin/env python
"""Pin the voxeliser to aposteriori itself: run aposteriori's `make_frame_dataset` on a structure with
exactly the arguments the reference passes (reference ui.py:73-86; README.md:83-97), read the frames it wrote, and compare them
with (a) oracle/voxel_oracle.py and (b) the HIP voxeliser when a GPU is present.

NOT runnable in the build image (aposteriori 2.4.0, ampal and h5py are absent — SURVEY.md §8c / Appendix D: the voxeliser is
"parity unpinned" for that reason).  Anyone who has the packages can close the gap:

    pip install aposteriori==2.4.0 h5py
    python tools/validate_against_aposteriori.py [--structure tests/golden/1ubq.pdb1.gz] [--emit-fixture]

`--emit-fixture` writes tests/golden/aposteriori_<code>.npz: the frames aposteriori produced (float32 [n, 21, 21, 21, C]), the
residue ids in dataset order, the one-hot labels, and every make_frame_dataset argument and file attribute.  tests/test_voxeliser.py
picks such a file up automatically and from then on the oracle and the HIP voxeliser are held to aposteriori's own output (no
kernel time goes into row f-4 before such a fixture exists: VERDICT r5 item 8).  `--dry-run` does what needs no aposteriori:
voxelises the structure with the oracle and prints the frame statistics the comparison would use.

Exit status 0 when max |frame_aposteriori - frame_ours| <= --tol on every frame and the residue order and labels agree."""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))


def run_aposteriori(structure: str, out_dir: str, gaussian: bool, encode_cb: bool):
    """the call of reference ui.py:73-86, argument for argument"""
    from pathlib import Path
    from aposteriori.data_prep.create_frame_data_set import Codec, make_frame_dataset
    p = Path(structure)
    code = p.name.split(".pdb")[0]
    kwargs = dict(structure_files=[p], output_folder=Path(out_dir), name=code, frame_edge_length=21.0, voxels_per_side=21,
                  codec=Codec.CNOCACB() if encode_cb else Codec.CNO(), processes=1, is_pdb_gzipped=p.suffix == ".gz",
                  require_confirmation=False, voxels_as_gaussian=gaussian, voxelise_all_states=False, verbosity=1)
    make_frame_dataset(**kwargs)
    return os.path.join(out_dir, code + ".hdf5"), {k: str(v) for k, v in kwargs.items()}


def read_dataset(path: str):
    """frames in flat-dataset-map order (reference design_utils/utils.py:362-393: pdb groups as stored, chains, residues by number)"""
    import h5py
    rows, frames, labels = [], [], []
    with h5py.File(path, "r") as f:
        attrs = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in f.attrs.items()}
        for pdb in f:
            for chain in f[pdb]:
                for res in sorted(f[pdb][chain], key=lambda s: int(s)):
                    d = f[pdb][chain][res]
                    frames.append(np.asarray(d[()]))
                    labels.append(np.asarray(d.attrs["encoded_residue"]))
                    rows.append((pdb, chain, res, str(d.attrs["label"])))
    return np.stack(frames).astype(np.float32), np.stack(labels).astype(np.uint8), rows, attrs


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--structure", default=os.path.join(ROOT, "tests", "golden", "1ubq.pdb1.gz"))
    ap.add_argument("--boolean", action="store_true", help="voxels_as_gaussian=False (one-hot voxels) instead of Gaussian frames")
    ap.add_argument("--no-cb", action="store_true", help="Codec.CNO() instead of CNOCACB")
    ap.add_argument("--emit-fixture", action="store_true")
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--tol", type=float, default=1e-4)
    args = ap.parse_args()
    from oracle import voxel_oracle
    from timed_hip import pdbio, voxeliser
    code = os.path.basename(args.structure).split(".pdb")[0]
    encoder = voxeliser.DEFAULT_ENCODER if not args.no_cb else tuple(a for a in voxeliser.DEFAULT_ENCODER if a != "CB")
    model = pdbio.read_pdb(args.structure)[0]
    xyz, ch, sg, frt, rows = voxeliser.prepare_structure(model, encode_cb=not args.no_cb, atom_encoder=encoder)
    ours = voxel_oracle.voxelise(xyz, ch, sg, frt, 21, 21.0, len(encoder), not args.boolean)
    print(f"{code}: {len(rows)} residues; oracle frames {ours.shape}, non-zero fraction {np.count_nonzero(ours) / ours.size:.4f}, max {ours.max():.4f}")
    if args.dry_run:
        return 0
    try:
        import aposteriori  # noqa: F401
        import h5py  # noqa: F401
    except ImportError as e:
        print(f"needs aposteriori 2.4.0 and h5py ({e}); nothing compared", file=sys.stderr)
        return 2
    with tempfile.TemporaryDirectory() as td:
        path, call = run_aposteriori(args.structure, td, not args.boolean, not args.no_cb)
        theirs, labels, drows, attrs = read_dataset(path)
    ok = True
    if [(c, n) for _p, c, n, _l in drows] != [(c, n) for c, n, _l in rows]:
        print("residue order differs:", drows[:3], rows[:3]); ok = False
    if theirs.shape != ours.shape:
        print("frame shapes differ:", theirs.shape, ours.shape); ok = False
    else:
        d = np.abs(theirs.astype(np.float64) - ours.astype(np.float64)).reshape(len(theirs), -1).max(1)
        print(f"oracle vs aposteriori: max |d| {d.max():.3e} (frame {int(d.argmax())}), frames above {args.tol:g}: {int((d > args.tol).sum())} of {len(d)}")
        ok &= bool(d.max() <= args.tol)
        try:
            gpu, _l, _f = voxeliser.voxelise_pdb(args.structure, gaussian=not args.boolean, encode_cb=not args.no_cb, atom_encoder=encoder)
            dg = np.abs(theirs.astype(np.float64) - gpu.astype(np.float64)).max()
            print(f"HIP voxeliser vs aposteriori: max |d| {dg:.3e}")
            ok &= bool(dg <= args.tol)
        except Exception as e:
            print(f"(HIP voxeliser not run: {e})")
    if args.emit_fixture:
        out = os.path.join(ROOT, "tests", "golden", f"aposteriori_{code}.npz")
        np.savez_compressed(out, frames=theirs, labels=labels, rows=np.asarray(drows, dtype=str), call=np.asarray(sorted(call.items()), dtype=str),
                            attrs=np.asarray(sorted((k, str(v)) for k, v in attrs.items()), dtype=str),
                            structure=os.path.basename(args.structure), gaussian=not args.boolean, encode_cb=not args.no_cb)
        print("wrote", out, "- tests/test_voxeliser.py picks it up")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
