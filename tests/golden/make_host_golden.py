#!/usr/bin/env python
"""Generate tests/golden/host_golden.npz — byte-exact outputs of the reference's host functions
either side of the kernels, captured by IMPORTING
/root/reference/design_utils in the build container (third-party imports stubbed, see
make_sampler_golden.py).  Only inputs and produced bytes/values are stored.

Usage:  python tests/golden/make_host_golden.py
"""
import io
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_sampler_golden import stub_modules  # noqa: E402


def main():
    stub_modules()
    sys.path.insert(0, "/root/reference")
    from design_utils import sampling_utils as ref_su
    from design_utils import utils as ref_utils

    rng = np.random.default_rng(2024)
    out = {}
    three = ["ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU", "MET", "ASN", "PRO", "GLN", "ARG",
             "SER", "THR", "VAL", "TRP", "TYR"]
    # ---- a flat dataset map (old 4-column form, utils.py:393) with two pdbs / three chains --------
    flat = []
    for pdb, chain, n in (("1ubq", "A", 9), ("2abc", "A", 5), ("2abc", "B", 4)):
        for r in range(1, n + 1):
            flat.append((pdb, chain, str(r + 3), three[int(rng.integers(0, 20))]))
    flat = np.array(flat)
    N = len(flat)
    probs = rng.dirichlet(np.full(20, 0.4), size=N).astype(np.float32)
    y_true = np.eye(20)[rng.integers(0, 20, N)]
    out["flat_map"] = flat
    out["probs32"] = probs
    out["y_true"] = y_true

    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        # save_outputs_to_file, two batches appended (utils.py:726-771)
        for lo, hi in ((0, 10), (10, N)):
            ref_utils.save_outputs_to_file(list(y_true[lo:hi]), {0: list(probs[lo:hi])}, flat, 0, "TIMED", td)
        for fn in ("encoded_labels.csv", "datasetmap.txt", "TIMED.csv"):
            out["file_" + fn] = np.array((td / fn).read_text())
        ref_utils.convert_dataset_map_for_srb(flat, "TIMED", td)
        out["file_TIMED.txt"] = np.array((td / "TIMED.txt").read_text())
        pm16 = np.genfromtxt(td / "TIMED.csv", delimiter=",", dtype=np.float16)   # predict.py:163
        out["pred_matrix_f16_reread"] = pm16.astype(np.float32)
        res = ref_utils.extract_sequence_from_pred_matrix(flat, pm16, rotamers_categories=None, old_datasetmap=True)
        out["extract_old_json"] = np.array(json.dumps(
            dict(seq=res[0], prob={k: np.asarray(v, dtype=np.float64).tolist() for k, v in res[1].items()}, real=res[2])))
        ref_utils.save_dict_to_fasta(res[0], "TIMED", td)
        ref_utils.save_dict_to_fasta(res[2], "dataset", td)
        out["file_TIMED.fasta"] = np.array((td / "TIMED.fasta").read_text())
        out["file_dataset.fasta"] = np.array((td / "dataset.fasta").read_text())
        # the sample.py side: float64 re-read + PDBench map (sample.py:32-38, utils.py:190-227)
        pm64 = np.genfromtxt(td / "TIMED.csv", delimiter=",", dtype=np.float64)
        dmap = ref_utils.load_datasetmap(td / "TIMED.txt")
        out["load_datasetmap"] = np.asarray(dmap)
        res2 = ref_utils.extract_sequence_from_pred_matrix(dmap, pm64, rotamers_categories=None, old_datasetmap=False)
        out["extract_new_json"] = np.array(json.dumps(dict(seq=res2[0], prob=res2[1], real=res2[2])))
        # consensus (NMR) path, utils.py:694-715: keys "<pdb>_<state>..." averaged pairwise
        flat_nmr = np.array([(f"1xyz_{s}", "A", str(r), three[(r * 3 + s) % 20]) for s in range(3) for r in range(1, 5)])
        p_nmr = rng.dirichlet(np.full(20, 0.4), size=len(flat_nmr)).astype(np.float16)
        rc = ref_utils.extract_sequence_from_pred_matrix(flat_nmr, p_nmr, rotamers_categories=None, is_consensus=True)
        out["nmr_flat_map"] = flat_nmr
        out["nmr_probs16"] = p_nmr.astype(np.float32)
        out["nmr_consensus_json"] = np.array(json.dumps(
            dict(seq=rc[0], consensus=rc[3], consensus_prob={k: np.asarray(v, dtype=np.float64).tolist() for k, v in rc[4].items()})))
        # rotamer mode: 338-wide matrix -> one-letter categories (sample.py:43-52)
        _, cats = ref_utils.get_rotamer_codec()
        aa = sys.modules["ampal.amino_acids"].standard_amino_acids
        res_to_r = dict(zip(aa.values(), aa.keys()))
        cats1 = [res_to_r[c.split("_")[0]] for c in cats]
        p338 = rng.dirichlet(np.full(338, 0.05), size=N)
        r3 = ref_utils.extract_sequence_from_pred_matrix(flat, p338, rotamers_categories=cats, old_datasetmap=True)
        r4 = ref_utils.extract_sequence_from_pred_matrix(flat, p338, rotamers_categories=cats1, old_datasetmap=True)
        out["probs338"] = p338
        out["extract_rot_seq_json"] = np.array(json.dumps(dict(full=r3[0], one=r4[0])))
        # predict.py:147-148: rotamer argmax -> 20-way one-hot through the codec
        codec, _ = ref_utils.get_rotamer_codec()
        out["rot_onehot20"] = np.array([codec[c] for c in np.argmax(p338, axis=1)])
        # save_as (sampling_utils.py:12-50)
        sampled = {"1ubqA": [("ACDEFGHIK", 0.5, 6.1, 1000.25, 120), ("KIHGFEDCA", -1.0, 5.0, 999.0, 0)],
                   "2abcA": [("MKV", 1.0, 9.7, 376.5, 0)]}
        paths = ref_su.save_as(sampled, str(td / "TIMED_temp_0.5_n_2_1ubqA"), "all")
        out["save_as_paths"] = np.array([os.path.basename(p) for p in paths])
        for p in paths:
            out["file_" + os.path.basename(p)] = np.array(Path(p).read_text())
        out["save_as_input_json"] = np.array(json.dumps(sampled))
    # blacklist key parsing (utils.py:284-315) is trivial text; skipped.
    path = os.path.join(ROOT, "tests", "golden", "host_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "entries")


if __name__ == "__main__":
    main()
