#!/usr/bin/env python
"""Pin the four sequence metrics to ampal itself: evaluate ampal's sequence_charge,
sequence_isoelectric_point, sequence_molecular_weight and sequence_molar_extinction_280 — the four calls of reference
design_utils/analyse_utils.py:351-371 — on a fixed set of sequences and compare with oracle/seqmetrics_oracle.py, the product's
host code (design_utils/analyse_utils.py here) and, with a GPU, the metrics kernel.

NOT runnable in the build image (ampal 1.5.1 is absent: "parity unpinned").  Where it is installed:

    pip install ampal==1.5.1
    python tools/validate_against_ampal.py [--emit-fixture]

`--emit-fixture` writes tests/golden/ampal_seqmetrics.npz (sequences + ampal's four numbers each, ampal's version);
tests/test_oracle_seqmetrics.py picks it up and holds the oracle and the product to it.  `--dry-run`: the oracle's numbers only."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))

FIXED = ["ACDEFGHIKLMNPQRSTVWY", "KKKKDE", "WYCG", "G", "W" * 50, "DDDDEEEE", "RRRRKKKKHH", "CCCC",
         "MQIFVKTLTGKTITLEVEPSDTIENVKAKIQDKEGIPPDQQRLIFAGKQLEDGRTLSDYNIQKESTLHLVLRLRGG"]


def sequences():
    rng = np.random.default_rng(20240930)
    letters = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    return FIXED + ["".join(letters[rng.integers(0, 20, size=int(rng.integers(1, 400)))]) for _ in range(300)]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--emit-fixture", action="store_true")
    ap.add_argument("--dry-run", action="store_true")
    args = ap.parse_args()
    from oracle import seqmetrics_oracle as so
    seqs = sequences()
    ours = np.array([so.seq_metrics(s) for s in seqs], dtype=np.float64)
    print(f"{len(seqs)} sequences; oracle on ubiquitin: {ours[len(FIXED) - 1]}")
    if args.dry_run:
        return 0
    try:
        import ampal
        from ampal.analyse_protein import (sequence_charge, sequence_isoelectric_point, sequence_molar_extinction_280,
                                           sequence_molecular_weight)
    except ImportError as e:
        print(f"needs ampal 1.5.1 ({e}); nothing compared", file=sys.stderr)
        return 2
    theirs = np.array([(sequence_charge(s), sequence_isoelectric_point(s), sequence_molecular_weight(s), sequence_molar_extinction_280(s))
                       for s in seqs], dtype=np.float64)
    d = np.abs(theirs - ours)
    rel = d / np.maximum(1.0, np.abs(theirs))
    names = ("charge", "isoelectric point", "molecular weight", "extinction 280")
    for k, nm in enumerate(names):
        print(f"{nm:18s} max |d| {d[:, k].max():.3e}  (relative {rel[:, k].max():.3e}, sequence #{int(d[:, k].argmax())})")
    ok = bool(rel[:, 0].max() <= 1e-9 and d[:, 1].max() <= 1e-9 and rel[:, 2].max() <= 1e-9 and d[:, 3].max() == 0)
    if args.emit_fixture:
        out = os.path.join(ROOT, "tests", "golden", "ampal_seqmetrics.npz")
        np.savez_compressed(out, sequences=np.asarray(seqs, dtype=str), metrics=theirs, ampal_version=str(getattr(ampal, "__version__", "?")))
        print("wrote", out, "- tests/test_oracle_seqmetrics.py picks it up")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
