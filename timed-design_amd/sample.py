"""sample.py — the reference's sampling command line, output files and ``main_sample(args) -> output_paths``
(reference sample.py:19-93, consumed by ui.py:320-321), with the temperature normaliser and every categorical draw
on the GPU: all PDB keys of the prediction matrix are loaded once into a resident sampler and drawn in one launch
sequence (design_utils.sampling_utils.sample_with_multiprocessing -> th_sampler_load / th_sampler_draw).

    python sample.py --path_to_pred_matrix TIMED.csv --path_to_datasetmap TIMED.txt --sample_n 200 --temperature 0.1

Difference from the reference, on purpose: ``--seed`` WORKS.  The reference builds
``np.random.default_rng(seed)`` and drops it (sample.py:21) while drawing from the never-seeded global legacy
generator, in forked workers that share one state; here ``np.random.seed(seed)`` seeds that same legacy stream and
the PDB keys are sampled in key order from it, so a run is reproducible and equals the reference's single-process
draw order.
"""
import argparse
from pathlib import Path

import numpy as np

from design_utils import sampling_utils as su
from design_utils import utils as du
from design_utils.amino_acids import standard_amino_acids


def _one_letter_rotamer_categories():
    """the 338 rotamer classes as the one-letter code of their residue ("ARG_1123" -> "R")"""
    three_to_one = {three: one for one, three in standard_amino_acids.items()}
    _, names = du.get_rotamer_codec()
    return [three_to_one[name.split("_")[0]] for name in names]


def _read_matrix(path) -> np.ndarray:
    """The prediction matrix as float64 — what ``np.genfromtxt(path, delimiter=",", dtype=np.float64)`` returns
    (reference sample.py:31-33) — through NumPy's C parser (both convert every field with a correctly rounded
    strtod); anything that parser refuses (missing fields, ragged rows) goes to genfromtxt and its rules."""
    try:
        return np.loadtxt(path, delimiter=",", dtype=np.float64, ndmin=2)
    except ValueError:
        return np.atleast_2d(np.genfromtxt(path, delimiter=",", dtype=np.float64))


def main_sample(args):
    np.random.seed(args.seed)
    su.reset_device_rng()                        # --seed restarts the device generators' streams too (they run on across library calls)
    matrix_path, map_path = Path(args.path_to_pred_matrix), Path(args.path_to_datasetmap)
    for what, path in (("prediction matrix", matrix_path), ("dataset map", map_path)):
        assert path.exists(), f"No {what} at {path}"
    probabilities = _read_matrix(matrix_path)
    if args.temperature != 1:           # at T == 1 the rows are used exactly as stored (not even renormalised)
        probabilities = su.apply_temp_to_probs(probabilities, t=args.temperature)
    categories = _one_letter_rotamer_categories() if args.predict_rotamers else None
    dataset_map = du.load_datasetmap(map_path, is_old=args.support_old_datasetmap)
    _seq, pdb_to_probability, _real, _c, _cp = du.extract_sequence_from_pred_matrix(
        dataset_map, probabilities, rotamers_categories=categories, old_datasetmap=args.support_old_datasetmap)
    keys = list(pdb_to_probability)
    print(f"Drawing {args.sample_n} sequence(s) for each of {len(keys)} structure(s) in {matrix_path.name}")
    sampled = su.sample_with_multiprocessing(args.workers, keys, args.sample_n, pdb_to_probability, categories,
                                             rng=getattr(args, "rng", "numpy"), seed=args.seed)
    stem = f"{matrix_path.stem}_temp_{args.temperature}_n_{args.sample_n}_{keys[0]}"
    return su.save_as(sampled, filename=stem, mode=args.save_as)


# (flag, argparse keywords): names, types and defaults are the reference's (sample.py:96-147)
CLI_FLAGS = (
    ("--path_to_pred_matrix", dict(type=str, help="probability matrix written by predict.py (.csv)")),
    ("--path_to_datasetmap", dict(type=str, default="datasetmap.txt", help="dataset map written by predict.py (.txt)")),
    ("--predict_rotamers", dict(action="store_true", default=False, help="the matrix has 338 rotamer columns instead of 20 residues")),
    ("--sample_n", dict(type=int, default=100, help="sequences to draw per structure")),
    ("--save_as", dict(type=str, default="all", const="all", nargs="?", choices=["fasta", "json", "all"],
                       help="sequence file formats to write (the metrics csv is always written)")),
    ("--workers", dict(type=int, default=8, help="accepted for compatibility; the draws run on the GPU")),
    ("--temperature", dict(type=float, default=1, help="softmax temperature applied to the probabilities (1 = unchanged)")),
    ("--support_old_datasetmap", dict(action="store_true", default=False, help="the dataset map is the old 4-column csv")),
    ("--seed", dict(type=int, default=42, help="seed of NumPy's legacy generator (and of the device generators of --rng)")),
    # not in the reference: where the uniforms come from.  numpy = np.random.rand replayed bit for bit (the reference's stream
    # after np.random.seed(seed)); philox = rocRAND Philox4x32-10 on the GPU; mt19937 = MT19937(seed) generated on the GPU
    ("--rng", dict(type=str, default="numpy", choices=list(su.RNG_CHOICES), help="source of the uniforms (default: numpy, the reference's stream)")),
)


def build_parser():
    parser = argparse.ArgumentParser(description="Monte-Carlo sequences from predicted residue probabilities (MI355X)")
    for flag, keywords in CLI_FLAGS:
        parser.add_argument(flag, **keywords)
    return parser


if __name__ == "__main__":
    main_sample(build_parser().parse_args())
