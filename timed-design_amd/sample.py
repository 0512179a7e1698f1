"""sample.py — same command line, outputs and ``main_sample(args) -> output_paths`` as the reference's
sample.py (reference sample.py:19-93), with temperature and the categorical draw on the GPU.

    python sample.py --path_to_pred_matrix TIMED.csv --path_to_datasetmap TIMED.txt --sample_n 200 --temperature 0.1

Difference from the reference, on purpose: ``--seed`` WORKS.  The
reference builds ``np.random.default_rng(seed)`` and drops it (sample.py:21) while drawing from the
never-seeded global legacy generator, in forked workers that share one state; here
``np.random.seed(seed)`` seeds that same legacy stream and the PDB keys are sampled in order in one
process, so a run is reproducible and equals the reference's single-process draw order.
"""
import argparse
from pathlib import Path

import numpy as np

from design_utils.amino_acids import standard_amino_acids
from design_utils.sampling_utils import apply_temp_to_probs, sample_with_multiprocessing, save_as
from design_utils.utils import extract_sequence_from_pred_matrix, get_rotamer_codec, load_datasetmap


def main_sample(args):
    np.random.seed(args.seed)
    args.path_to_pred_matrix = Path(args.path_to_pred_matrix)
    args.path_to_datasetmap = Path(args.path_to_datasetmap)
    assert args.path_to_pred_matrix.exists(), f"Prediction Matrix file {args.path_to_pred_matrix} does not exist"
    assert args.path_to_datasetmap.exists(), f"Dataset Map file {args.path_to_datasetmap} does not exist"
    prediction_matrix = np.atleast_2d(np.genfromtxt(args.path_to_pred_matrix, delimiter=",", dtype=np.float64))
    datasetmap = load_datasetmap(args.path_to_datasetmap, is_old=args.support_old_datasetmap)
    if args.temperature != 1:
        prediction_matrix = apply_temp_to_probs(prediction_matrix, t=args.temperature)
    if args.predict_rotamers:
        _, flat_categories = get_rotamer_codec()
        res_to_r = dict(zip(standard_amino_acids.values(), standard_amino_acids.keys()))
        flat_categories = [res_to_r[res.split("_")[0]] for res in flat_categories]
    else:
        flat_categories = None
    (pdb_to_sequence, pdb_to_probability, pdb_to_real_sequence, _, _) = extract_sequence_from_pred_matrix(
        datasetmap, prediction_matrix, rotamers_categories=flat_categories, old_datasetmap=args.support_old_datasetmap)
    pdb_codes = list(pdb_to_probability.keys())
    print(f"Ready to sample {args.sample_n} for each of the {len(pdb_codes)} proteins from {args.path_to_pred_matrix.stem}.")
    pdb_to_sample = sample_with_multiprocessing(args.workers, pdb_codes, args.sample_n, pdb_to_probability, flat_categories)
    output_paths = save_as(
        pdb_to_sample,
        filename=f"{args.path_to_pred_matrix.stem}_temp_{args.temperature}_n_{args.sample_n}_{pdb_codes[0]}",
        mode=args.save_as,
    )
    return output_paths


def build_parser():
    parser = argparse.ArgumentParser(description="")
    parser.add_argument("--path_to_pred_matrix", type=str, help="Path to prediction matrix file ending with .csv")
    parser.add_argument("--path_to_datasetmap", default="datasetmap.txt", type=str,
                        help="Path to dataset map ending with .txt")
    parser.add_argument("--predict_rotamers", default=False, action="store_true",
                        help="Whether model outputs predictions for 338 rotamers (True) or 20 residues (False).")
    parser.add_argument("--sample_n", type=int, default=100, help="Number of samples to be drawn from the distribution.")
    parser.add_argument("--save_as", type=str, default="all", const="all", nargs="?", choices=["fasta", "json", "all"],
                        help="Whether to save as fasta and json (default: all) or either of them.")
    parser.add_argument("--workers", type=int, default=8, help="Accepted for compatibility; sampling runs on the GPU")
    parser.add_argument("--temperature", type=float, default=1,
                        help="Temperature factor to apply to softmax prediction. (default: 1.0 - unchanged)")
    parser.add_argument("--support_old_datasetmap", default=False, action="store_true",
                        help="Whether model to import from the old datasetmap (default: False)")
    parser.add_argument("--seed", type=int, default=42, help="random seed (default: 42)")
    return parser


if __name__ == "__main__":
    main_sample(build_parser().parse_args())
