"""predict.py — same command line, function signature and output files as the reference's
predict.py, with the Keras model object replaced by the HIP engine:

    reference predict.py:121   frame_model = tf.keras.models.load_model(Path(m))
    reference predict.py:142   y_pred_batch = frame_model.predict(X_batch)
    here                       frame_model = timed_hip.engine.load_model(m); frame_model.predict(X_batch)

    python3 predict.py --path_to_dataset data.hdf5 --path_to_model TIMED.h5 --path_to_output .

``--path_to_model`` takes a Keras legacy ``.h5`` (converted on the fly) or a ``.pack``.  Outputs
(reference README.md:119-131): <model>.csv, <model>.fasta, <model>.txt, dataset.fasta, datasetmap.txt,
encoded_labels.csv, plus <model>_rot.csv in rotamer mode.

Deliberate differences from the reference: the raw rotamer probabilities go to
``<model_name>_rot.csv`` (the reference's missing f-string writes a file literally called
"{model_name}_rot.csv", predict.py:123 — the UI and scripts expect the real name); the consensus
fasta honours --path_to_output.  Extra, opt-in flags: --device.
"""
import argparse
from math import ceil
from pathlib import Path

import numpy as np
from numpy import genfromtxt

from design_utils.utils import (
    convert_dataset_map_for_srb,
    create_flat_dataset_map,
    extract_sequence_from_pred_matrix,
    get_pdb_keys_to_filter,
    get_rotamer_codec,
    load_batch,
    save_consensus_probs,
    save_dict_to_fasta,
    save_outputs_to_file,
)
from timed_hip import engine, textio


def load_dataset_and_predict(
    models: list,
    dataset_path: Path,
    batch_size: int = 20,
    start_batch: int = 0,
    dataset_map_path: Path = "datasetmap.txt",
    blacklist: Path = None,
    predict_rotamers: bool = False,
    model_name_suffix: str = "",
    is_consensus: bool = False,
    path_to_output: Path = Path.cwd(),
    device: int = 0,
    frames_per_call: int = 1024,
) -> (np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray):
    """reference predict.py:28-194 — same parameters and return tuple
    (flat_dataset_map, pdb_to_sequence, pdb_to_probability, pdb_to_real_sequence, pdb_to_consensus,
    pdb_to_consensus_prob).  ``start_batch`` keeps the reference's resume semantics: batches before it
    are skipped and outputs are appended.  ``batch_size`` keeps its meaning for resuming, but consecutive
    batches are handed to the GPU together (about ``frames_per_call`` frames per launch): the per-batch appends
    of the reference concatenate to exactly the same files, and the CLI default of 12 frames per batch no longer
    costs one 1.7 ms launch sequence per 12 frames."""
    path_to_output = Path(path_to_output)
    n_classes = 338 if predict_rotamers else 20
    print(f"Running model on {n_classes} classes. Rotamer Mode is {predict_rotamers}")
    filter_pdb_list = get_pdb_keys_to_filter(blacklist) if blacklist else []
    if Path(dataset_map_path).exists():
        flat_dataset_map = genfromtxt(dataset_map_path, delimiter=",", dtype="str")
        flat_dataset_map = np.atleast_2d(flat_dataset_map)
    else:
        flat_dataset_map, _training_set_pdbs = create_flat_dataset_map(dataset_path, filter_pdb_list)
    old_datasetmap = True if len(flat_dataset_map[0]) == 4 else False
    codec, flat_categories = get_rotamer_codec() if predict_rotamers else (None, None)
    n_batches = ceil(len(flat_dataset_map) / batch_size)
    pdb_to_sequence = pdb_to_probability = pdb_to_real_sequence = pdb_to_consensus = pdb_to_consensus_prob = None
    for i, m in enumerate(models):
        model_name = (m.stem if isinstance(m, Path) else str(m)) + model_name_suffix
        frame_model = engine.load_model(Path(m), device=device)
        if frame_model.n_classes != n_classes:
            raise ValueError(f"{m}: model has {frame_model.n_classes} outputs but predict_rotamers={predict_rotamers} "
                             f"expects {n_classes}")
        model_out = path_to_output / (f"{model_name}_rot.csv" if predict_rotamers else f"{model_name}.csv")
        group = max(1, int(frames_per_call) // max(1, batch_size))    # reference batches per GPU call
        for index in range(start_batch, n_batches, group):
            current_batch_map = flat_dataset_map[index * batch_size: (index + group) * batch_size]
            X_batch, y_true_batch = load_batch(dataset_path, current_batch_map)
            y_pred_batch = frame_model.predict(X_batch)
            if predict_rotamers:
                with open(model_out, "ab") as f:
                    textio.savetxt_csv(f, y_pred_batch)    # = np.savetxt(f, y_pred_batch, delimiter=","), full precision
                current_batch = np.argmax(y_pred_batch, axis=1)
                y_pred_batch = np.array([codec[c] for c in current_batch])
            save_outputs_to_file(list(y_true_batch), {i: list(y_pred_batch)}, flat_dataset_map, i, model_name, path_to_output)
        frame_model.close()
        flat_dataset_map = np.array(flat_dataset_map)
        convert_dataset_map_for_srb(flat_dataset_map, model_name, path_to_output)
        prediction_matrix = textio.loadtxt_f16(model_out)   # = np.atleast_2d(genfromtxt(..., dtype=np.float16)), C parser
        (pdb_to_sequence, pdb_to_probability, pdb_to_real_sequence, pdb_to_consensus,
         pdb_to_consensus_prob) = extract_sequence_from_pred_matrix(
            flat_dataset_map, prediction_matrix,
            rotamers_categories=flat_categories if predict_rotamers else None,
            old_datasetmap=old_datasetmap, is_consensus=is_consensus)
        save_dict_to_fasta(pdb_to_sequence, model_name, path_to_output)
        save_dict_to_fasta(pdb_to_real_sequence, "dataset", path_to_output)
        if pdb_to_consensus:
            save_dict_to_fasta(pdb_to_consensus, model_name + "_consensus", path_to_output)
            save_consensus_probs(pdb_to_consensus_prob, model_name, path_to_output)
    return (flat_dataset_map, pdb_to_sequence, pdb_to_probability, pdb_to_real_sequence, pdb_to_consensus,
            pdb_to_consensus_prob)


def main(args):
    args.path_to_dataset = Path(args.path_to_dataset)
    args.path_to_model = Path(args.path_to_model)
    args.path_to_datasetmap = Path(args.path_to_datasetmap)
    args.path_to_output = Path(args.path_to_output)
    if not args.path_to_output.exists():
        print(f"Output directory at {args.path_to_output} does not exist. Do you want to create it? (y/n)")
        if input() == "y":
            args.path_to_output.mkdir(parents=True, exist_ok=True)
        else:
            print("Exiting...")
            exit()
    if args.path_to_blacklist:
        args.path_to_blacklist = Path(args.path_to_blacklist)
        assert args.path_to_blacklist.exists(), f"Path to blacklist at {args.path_to_blacklist} does not exists."
    assert args.path_to_model.exists(), f"Path to model at {args.path_to_model} does not exists."
    assert args.path_to_dataset.exists(), f"Path to dataset at {args.path_to_dataset} does not exists."
    assert args.batch_size > 0, f"Batch size must be higher than 0 but got {args.batch_size}"
    return load_dataset_and_predict(
        [args.path_to_model],
        args.path_to_dataset,
        batch_size=args.batch_size,
        start_batch=0,
        blacklist=args.path_to_blacklist,
        dataset_map_path=args.path_to_datasetmap,
        predict_rotamers=args.predict_rotamers,
        is_consensus=args.is_structure_nmr,
        path_to_output=args.path_to_output,
        device=args.device,
    )


def build_parser():
    parser = argparse.ArgumentParser(description="Predict with TIMED")
    parser.add_argument("--batch_size", type=int, default=12,
                        help="Number of batches of frames to predict at once (default: 12)")
    parser.add_argument("--path_to_dataset", type=str, help="Path to dataset file ending with .hdf5")
    parser.add_argument("--path_to_datasetmap", default="datasetmap.txt", type=str,
                        help="Path to dataset map ending with .txt")
    parser.add_argument("--path_to_model", type=str, help="Path to model file ending with .h5 (or .pack)")
    parser.add_argument("--path_to_blacklist", type=str, default=None,
                        help="Path to csv file containing PDBs in the training set.")
    parser.add_argument("--path_to_output", type=str, default=".",
                        help="Directory to save output files. Defaults to current working directory. If the directory "
                             "does not exist, the user will be prompted to create it.")
    parser.add_argument("--output_analysis", action="store_true", help="Whether to output analysis graphs.")
    parser.add_argument("--predict_rotamers", action="store_true",
                        help="Whether model outputs predictions for 338 rotamers (True) or 20 residues (False).")
    parser.add_argument("--is_structure_nmr", action="store_true",
                        help="Whether the structure is NMR. NMR will have different states so TIMED will try to build a consensus")
    parser.add_argument("--device", type=int, default=0, help="HIP device index (default: 0)")
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
