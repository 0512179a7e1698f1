#!/usr/bin/env python
"""Pin the CNN path to the real reference arithmetic: run a released Keras .h5 through TensorFlow
(`tf.keras.models.load_model(...).predict(X)`, exactly the calls at reference predict.py:121 and :142)
and compare with (a) oracle/cnn_oracle.py and (b) the HIP engine when a GPU is present.

NOT runnable in the build image (no TensorFlow, no released .h5 — SURVEY.md §8c: the CNN oracle is
"parity unpinned" against TF for that reason).  Anyone who has both can close that gap:

    pip install tensorflow==2.13.0 h5py
    python tools/validate_against_keras.py --model TIMED.h5 [--frames data.hdf5 | --synthetic 64] [--emit-fixture]
    python tools/validate_against_keras.py --synth timed|timed_rotamer|densecpd|prodconn --emit-fixture

`--synth NAME` needs no released file: the benchmark topology of that name (timed_hip/synth.py — the very configs and
weights bench.py and the golden fixtures use) is instantiated IN KERAS from its model_config
(`tf.keras.Model.from_config` + `layer.set_weights`), saved as a legacy .h5 with Keras' own writer and then treated exactly
like a released model — so one command on any machine with TensorFlow pins the oracle, the .h5 reader, the converter and
the HIP engine to TensorFlow's arithmetic.  `--dry-run` does everything that needs no TensorFlow (builds the config,
parses and packs it, evaluates the oracle on two frames, prints the plan) and exits.

Exit status 0 when max |p_keras - p_oracle| <= 1e-4 and the argmax agrees on every frame whose top-2 margin
exceeds 1e-4 (the north-star tolerance), for the oracle and, if available, for the HIP engine.
`--emit-fixture` writes tests/golden/keras_real_<model>.npz (the frames, Keras' probabilities and — when the model ends in
a Softmax layer — Keras' logits) and a copy of the .h5 beside it as keras_real_<model>.h5: tests/test_oracle_cnn.py and
tests/test_gpu_cnn.py pick every such pair up automatically and from then on the oracle and the HIP engine are pinned to
TensorFlow's own output."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", help="Keras legacy .h5 (e.g. a released TIMED model)")
    ap.add_argument("--synth", choices=["timed", "timed_rotamer", "densecpd", "prodconn"],
                    help="build this benchmark topology in Keras from timed_hip/synth.py instead of loading a released .h5")
    ap.add_argument("--dry-run", action="store_true", help="do what needs no TensorFlow (config, pack, oracle on 2 frames) and exit")
    ap.add_argument("--frames", help="aposteriori .hdf5 frame dataset; default: synthetic frames")
    ap.add_argument("--synthetic", type=int, default=64, help="number of synthetic frames when --frames is not given")
    ap.add_argument("--save-golden", help="write inputs and Keras probabilities to this .npz")
    ap.add_argument("--emit-fixture", action="store_true", help="write tests/golden/keras_real_<model>.npz + .h5 (see above)")
    ap.add_argument("--tol", type=float, default=1e-4)
    args = ap.parse_args()
    if bool(args.model) == bool(args.synth):
        ap.error("give exactly one of --model FILE.h5 and --synth NAME")

    from timed_hip import h5model, synth
    from oracle import cnn_oracle

    synth_cfg = synth_weights = None
    if args.synth:
        synth_cfg, synth_weights = synth.TOPOLOGIES[args.synth]()
    if args.dry_run:
        from timed_hip import keras_config, pack
        if args.synth:
            cfg, weights = synth_cfg, synth_weights
        else:
            cfg, weights = h5model.read_keras_h5(args.model)
        layers = keras_config.parse_keras_model(cfg, weights)
        blob = pack.keras_to_pack(cfg, weights)
        shape = layers[0].out_shape
        X = synth.synthetic_frames(2, seed=1234, side=shape[0], channels=shape[-1])
        p = cnn_oracle.forward(cfg, weights, X, np.float32)
        print(f"[dry-run] {args.synth or args.model}: {len(cfg['config']['layers'])} Keras layers -> {len(layers)} engine nodes, "
              f"{keras_config.flops_per_frame(layers) / 1e6:.1f} MFLOP/frame, pack {len(blob) / 1e6:.2f} MB, input {shape}, "
              f"{p.shape[1]} classes; oracle rows sum to {p.sum(1)}")
        print("[dry-run] with TensorFlow this run would: " +
              ("tf.keras.Model.from_config(config) + set_weights per layer, save keras_real_<name>.h5; " if args.synth else "") +
              "tf.keras.models.load_model(.h5).predict(X); compare with the oracle (and the HIP engine when a GPU is visible)"
              + ("; write tests/golden/keras_real_<name>.npz/.h5" if args.emit_fixture else ""))
        sys.exit(0)

    try:
        import tensorflow as tf
    except ImportError:
        sys.exit("TensorFlow is not installed: this tool needs the reference's own stack (tensorflow==2.13.0); "
                 "--dry-run shows what it would do")

    def top_3_cat_acc(y_true, y_pred):   # the custom metric the reference registers (predict.py:24-25, :88)
        return tf.keras.metrics.top_k_categorical_accuracy(y_true, y_pred, k=3)

    tf.keras.utils.get_custom_objects()["top_3_cat_acc"] = top_3_cat_acc
    if args.synth:
        # the benchmark topology, built by Keras itself from the same model_config the engine packs, written with Keras'
        # own legacy-HDF5 writer: from here on it is handled exactly like a released model file
        import tempfile
        built = tf.keras.Model.from_config(synth_cfg["config"])
        for layer in built.layers:
            if layer.name in synth_weights and synth_weights[layer.name]:
                layer.set_weights([np.asarray(a) for a in synth_weights[layer.name]])
        args.model = os.path.join(tempfile.mkdtemp(), f"{args.synth}_synth.h5")
        built.save(args.model, save_format="h5")
    keras_model = tf.keras.models.load_model(args.model)                 # predict.py:121
    cfg, weights = h5model.read_keras_h5(args.model)                     # our reader of the same file
    in_shape = tuple(int(d) for d in keras_model.input_shape[1:])

    if args.frames:
        from design_utils import utils
        flat, _ = utils.create_flat_dataset_map(args.frames)
        X, _ = utils.load_batch(args.frames, flat[: max(1, args.synthetic)])
    else:
        X = synth.synthetic_frames(args.synthetic, seed=1234, side=in_shape[0], channels=in_shape[-1])
    X = np.ascontiguousarray(X)
    want = np.asarray(keras_model.predict(X), dtype=np.float32)          # predict.py:142

    results = {"oracle": cnn_oracle.forward(cfg, weights, X, np.float32)}
    try:
        from timed_hip import _lib, engine
        if _lib.device_count() > 0:
            results["hip"] = engine.HipFrameModel.from_keras(cfg, weights).predict(X)
    except Exception as e:   # no library / no GPU: oracle-only validation is still meaningful
        print(f"[validate] HIP engine not available here ({e}); validating the oracle only")

    ok = True
    top2 = np.sort(want, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > args.tol
    for name, got in results.items():
        err = float(np.abs(got - want).max())
        agree = bool(np.array_equal(got.argmax(1)[clear], want.argmax(1)[clear]))
        print(f"[validate] {name:6s}: max |dp| = {err:.3e} (tolerance {args.tol:g}); argmax agrees on {int(clear.sum())} clear frames: {agree}")
        ok &= err <= args.tol and agree
    if args.emit_fixture:
        import shutil
        stem = os.path.splitext(os.path.basename(args.model))[0]
        gold = os.path.join(ROOT, "tests", "golden")
        extra = {}
        last = keras_model.layers[-1]
        if last.__class__.__name__ == "Softmax":     # logits = the tensor the final Softmax layer reads
            logits_model = tf.keras.Model(keras_model.input, last.input)
            extra["keras_logits"] = np.asarray(logits_model.predict(X), dtype=np.float32)
        np.savez_compressed(os.path.join(gold, f"keras_real_{stem}.npz"), frames=X, keras_probs=want,
                            tf_version=tf.__version__, **extra)
        shutil.copyfile(args.model, os.path.join(gold, f"keras_real_{stem}.h5"))
        print(f"[validate] wrote tests/golden/keras_real_{stem}.npz and .h5")
    if args.save_golden:
        np.savez_compressed(args.save_golden, frames=X, keras_probs=want, model=os.path.basename(args.model))
        print(f"[validate] wrote {args.save_golden}: drop it under tests/golden/ to pin the oracle to Keras")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
