"""End-to-end on the GPU through the reference-named entry points: predict.py / sample.py /
design_utils.sampling_utils."""
import argparse
import json
import os

import numpy as np
import pytest

from oracle import cnn_oracle
from oracle import sampler_oracle as so

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_predict_end_to_end_from_h5_and_hdf5(gpu, tmp_path):
    import predict
    from design_utils import utils
    from timed_hip import h5model
    model_path = os.path.join(G, "keras_tiny.h5")
    data_path = os.path.join(G, "frames_tiny.hdf5")
    res = predict.load_dataset_and_predict([__import__("pathlib").Path(model_path)], data_path, batch_size=7,
                                           dataset_map_path=tmp_path / "datasetmap.txt", path_to_output=tmp_path)
    flat, pdb_to_seq, pdb_to_prob, pdb_to_real, cons, consp = res
    assert cons is None and consp is None and len(flat) == 26
    for fn in ("keras_tiny.csv", "keras_tiny.fasta", "keras_tiny.txt", "dataset.fasta", "datasetmap.txt", "encoded_labels.csv"):
        assert (tmp_path / fn).exists(), fn
    # oracle on the very same frames, in map order
    cfg, weights = h5model.read_keras_h5(model_path)
    with pytest.warns(UserWarning):
        flat_ref, _ = utils.create_flat_dataset_map(data_path)
    X, y = utils.load_batch(data_path, flat_ref)
    want = cnn_oracle.forward(cfg, weights, X)
    csv = np.genfromtxt(tmp_path / "keras_tiny.csv", delimiter=",")
    assert csv.shape == (26, 20)
    # the CSV holds float16-rounded probabilities (reference utils.py:768)
    assert np.array_equal(csv, csv.astype(np.float16).astype(np.float64))
    np.testing.assert_allclose(csv, want, atol=1e-4 + 2 ** -11, rtol=2 ** -10)
    letters = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    fasta = (tmp_path / "keras_tiny.fasta").read_text().split("\n")
    assert fasta[0] == ">1ubqA" and fasta[2] == ">2xyz_0A" and fasta[4] == ">2xyz_0B"
    got_seq = fasta[1] + fasta[3] + fasta[5]
    want_seq = "".join(letters[csv.argmax(1)])
    assert got_seq == want_seq
    assert np.array_equal(np.genfromtxt(tmp_path / "encoded_labels.csv", delimiter=","), y)
    assert (tmp_path / "keras_tiny.txt").read_text() == "ignore_uncommon False\ninclude_pdbs\n##########\n1ubqA 12\n2xyzA 3\n2xyzB 11\n"
    real = (tmp_path / "dataset.fasta").read_text().split("\n")
    three_to_one = dict(zip(["ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU", "MET", "ASN", "PRO",
                             "GLN", "ARG", "SER", "THR", "VAL", "TRP", "TYR"], "ACDEFGHIKLMNPQRSTVWY"))
    assert real[1] + real[3] + real[5] == "".join(three_to_one[r[3]] for r in flat_ref)
    # resume semantics (reference predict.py:32,54-57): restart at batch 2 appends only the tail
    # simulate a crash after two batches: keep the first 14 rows, then restart from batch 2 in the same directory
    full_csv = (tmp_path / "keras_tiny.csv").read_text()
    (tmp_path / "keras_tiny.csv").write_text("".join(full_csv.splitlines(True)[:14]))
    predict.load_dataset_and_predict([__import__("pathlib").Path(model_path)], data_path, batch_size=7, start_batch=2,
                                     dataset_map_path=tmp_path / "datasetmap.txt", path_to_output=tmp_path)
    assert (tmp_path / "keras_tiny.csv").read_text() == full_csv


def test_predict_cli_parser_matches_reference_flags():
    import predict
    import sample
    p = predict.build_parser().parse_args([])
    assert (p.batch_size, p.path_to_datasetmap, p.path_to_output, p.predict_rotamers, p.is_structure_nmr,
            p.path_to_blacklist, p.output_analysis) == (12, "datasetmap.txt", ".", False, False, None, False)
    s = sample.build_parser().parse_args([])
    assert (s.sample_n, s.save_as, s.workers, s.temperature, s.support_old_datasetmap, s.seed, s.predict_rotamers,
            s.path_to_datasetmap) == (100, "all", 8, 1, False, 42, False, "datasetmap.txt")
    assert s.rng == "numpy"                       # opt-in flag of this build: the default is the reference's stream
    assert sample.build_parser().parse_args(["--rng", "philox"]).rng == "philox"
    with pytest.raises(SystemExit):
        sample.build_parser().parse_args(["--rng", "lcg"])


@pytest.mark.parametrize("name", ["dir20_f64", "dir20_f16", "dir338_f16", "edge20"])
def test_sampling_utils_replay_reference_under_seed(gpu, sampler_golden, name):
    """np.random.seed(s) then the reference-named functions return what the reference returned."""
    from design_utils import sampling_utils as su
    g = sampler_golden
    p = g[f"probs_{name}"]
    for seed in (0, 42):
        want = g[f"idx_{name}_s{seed}"]
        np.random.seed(seed)
        with np.errstate(invalid="ignore"):
            got = np.array([su.random_choice_prob_index(p, return_seq=False) for _ in range(want.shape[0])])
        assert np.array_equal(got, want)
        if p.shape[1] == 20:
            np.random.seed(seed)
            out = su.sample_from_sequences("k", want.shape[0], {"k": [list(r) for r in p]}, None)
            assert [t[0] for t in out["k"]] == list(g[f"seq_{name}_s{seed}"])
            assert all(len(t) == 5 for t in out["k"])


def test_reference_unit_tests_on_gpu_functions(gpu, sampler_golden):
    """reference tests/test_sampling_utils.py:31-62 against our same-named functions (200k draws here;
    the 1e6-draw version runs on the fused kernel in test_gpu_sampler.py)."""
    from design_utils import sampling_utils as su
    theo = sampler_golden["theoretical_prob"]
    new = su.apply_temp_to_probs(probs=np.array(theo), t=1)
    assert np.allclose(new, theo)
    new = su.apply_temp_to_probs(probs=np.array(theo), t=0.01)
    assert np.argmax(new) == np.argmax(theo) and np.isclose(new[:, np.argmax(new)], 1.0)
    new = su.apply_temp_to_probs(probs=np.array(theo), t=100)
    assert np.allclose(np.array([1 / 20] * 20), new, rtol=0.01, atol=0.01)
    np.random.seed(1)
    rows = np.repeat(theo, 200_000, axis=0)
    idx = su.random_choice_prob_index(rows, return_seq=False)
    real = np.bincount(idx, minlength=20) / idx.size
    assert np.allclose(theo[0], real, rtol=0.02, atol=0.01)
    assert su.random_choice_prob_index(theo, return_seq=True).shape == (1,)
    # axis=0 form of the reference signature
    np.random.seed(3); a = su.random_choice_prob_index(rows[:50], return_seq=False)
    np.random.seed(3); b = su.random_choice_prob_index(rows[:50].T.copy(), axis=0, return_seq=False)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("temperature", [1.0, 0.5, 0.1])
def test_sample_cli_end_to_end(gpu, tmp_path, monkeypatch, temperature):
    import sample
    hg = np.load(os.path.join(G, "host_golden.npz"))
    (tmp_path / "TIMED.csv").write_text(str(hg["file_TIMED.csv"]))
    (tmp_path / "TIMED.txt").write_text(str(hg["file_TIMED.txt"]))
    monkeypatch.chdir(tmp_path)
    args = argparse.Namespace(path_to_pred_matrix=str(tmp_path / "TIMED.csv"), path_to_datasetmap=str(tmp_path / "TIMED.txt"),
                              predict_rotamers=False, sample_n=6, save_as="all", workers=3, temperature=temperature,
                              support_old_datasetmap=False, seed=42)
    paths = sample.main_sample(args)
    stem = f"TIMED_temp_{temperature}_n_6_1ubqA"
    assert paths == [f"{stem}.json", f"{stem}.fasta", f"{stem}_metrics.csv"]
    got = json.load(open(paths[0]))
    # expected: the reference's single-process order under np.random.seed(42)
    pm = np.genfromtxt(tmp_path / "TIMED.csv", delimiter=",", dtype=np.float64)
    q = so.apply_temp(pm, temperature) if temperature != 1 else pm
    letters = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    stream = so.legacy_uniforms(42, 6 * len(pm))
    pos, row = 0, 0
    for key, count in (("1ubqA", 9), ("2abcA", 5), ("2abcB", 4)):
        r = stream[pos:pos + 6 * count].reshape(6, count)
        idx = so.choice_indices(q[row:row + count], r)
        assert ["".join(letters[i]) for i in idx] == [s[0] for s in got[key]]      # bit-exact at every temperature
        pos += 6 * count
        row += count
    fasta = open(paths[1]).read().split("\n")
    assert fasta[0] == ">1ubqA_0" and fasta[1] == got["1ubqA"][0][0]
    assert open(paths[2]).readline().strip() == "pdb,sequence,charge,isoelectric_point,molecular_weight,molar_extinction"
    # --seed is honoured: same seed, same output; different seed, different output
    again = json.load(open(sample.main_sample(args)[0]))
    assert again == got
    args.seed = 7
    assert json.load(open(sample.main_sample(args)[0])) != got
    # --rng philox / mt19937: the uniforms are drawn on the GPU from --seed; mt19937 with the same seed IS the numpy stream
    args.seed = 42
    args.rng = "mt19937"
    assert json.load(open(sample.main_sample(args)[0])) == got
    args.rng = "philox"
    ph = json.load(open(sample.main_sample(args)[0]))
    assert ph != got and json.load(open(sample.main_sample(args)[0])) == ph
    assert [len(v) for v in ph.values()] == [len(v) for v in got.values()]


def test_predict_from_frame_pack_equals_hdf5(gpu, tmp_path):
    """§8 f-1: the HDF5-free packed dataset gives byte-identical prediction files."""
    import warnings
    from pathlib import Path
    import predict
    from timed_hip import framepack
    model_path = Path(os.path.join(G, "keras_tiny.h5"))
    src = os.path.join(G, "frames_tiny.hdf5")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        framepack.pack_dataset(src, tmp_path / "tiny")
        a, b = tmp_path / "a", tmp_path / "b"
        a.mkdir(); b.mkdir()
        predict.load_dataset_and_predict([model_path], src, batch_size=9, dataset_map_path=a / "datasetmap.txt", path_to_output=a)
        predict.load_dataset_and_predict([model_path], str(tmp_path / "tiny.framepack"), batch_size=9,
                                         dataset_map_path=b / "datasetmap.txt", path_to_output=b)
    for fn in ("keras_tiny.csv", "keras_tiny.fasta", "keras_tiny.txt", "dataset.fasta", "datasetmap.txt", "encoded_labels.csv"):
        assert (a / fn).read_bytes() == (b / fn).read_bytes(), fn


def test_predict_rotamer_mode_end_to_end(gpu, tmp_path):
    """predict_rotamers=True (reference predict.py:90,143-151): the raw 338-way matrix goes to <model>_rot.csv at full
    precision, <model>.csv holds the one-hot residue of the arg-max rotamer (float16 text), the FASTA is read off the
    rotamer matrix; a 20-class model is refused."""
    import warnings
    from pathlib import Path
    import predict
    from design_utils import utils
    from timed_hip import pack, synth, textio
    data_path = os.path.join(G, "frames_tiny.hdf5")
    cfg, weights = synth.timed_synth(338, widths=(8, 16), side=7, in_channels=5, seed=4, bias_std=0.1)
    mp = tmp_path / "ROT.pack"
    mp.write_bytes(pack.keras_to_pack(cfg, weights))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = predict.load_dataset_and_predict([mp], data_path, batch_size=9, predict_rotamers=True,
                                               dataset_map_path=tmp_path / "datasetmap.txt", path_to_output=tmp_path)
        flat, _ = utils.create_flat_dataset_map(data_path)
    X, _y = utils.load_batch(data_path, flat)
    want = cnn_oracle.forward(cfg, weights, X)
    raw = np.loadtxt(tmp_path / "ROT_rot.csv", delimiter=",")
    assert raw.shape == (26, 338)
    np.testing.assert_allclose(raw, want, atol=1e-5)
    assert np.array_equal(raw.astype(np.float32).astype(np.float64), raw)        # full fp32 precision survived the text
    codec, cats = utils.get_rotamer_codec()
    onehot = np.loadtxt(tmp_path / "ROT.csv", delimiter=",")
    assert onehot.shape == (26, 20) and np.array_equal(onehot, np.array([codec[c] for c in raw.argmax(1)]))
    fasta = (tmp_path / "ROT.fasta").read_text().split("\n")
    res_to_r = dict(zip(["ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU", "MET", "ASN", "PRO", "GLN", "ARG", "SER",
                         "THR", "VAL", "TRP", "TYR"], "ACDEFGHIKLMNPQRSTVWY"))
    seq16 = textio.loadtxt_f16(tmp_path / "ROT_rot.csv").argmax(1)                 # the reference re-reads the file as float16
    assert fasta[1] + fasta[3] + fasta[5] == "".join(res_to_r[cats[i].split("_")[0]] for i in seq16)
    assert res[1] is not None and set(res[1]) == {"1ubqA", "2xyz_0A", "2xyz_0B"}
    with pytest.raises(ValueError):
        predict.load_dataset_and_predict([Path(os.path.join(G, "keras_tiny.h5"))], data_path, batch_size=9, predict_rotamers=True,
                                         dataset_map_path=tmp_path / "datasetmap.txt", path_to_output=tmp_path)


def test_device_csv_formatter_equals_the_host_formatter(gpu):
    """th_format_csv_device (one GPU lane per float32 value, fixed 25-byte records) against th_format_csv — itself pinned to
    np.savetxt / Python's '%.18e' in tests/test_textio.py: probabilities (softmax rows of 338), every non-negative finite bit
    pattern class below 2^24 (all exponents, subnormals, ties), sizes around the 256-value workgroup and the 4-byte store tail;
    a block with a negative, NaN, infinite or large value is declined (TH_EUNSUP) and textio falls back to the host threads
    for it — same bytes either way."""
    import ctypes as C
    from timed_hip import _lib, textio
    lib = _lib.load()
    rng = np.random.default_rng(23)

    def dev(x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        x2 = x.reshape(-1, 1) if x.ndim == 1 else x
        out = np.empty(x2.size * 25 + 8, np.uint8)
        got = lib.th_format_csv_device(gpu, x2.ctypes.data_as(C.c_void_p), x2.shape[0], x2.shape[1], out.ctypes.data_as(C.c_void_p), out.size)
        return got, out[:max(got, 0)].tobytes()

    z = rng.standard_normal((1000, 338)).astype(np.float32) * 4
    rot = np.exp(z - z.max(1, keepdims=True)); rot = (rot / rot.sum(1, keepdims=True)).astype(np.float32)
    for x in (rot, rot[:1], rot[:3, :5], rot[:257, :1], rot[:1, :255], rot[:1, :257], rot[:7, :37]):
        got, text = dev(x)
        assert got == x.size * 25 and text == textio.format_csv(x)
    bits = rng.integers(0, 151 << 23, 300000, dtype=np.uint64).astype(np.uint32)          # non-negative, below 2^24
    got, text = dev(bits.view(np.float32))
    assert got == bits.size * 25 and text == textio.format_csv(bits.view(np.float32))
    ties = np.array([m / 2.0 ** k for k in range(1, 80) for m in range(1, 1 << 12, 2) if len(str(m * 5 ** k)) == 20], dtype=np.float32)
    got, text = dev(ties)
    assert got == ties.size * 25 and text == textio.format_csv(ties)
    edge = np.array([0.0, 1.0, 0.5, 1e-45, 1.1754944e-38, 16777215.0, 0.99999994, 9.9999999e-5, 8388607.5], dtype=np.float32)
    got, text = dev(edge)
    assert got == edge.size * 25 and text == textio.format_csv(edge)
    for bad in (-0.0, -1e-3, np.nan, np.inf, 16777216.0):
        x = rot[:4].copy(); x[2, 17] = bad
        got, _ = dev(x)
        assert got == _lib.TH_EUNSUP, (bad, got)
        with np.errstate(all="ignore"):
            assert textio.format_csv(x, device=gpu) == textio.format_csv(x)            # the declined block comes from the host threads
    assert textio.format_csv(rot, device=gpu) == textio.format_csv(rot)
    assert textio.format_csv(rot.astype(np.float64), device=gpu) == textio.format_csv(rot.astype(np.float64))   # float64: host only
    assert lib.th_format_csv_device_release() == 0
    got, text = dev(rot[:9])                                                             # scratch comes back after a release
    assert got == 9 * 338 * 25 and text == textio.format_csv(rot[:9])


def test_rotamer_matrix_text_is_the_same_from_gpu_and_host_formatters(gpu, tmp_path, monkeypatch):
    """predict.py --predict_rotamers formats <model>_rot.csv on the GPU (th_format_csv_device); TIMED_GPU_FORMAT=0 keeps the host
    threads: every output file byte for byte the same"""
    import warnings
    import predict
    from timed_hip import pack, synth
    data_path = os.path.join(G, "frames_tiny.hdf5")
    cfg, weights = synth.timed_synth(338, widths=(8, 16), side=7, in_channels=5, seed=4, bias_std=0.1)
    mp = tmp_path / "ROT.pack"
    mp.write_bytes(pack.keras_to_pack(cfg, weights))
    outs = {}
    for tag, env in (("gpu", "1"), ("host", "0")):
        monkeypatch.setenv("TIMED_GPU_FORMAT", env)
        out = tmp_path / tag
        out.mkdir()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            predict.load_dataset_and_predict([mp], data_path, batch_size=9, predict_rotamers=True,
                                             dataset_map_path=out / "datasetmap.txt", path_to_output=out)
        outs[tag] = {f.name: f.read_bytes() for f in sorted(out.iterdir())}
    assert outs["gpu"].keys() == outs["host"].keys() and "ROT_rot.csv" in outs["gpu"]
    for name in outs["gpu"]:
        assert outs["gpu"][name] == outs["host"][name], name
    assert outs["gpu"]["ROT_rot.csv"].count(b"\n") == 26


def test_grouping_batches_per_gpu_call_keeps_every_output_byte(gpu, tmp_path):
    """predict.py hands several reference batches to the GPU at once (frames_per_call): the per-batch appends
    concatenate to the same files, whatever the batch size and wherever a resume starts."""
    import warnings
    from pathlib import Path
    import predict
    model_path = Path(os.path.join(G, "keras_tiny.h5"))
    src = os.path.join(G, "frames_tiny.hdf5")
    outs = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, bs, fpc in (("one_by_one", 5, 5), ("grouped", 5, 1024), ("big", 26, 1024), ("odd", 3, 7)):
            d = tmp_path / name
            d.mkdir()
            predict.load_dataset_and_predict([model_path], src, batch_size=bs, frames_per_call=fpc,
                                             dataset_map_path=d / "datasetmap.txt", path_to_output=d)
            outs[name] = d
        # resume from batch 3 of 5-frame batches in a directory that already holds batches 0..2
        r = tmp_path / "resume"
        r.mkdir()
        predict.load_dataset_and_predict([model_path], src, batch_size=5, dataset_map_path=r / "datasetmap.txt", path_to_output=r)
        full = (r / "keras_tiny.csv").read_text()
        (r / "keras_tiny.csv").write_text("".join(full.splitlines(True)[:15]))
        lab = (r / "encoded_labels.csv").read_text()
        (r / "encoded_labels.csv").write_text("".join(lab.splitlines(True)[:15]))
        predict.load_dataset_and_predict([model_path], src, batch_size=5, start_batch=3, dataset_map_path=r / "datasetmap.txt",
                                         path_to_output=r)
        outs["resume"] = r
    ref = outs["one_by_one"]
    for name, d in outs.items():
        for fn in ("keras_tiny.csv", "keras_tiny.fasta", "keras_tiny.txt", "dataset.fasta", "datasetmap.txt", "encoded_labels.csv"):
            assert (d / fn).read_bytes() == (ref / fn).read_bytes(), (name, fn)


def test_frame_pack_rows_through_the_page_locked_staging_ring_keep_every_output_byte(gpu, tmp_path, monkeypatch):
    """predict.py copies the memory-mapped rows of a frame pack into a ring of page-locked slots (engine.StagingRing, slots reused
    as predictions complete) before th_predict_async reads them: 23 groups through 5 slots give the files that the mapped rows give
    (TIMED_STAGING=0), byte for byte, for float32 and uint8 packs; the ring is kept for the next call and closed by
    design_utils.utils.release_device_memory()."""
    import sys
    import warnings
    from pathlib import Path
    sys.path.insert(0, os.path.join(os.path.dirname(G), "..", "tools"))
    import bench_legs
    import predict
    from design_utils import utils as du
    from timed_hip import pack, synth
    cfg, w = synth.timed_synth(20)
    mp = tmp_path / "TIMED.pack"
    mp.write_bytes(pack.keras_to_pack(cfg, w))
    monkeypatch.setenv("TIMED_STAGING_THREADS", "3")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for gaussian in (True, False):
            stem = str(tmp_path / ("f32" if gaussian else "u8"))
            bench_legs.make_frame_pack(stem, 1111, gaussian=gaussian)
            outs = {}
            for mode in ("1", "0"):
                monkeypatch.setenv("TIMED_STAGING", mode)
                d = tmp_path / f"out_{gaussian}_{mode}"
                d.mkdir()
                predict.load_dataset_and_predict([mp], stem + ".framepack", batch_size=50, frames_per_call=50,
                                                 dataset_map_path=d / "datasetmap.txt", path_to_output=d)
                outs[mode] = d
                if mode == "1":
                    rings = list(du._STAGING_RINGS.values())
                    assert len(rings) == 1 and rings[0].enabled and sum(rings[0]._pinned) == 5, "the staging ring was not used"
                    du.release_device_memory()
                    assert not du._STAGING_RINGS and not rings[0].enabled
            for fn in ("TIMED.csv", "TIMED.fasta", "TIMED.txt", "dataset.fasta", "datasetmap.txt", "encoded_labels.csv"):
                assert (outs["1"] / fn).read_bytes() == (outs["0"] / fn).read_bytes(), (gaussian, fn)
            assert sum(1 for _ in open(outs["1"] / "TIMED.csv")) == 1111
