"""Host mirrors of the reference's design_utils against byte-exact outputs
captured from the reference itself (tests/golden/make_host_golden.py) and against real-h5py fixtures."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from design_utils import utils
from design_utils import sampling_utils as su
from timed_hip import h5lite, h5model, synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def hg():
    return np.load(os.path.join(G, "host_golden.npz"))


def test_rotamer_codec_matches_reference(sampler_golden):
    codec, cats, guide = utils.get_rotamer_codec(return_reduction_guide=True)
    assert cats == list(sampler_golden["rot_categories"])
    assert guide == list(sampler_golden["rot_reduction_guide"])
    assert guide == [0, 1, 4, 13, 40, 49, 50, 59, 68, 149, 158, 185, 194, 203, 230, 311, 314, 317, 320, 329]  # ref utils.py:425
    assert [int(np.argmax(codec[i])) for i in range(338)] == list(sampler_golden["rot_codec_argmax"])
    assert all(codec[i].sum() == 1 and codec[i].shape == (20,) for i in range(338))
    got = utils.compress_rotamer_predictions_to_20(sampler_golden["rot_compress_in"])
    assert np.array_equal(got, sampler_golden["rot_compress_out"])
    assert utils.compress_rotamer_predictions_to_20(np.ones((1, 338))).shape[-1] == 20  # reference tests/test_utils.py:6-11


def test_writers_byte_exact(hg, tmp_path):
    flat, probs, y = hg["flat_map"], hg["probs32"], hg["y_true"]
    n = len(flat)
    for lo, hi in ((0, 10), (10, n)):
        utils.save_outputs_to_file(list(y[lo:hi]), {0: list(probs[lo:hi])}, flat, 0, "TIMED", tmp_path)
    utils.convert_dataset_map_for_srb(flat, "TIMED", tmp_path)
    for fn in ("encoded_labels.csv", "datasetmap.txt", "TIMED.csv", "TIMED.txt"):
        assert (tmp_path / fn).read_text() == str(hg["file_" + fn]), fn
    pm16 = np.genfromtxt(tmp_path / "TIMED.csv", delimiter=",", dtype=np.float16)
    assert np.array_equal(pm16.astype(np.float32), hg["pred_matrix_f16_reread"])
    seq, prob, real, cons, consp = utils.extract_sequence_from_pred_matrix(flat, pm16, None, old_datasetmap=True)
    want = json.loads(str(hg["extract_old_json"]))
    assert seq == want["seq"] and real == want["real"] and cons is None and consp is None
    assert list(prob) == list(want["prob"])
    for k in prob:
        assert np.array_equal(np.asarray(prob[k], dtype=np.float64), np.asarray(want["prob"][k]))
    utils.save_dict_to_fasta(seq, "TIMED", tmp_path)
    utils.save_dict_to_fasta(real, "dataset", tmp_path)
    assert (tmp_path / "TIMED.fasta").read_text() == str(hg["file_TIMED.fasta"])
    assert (tmp_path / "dataset.fasta").read_text() == str(hg["file_dataset.fasta"])
    # sample.py side
    dmap = utils.load_datasetmap(tmp_path / "TIMED.txt")
    assert np.array_equal(np.asarray(dmap), hg["load_datasetmap"])
    pm64 = np.genfromtxt(tmp_path / "TIMED.csv", delimiter=",", dtype=np.float64)
    s2, p2, r2, _, _ = utils.extract_sequence_from_pred_matrix(dmap, pm64, None)
    want2 = json.loads(str(hg["extract_new_json"]))
    assert s2 == want2["seq"] and r2 == want2["real"]
    assert {k: [list(map(float, row)) for row in v] for k, v in p2.items()} == want2["prob"]


def test_consensus_and_rotamer_paths(hg):
    rc = utils.extract_sequence_from_pred_matrix(hg["nmr_flat_map"], hg["nmr_probs16"].astype(np.float16), None,
                                                 is_consensus=True)
    want = json.loads(str(hg["nmr_consensus_json"]))
    assert rc[0] == want["seq"] and rc[3] == want["consensus"]
    for k, v in rc[4].items():
        assert np.array_equal(np.asarray(v, dtype=np.float64), np.asarray(want["consensus_prob"][k]))
    _, cats = utils.get_rotamer_codec()
    cats1 = [c.split("_")[0] for c in cats]
    want = json.loads(str(hg["extract_rot_seq_json"]))
    r3 = utils.extract_sequence_from_pred_matrix(hg["flat_map"], hg["probs338"], cats, old_datasetmap=True)
    assert r3[0] == want["full"]
    one = [dict(zip(["ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU", "MET", "ASN", "PRO", "GLN",
                     "ARG", "SER", "THR", "VAL", "TRP", "TYR"], "ACDEFGHIKLMNPQRSTVWY"))[c] for c in cats1]
    r4 = utils.extract_sequence_from_pred_matrix(hg["flat_map"], hg["probs338"], one, old_datasetmap=True)
    assert r4[0] == want["one"]
    codec, _ = utils.get_rotamer_codec()
    assert np.array_equal(np.array([codec[c] for c in np.argmax(hg["probs338"], axis=1)]), hg["rot_onehot20"])


def test_save_as_byte_exact(hg, tmp_path):
    sampled = {k: [tuple(x) for x in v] for k, v in json.loads(str(hg["save_as_input_json"])).items()}
    paths = su.save_as(sampled, str(tmp_path / "TIMED_temp_0.5_n_2_1ubqA"), "all")
    assert [os.path.basename(p) for p in paths] == list(hg["save_as_paths"])
    for p in paths:
        assert open(p).read() == str(hg["file_" + os.path.basename(p)]), p
    assert [os.path.basename(p) for p in su.save_as(sampled, str(tmp_path / "x"), "fasta")] == ["x.fasta", "x_metrics.csv"]
    assert [os.path.basename(p) for p in su.save_as(sampled, str(tmp_path / "y"), "json")] == ["y.json", "y_metrics.csv"]


# ---- HDF5 reader against files written by real h5py ------------------------------------------------------
def test_h5lite_reads_aposteriori_style_dataset():
    exp = np.load(os.path.join(G, "h5_expected.npz"))
    with h5lite.File(os.path.join(G, "frames_tiny.hdf5")) as f:
        assert tuple(f.attrs["frame_dims"]) == (7, 7, 7, 5)
        assert f.attrs["voxels_as_gaussian"] is True and f.attrs["make_frame_dataset_ver"] == "2.4.0"
        assert list(f.attrs["atom_encoder"]) == ["C", "N", "O", "CA", "CB"]
        assert list(f) == ["1ubq", "2xyz_0"] and list(f["2xyz_0"].keys()) == ["A", "B"]
        d = f["1ubq"]["A"]["5"]
        assert d.shape == (7, 7, 7, 5) and d.dtype == np.float32
        assert np.array_equal(d[()], exp["g_1ubq_A_5"])
        assert np.array_equal(f["2xyz_0/B/12"][()], exp["g_2xyz_B_12"])
        assert f["2xyz_0"]["A"]["4"].attrs["label"] == "MSE"
        assert np.array_equal(f["2xyz_0"]["A"]["4"].attrs["encoded_residue"], exp["g_enc"])
        with pytest.raises(KeyError):
            f["nope"]
    with h5lite.File(os.path.join(G, "frames_tiny_bool.hdf5")) as f:
        assert f.attrs["voxels_as_gaussian"] is False
        assert f["1ubq"]["A"]["5"].dtype == np.bool_
        assert np.array_equal(f["1ubq"]["A"]["5"][()], exp["b_1ubq_A_5"])
        assert np.array_equal(f["1ubq"]["A"]["13"][()], exp["b_1ubq_A_13"])


def test_h5lite_rejects_non_hdf5(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"definitely not hdf5" * 100)
    with pytest.raises(h5lite.H5FormatError):
        h5lite.File(p)


@pytest.mark.skipif(not os.path.exists("/opt/conda/bin/python3.9"), reason="needs an interpreter with real h5py")
def test_h5lite_large_groups_and_layouts(tmp_path):
    """Multi-level group B-trees (1500 links), chunk B-trees with edge chunks, compact/contiguous data."""
    script = tmp_path / "w.py"
    script.write_text(
        "import h5py, numpy as np, sys\n"
        "r = np.random.default_rng(0)\n"
        "with h5py.File(sys.argv[1], 'w') as f:\n"
        "    g = f.create_group('big')\n"
        "    for i in range(1500): g.create_dataset(str(i), data=np.array([i, i * 2], dtype=np.int64))\n"
        "    f.create_dataset('chunked', data=r.random((37, 23, 5)), chunks=(8, 8, 5), compression='gzip', shuffle=True, fletcher32=True)\n"
        "    f.create_dataset('u8', data=r.integers(0, 255, (1000,), dtype=np.uint8), chunks=(128,))\n"
        "    f.create_dataset('scalar', data=np.float32(3.5))\n"
        "    f.attrs['names'] = [b'alpha', b'be']\n"
        "    f.attrs['cfg'] = 'x' * 40000\n"
        "    np.save(sys.argv[2], f['chunked'][()]); np.save(sys.argv[3], f['u8'][()])\n")
    h5, a, b = tmp_path / "t.h5", tmp_path / "a.npy", tmp_path / "b.npy"
    subprocess.run(["/opt/conda/bin/python3.9", str(script), str(h5), str(a), str(b)], check=True)
    with h5lite.File(h5) as f:
        keys = f["big"].keys()
        assert len(keys) == 1500 and set(keys) == {str(i) for i in range(1500)}
        assert list(f["big"]["1234"][()]) == [1234, 2468]
        assert np.array_equal(f["chunked"][()], np.load(a))
        assert np.array_equal(f["u8"][()], np.load(b))
        assert f["scalar"][()] == np.float32(3.5)
        assert [x.decode() if isinstance(x, bytes) else str(x) for x in f.attrs["names"]] == ["alpha", "be"]
        assert f.attrs["cfg"] == "x" * 40000


def test_keras_h5_roundtrip():
    cfg, weights = h5model.read_keras_h5(os.path.join(G, "keras_tiny.h5"))
    cfg2, w2 = synth.timed_synth(20, widths=(8, 16), side=7, in_channels=5, seed=11, bias_std=0.1)
    assert cfg == cfg2
    for k in w2:
        assert len(weights[k]) == len(w2[k]) and all(np.array_equal(a, b) for a, b in zip(weights[k], w2[k]))


def test_dataset_map_and_batches_from_hdf5():
    path = os.path.join(G, "frames_tiny.hdf5")
    with pytest.warns(UserWarning):
        flat, pdbs = utils.create_flat_dataset_map(path)
    assert pdbs == {"1ubq", "2xyz_0"} and len(flat) == 26
    # residues sorted numerically, not as strings (reference utils.py:367-371)
    assert [r for p, c, r, _ in flat if p == "1ubq"] == [str(i) for i in range(2, 14)]
    assert ("2xyz_0", "A", "4", "MET") in flat  # MSE -> MET through the uncommon-residue table
    X, y = utils.load_batch(path, flat[3:9])
    assert X.shape == (6, 7, 7, 7, 5) and X.dtype == np.float64 and y.shape == (6, 20)
    assert np.all(y.sum(1) == 1)
    with h5lite.File(path) as f:
        assert np.array_equal(X[0], f["1ubq"]["A"][flat[3][2]][()].astype(np.float64))
    Xb, _ = utils.load_batch(os.path.join(G, "frames_tiny_bool.hdf5"), flat[:2])
    assert Xb.dtype == np.bool_
    with pytest.raises(ValueError):
        utils.create_flat_dataset_map(path, filter_list=["1ubq"])
    with pytest.warns(UserWarning):
        flat2, _ = utils.create_flat_dataset_map(path, filter_list=["1ubq"], remove_blacklist_silently=True)
    assert all(p != "1ubq" for p, *_ in flat2)


def test_seq_metrics_shapes():
    from design_utils.analyse_utils import calculate_seq_metrics, seq_metrics_batch
    m = seq_metrics_batch(["ACDEFGHIKLMNPQRSTVWY", "KKKK", "DDDD"])
    assert m.shape == (3, 4)
    assert m[1, 0] > 3 and m[2, 0] < -3 and m[1, 1] > m[2, 1]          # lysines positive / high pI
    assert abs(m[0, 2] - 2395.7) < 2.0 and m[0, 3] == 5690 + 1280 + 120  # 20-mer mass, W+Y+C extinction
    assert calculate_seq_metrics("KKKK") == tuple(float(x) for x in m[1])


def test_frame_pack_matches_hdf5(tmp_path):
    """§8 f-1: the packed (HDF5-free) dataset returns the same frames, labels and map as the HDF5 path."""
    import warnings
    from timed_hip import framepack
    for name in ("frames_tiny.hdf5", "frames_tiny_bool.hdf5"):
        src = os.path.join(G, name)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fp = framepack.pack_dataset(src, tmp_path / name)
            flat, pdbs = utils.create_flat_dataset_map(src)
        stem = str(tmp_path / name)
        assert framepack.is_pack(stem) and framepack.is_pack(stem + ".framepack") and framepack.is_pack(stem + ".frames.npy")
        flat2, pdbs2 = utils.create_flat_dataset_map(stem)
        assert [tuple(map(str, r)) for r in flat] == [tuple(r) for r in flat2] and pdbs == pdbs2
        X, y = utils.load_batch(src, flat)
        X2, y2 = utils.load_batch(stem, flat2)
        assert X2.dtype == (np.float32 if "bool" not in name else np.uint8)
        assert np.array_equal(X.astype(np.float32), np.asarray(X2, dtype=np.float32)) and np.array_equal(y, y2)
        # arbitrary (non-contiguous, reordered) rows
        pick = [flat2[i] for i in (7, 3, 20, 4)]
        Xa, ya = utils.load_batch(stem, pick)
        Xb, yb = utils.load_batch(src, pick)
        assert np.array_equal(np.asarray(Xa, dtype=np.float32), Xb.astype(np.float32)) and np.array_equal(ya, yb)
        assert len(fp) == 26 and fp.frame_dims == (7, 7, 7, 5)
    assert not framepack.is_pack(os.path.join(G, "frames_tiny.hdf5"))


def test_legacy_rand_replays_numpys_global_generator():
    """design_utils.sampling_utils._legacy_rand(n) = np.random.rand(n) (reference sampling_utils.py:81: the uniforms come from NumPy's
    GLOBAL legacy generator): same values, and the generator continues exactly where np.random.rand would have left it — seeds,
    start positions around the 624-word block boundary (an odd position makes a pair straddle two blocks), lengths below and above
    the native threshold, a cached Gaussian left untouched"""
    from design_utils import sampling_utils as su
    for seed in (0, 1, 12345):
        for pre in (0, 1, 311, 312, 623, 624, 625, 1000):
            for n in (1, 4095, 4096, 4097, 5000, 300000):
                np.random.seed(seed)
                if pre:
                    np.random.rand(pre)
                want, after = np.random.rand(n), np.random.rand(7)
                np.random.seed(seed)
                if pre:
                    np.random.rand(pre)
                got, after2 = su._legacy_rand(n), np.random.rand(7)
                assert np.array_equal(want, got) and np.array_equal(after, after2), (seed, pre, n)
    np.random.seed(5)
    np.random.randn(1)                                    # leaves a cached Gaussian in the state
    st = np.random.get_state()
    su._legacy_rand(10000)
    assert np.random.get_state()[3:] == st[3:]
    np.random.seed(5); np.random.randn(1); a = (np.random.rand(10000), np.random.randn(3))
    np.random.seed(5); np.random.randn(1); b = (su._legacy_rand(10000), np.random.randn(3))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
