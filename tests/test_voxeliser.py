"""Voxeliser (replaces aposteriori.make_frame_dataset as called at reference ui.py:73-86).
PARITY UNPINNED against aposteriori itself (not in the reference tree): the GPU kernel is checked against the written
specification's NumPy restatement (oracle/voxel_oracle.py), and the specification against the few facts the reference
does pin — most usefully the average C-beta position in the aligned residue frame quoted at design_utils/utils.py:247.
Structure fixture: tests/golden/1ubq.pdb1.gz, the data file the reference's own tests directory holds."""
import os

import numpy as np
import pytest

from oracle import voxel_oracle
from timed_hip import pdbio, voxeliser

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
UBQ = os.path.join(G, "1ubq.pdb1.gz")


@pytest.fixture(scope="module")
def ubq():
    models = pdbio.read_pdb(UBQ)
    assert len(models) == 1 and models[0].number == 1
    return models[0]


def test_pdb_reader_on_the_reference_test_structure(ubq):
    protein = [r for r in ubq.residues if not r.hetero]
    assert len(protein) == 76 and sum(len(r.atoms) for r in protein) == 602
    assert len(ubq.residues) == 134 and all(r.name == "HOH" for r in ubq.residues if r.hetero)      # 58 waters (HETATM)
    first, last = protein[0], protein[-1]
    assert (first.name, first.chain, first.number) == ("MET", "A", "1") and (last.name, last.number) == ("GLY", "76")
    assert np.allclose(first.atoms["N"], [27.340, 24.430, 2.614]) and first.elements["CA"] == "C"
    assert "CB" not in last.atoms


def test_residue_frame_convention_reproduces_the_reference_cbeta_constant(ubq):
    """Item 3 of the specification: origin CA, +y along CA->N, C in the xy half-plane x > 0.  In THAT frame the real
    C-beta atoms of ubiquitin average to the constant the reference quotes for aposteriori's idealised C-beta
    (-0.741287356, -0.53937931, -1.224287356; design_utils/utils.py:247) — the alignment convention is the right one."""
    cbs = []
    for res in (r for r in ubq.residues if not r.hetero):
        R = voxeliser.residue_frame(res.atoms["N"], res.atoms["CA"], res.atoms["C"])
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and np.isclose(np.linalg.det(R), 1.0)
        n_loc, c_loc = R @ (res.atoms["N"] - res.atoms["CA"]), R @ (res.atoms["C"] - res.atoms["CA"])
        assert abs(n_loc[0]) < 1e-9 and abs(n_loc[2]) < 1e-9 and n_loc[1] > 1.3          # N on +y
        assert abs(c_loc[2]) < 1e-9 and c_loc[0] > 1.0                                     # C in the xy plane, x > 0
        if "CB" in res.atoms:
            cbs.append(R @ (res.atoms["CB"] - res.atoms["CA"]))
    mean = np.mean(cbs, axis=0)
    assert len(cbs) == 70 and np.linalg.norm(mean - voxeliser.CB_LOCAL) < 0.06, mean
    assert max(np.linalg.norm(c - voxeliser.CB_LOCAL) for c in cbs) < 0.45


def test_prepare_structure_and_oracle_properties(ubq):
    xyz, ch, sg, frt, rows = voxeliser.prepare_structure(ubq)
    assert xyz.shape == (76 * 5, 3) and frt.shape == (76, 12) and len(rows) == 76      # N, CA, C, O + idealised CB per residue
    assert rows[0] == ("A", "1", "MET") and rows[-1] == ("A", "76", "GLY")
    assert sorted(set(ch.tolist())) == [0, 1, 2, 3, 4]
    for gaussian in (False, True):
        fr = voxel_oracle.voxelise(xyz, ch, sg, frt[20:23], gaussian=gaussian)
        assert fr.shape == (3, 21, 21, 21, 5) and fr.dtype == (np.float32 if gaussian else np.uint8)
        # the residue's own CA sits in the central voxel of the CA channel, its idealised CB next to it
        if gaussian:
            assert np.all(fr[:, 10, 10, 10, 3] > 0.1)      # the CA itself: largest of its 27 normalised weights
            assert np.all(fr.reshape(3, -1).sum(1) <= 300.0) and np.all(fr.reshape(3, -1).sum(1) > 100.0)   # one unit of mass per atom inside
        else:
            assert np.all(fr[:, 10, 10, 10, 3] == 1) and np.all(fr[:, 9, 9, 9, 4] == 1)   # CB_LOCAL rounds to (-1, -1, -1)
            assert 100 < fr[0].sum() < 300
    # rigid motion of the whole structure leaves every frame unchanged (frames are residue-local)
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    Q *= np.sign(np.linalg.det(Q))
    moved = pdbio.Model(1, [pdbio.Residue(r.chain, r.number, r.name, {k: Q @ v + [5.0, -3.0, 11.0] for k, v in r.atoms.items()},
                                          dict(r.elements), r.hetero) for r in ubq.residues])
    xyz2, ch2, sg2, frt2, _ = voxeliser.prepare_structure(moved)
    a = voxel_oracle.voxelise(xyz, ch, sg, frt[30:32], gaussian=True)
    b = voxel_oracle.voxelise(xyz2, ch2, sg2, frt2[30:32], gaussian=True)
    assert np.abs(a - b).max() < 2e-3          # float32 coordinates: atoms near a voxel boundary move a little mass


@pytest.mark.gpu
@pytest.mark.parametrize("gaussian", [False, True])
def test_kernel_matches_the_specification(gpu, ubq, gaussian):
    xyz, ch, sg, frt, _ = voxeliser.prepare_structure(ubq)
    pick = [0, 1, 17, 40, 75]
    want = voxel_oracle.voxelise(xyz, ch, sg, frt[pick], gaussian=gaussian)
    got = voxeliser.voxelise(xyz, ch, sg, frt[pick], gaussian=gaussian, device=gpu)
    assert got.dtype == want.dtype and got.shape == want.shape
    if gaussian:
        assert np.array_equal(got != 0, want != 0)                    # the same voxels are touched
        np.testing.assert_allclose(got, want, rtol=5e-6, atol=1e-9)   # expf vs NumPy's float32 exp
    else:
        assert np.array_equal(got, want)                              # bit-exact
    # all 76 frames in one launch are the per-residue results
    full = voxeliser.voxelise(xyz, ch, sg, frt, gaussian=gaussian, device=gpu)
    assert np.array_equal(full[pick], got)


@pytest.mark.gpu
def test_frames_stay_on_the_device_for_the_cnn(gpu, ubq):
    """PDB -> frames in HBM -> th_predict_device: no HDF5, no host copy of the frames"""
    from timed_hip import engine, synth
    xyz, ch, sg, frt, _ = voxeliser.prepare_structure(ubq)
    cfg, weights = synth.timed_synth(20, widths=(8, 16), side=21, in_channels=5, seed=3)
    model = engine.HipFrameModel.from_keras(cfg, weights, device=gpu)
    n = frt.shape[0]
    d_frames = engine.DeviceBuffer(n * 21 ** 3 * 5 * 4, gpu)
    d_probs = engine.DeviceBuffer(n * 20 * 4, gpu)
    assert voxeliser.voxelise(xyz, ch, sg, frt, device=gpu, d_out=d_frames.ptr) is None
    model.predict_device(d_frames.ptr, n, d_probs.ptr)
    on_device = d_probs.download((n, 20), np.float32)
    host_frames = voxeliser.voxelise(xyz, ch, sg, frt, device=gpu)
    assert np.array_equal(d_frames.download(host_frames.shape, np.float32), host_frames)
    assert np.array_equal(model.predict(host_frames), on_device)
    np.testing.assert_allclose(on_device.sum(1), 1.0, atol=1e-5)


@pytest.mark.gpu
def test_pdb_to_frame_pack_to_predict(gpu, tmp_path):
    """the whole chain the reference runs through aposteriori + HDF5: structure file -> frames -> predict.py outputs"""
    import warnings
    import predict
    from timed_hip import pack, synth
    X, labels, flat = voxeliser.voxelise_pdb(UBQ, device=gpu)
    assert X.shape == (76, 21, 21, 21, 5) and labels.shape == (76, 20) and flat[0] == ("1ubq", "A", "1", "MET")
    assert labels.sum() == 76 and labels[0, 10] == 1          # MET is index 10 of the one-letter order ACDEFGHIKLMNPQRSTVWY
    voxeliser.write_frame_pack(tmp_path / "ubq", X, labels, flat, gaussian=True, source="1ubq.pdb1.gz")
    cfg, weights = synth.timed_synth(20, widths=(8, 16), side=21, in_channels=5, seed=3)
    mp = tmp_path / "M.pack"
    mp.write_bytes(pack.keras_to_pack(cfg, weights))
    out = tmp_path / "out"
    out.mkdir()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = predict.load_dataset_and_predict([mp], str(tmp_path / "ubq.framepack"), batch_size=20,
                                               dataset_map_path=out / "datasetmap.txt", path_to_output=out)
    assert list(res[1]) == ["1ubqA"] and len(res[1]["1ubqA"]) == 76
    real = (out / "dataset.fasta").read_text().split("\n")[1]
    assert real == "MQIFVKTLTGKTITLEVEPSDTIENVKAKIQDKEGIPPDQQRLIFAGKQLEDGRTLSDYNIQKESTLHLVLRLRGG"    # ubiquitin


@pytest.mark.gpu
def test_voxeliser_argument_checks(gpu, lib):
    import ctypes as C
    out = np.zeros(10, np.float32)
    f = np.zeros(12, np.float32)
    assert lib.th_voxelise(gpu, None, None, None, 0, f.ctypes.data, 1, 20, 21.0, 5, 1, out.ctypes.data, 0) == -1      # even side
    assert lib.th_voxelise(gpu, None, None, None, 0, f.ctypes.data, 1, 21, 21.0, 9, 1, out.ctypes.data, 0) == -4      # > 8 channels
    assert lib.th_voxelise(gpu, None, None, None, 0, None, 1, 21, 21.0, 5, 1, out.ctypes.data, 0) == -1
    # more atoms inside one frame than the kernel's list holds: an error, not silence
    n = 3000
    xyz = np.zeros((n, 3), np.float32)
    ch = np.zeros(n, np.int32)
    sg = np.ones(n, np.float32)
    frt = np.array([[1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]], np.float32)
    big = np.zeros((1, 21, 21, 21, 5), np.float32)
    assert lib.th_voxelise(gpu, xyz.ctypes.data, ch.ctypes.data, sg.ctypes.data, n, frt.ctypes.data, 1, 21, 21.0, 5, 1, big.ctypes.data, 0) == -4
    assert b"encodable atoms" in lib.th_last_error()
    # empty structure: all-zero frames
    z = voxeliser.voxelise(np.zeros((0, 3)), np.zeros(0), np.zeros(0), frt, device=gpu)
    assert z.shape == (1, 21, 21, 21, 5) and not z.any()


@pytest.mark.gpu
def test_predict_py_takes_a_pdb_file_directly(gpu, tmp_path):
    """--path_to_dataset structure.pdb1.gz: no aposteriori, no HDF5; same files as through a frame pack of the same frames"""
    import warnings
    import predict
    from timed_hip import pack, synth
    cfg, weights = synth.timed_synth(20, widths=(8, 16), side=21, in_channels=5, seed=3)
    mp = tmp_path / "M.pack"
    mp.write_bytes(pack.keras_to_pack(cfg, weights))
    X, labels, flat = voxeliser.voxelise_pdb(UBQ, device=gpu)
    voxeliser.write_frame_pack(tmp_path / "ubq", X, labels, flat, gaussian=True)
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        predict.load_dataset_and_predict([mp], UBQ, batch_size=16, dataset_map_path=a / "datasetmap.txt", path_to_output=a)
        predict.load_dataset_and_predict([mp], str(tmp_path / "ubq.framepack"), batch_size=16, dataset_map_path=b / "datasetmap.txt",
                                         path_to_output=b)
    for fn in sorted(p.name for p in a.iterdir()):
        assert (a / fn).read_bytes() == (b / fn).read_bytes(), fn
    assert (a / "M.txt").read_text().endswith("1ubqA 76\n")


@pytest.mark.gpu
def test_config1_predict_py_on_1ubq_matches_cnn_oracle(gpu, tmp_path):
    """BASELINE config 1 (predict.py on the reference's own tests/testing_files/1ubq.pdb1.gz): the probabilities predict.py
    writes (reference predict.py:142 -> utils.py:768, float16-rounded CSV) against oracle/cnn_oracle.py evaluated on the
    same 76 voxelised frames with a full-width 5-channel TIMED-synth; argmax sequence in the FASTA included."""
    import warnings
    import predict
    from oracle import cnn_oracle
    from timed_hip import pack, synth
    cfg, weights = synth.timed_synth(20, in_channels=5)
    mp = tmp_path / "TIMED5.pack"
    mp.write_bytes(pack.keras_to_pack(cfg, weights))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        predict.load_dataset_and_predict([mp], UBQ, batch_size=12, dataset_map_path=tmp_path / "datasetmap.txt", path_to_output=tmp_path)
    X, _labels, flat = voxeliser.voxelise_pdb(UBQ, device=gpu)
    assert X.shape == (76, 21, 21, 21, 5)
    ref = cnn_oracle.forward(cfg, weights, X)
    got = np.loadtxt(tmp_path / "TIMED5.csv", delimiter=",")
    assert got.shape == (76, 20)
    ref16 = ref.astype(np.float16).astype(np.float64)
    # a value within 5e-6 of a float16 rounding boundary may round the other way: at most one float16 step apart, and rare
    step = np.spacing(np.maximum(ref16, 2.0 ** -14).astype(np.float16)).astype(np.float64)
    assert np.all(np.abs(got - ref16) <= step)
    assert np.mean(got == ref16) > 0.99
    letters = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    clear = np.sort(ref, 1)[:, -1] - np.sort(ref, 1)[:, -2] > 2e-3
    fasta = (tmp_path / "TIMED5.fasta").read_text().split("\n")
    assert fasta[0] == ">1ubqA" and len(fasta[1]) == 76
    assert all(a == b for a, b, c in zip(fasta[1], "".join(letters[ref.argmax(1)]), clear) if c)


@pytest.mark.gpu
def test_voxelised_structure_as_aposteriori_layout_hdf5(gpu, tmp_path):
    """voxeliser -> .hdf5 in aposteriori's layout (timed_hip.h5write) -> predict.py: the files equal those of predicting
    straight from the PDB, and the dataset reads back through the reference-named loaders"""
    import warnings
    import predict
    from design_utils import utils
    from timed_hip import pack, synth
    X, labels, flat = voxeliser.voxelise_pdb(UBQ, device=gpu)
    h5 = tmp_path / "1ubq.hdf5"
    voxeliser.write_hdf5(h5, X, labels, flat, gaussian=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fmap, _ = utils.create_flat_dataset_map(h5)
        Xb, yb = utils.load_batch(h5, fmap)
    assert [tuple(r) for r in fmap] == flat and Xb.dtype == np.float64
    assert np.array_equal(Xb.astype(np.float32), X) and np.array_equal(yb, labels.astype(float))
    cfg, weights = synth.timed_synth(20, widths=(8, 16), side=21, in_channels=5, seed=3)
    mp = tmp_path / "M.pack"
    mp.write_bytes(pack.keras_to_pack(cfg, weights))
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        predict.load_dataset_and_predict([mp], UBQ, batch_size=16, dataset_map_path=a / "datasetmap.txt", path_to_output=a)
        predict.load_dataset_and_predict([mp], h5, batch_size=16, dataset_map_path=b / "datasetmap.txt", path_to_output=b)
    for fn in sorted(p.name for p in a.iterdir()):
        assert (a / fn).read_bytes() == (b / fn).read_bytes(), fn


APOSTERIORI_FIXTURES = sorted(f for f in os.listdir(G) if f.startswith("aposteriori_") and f.endswith(".npz"))


@pytest.mark.parametrize("name", APOSTERIORI_FIXTURES or [None])
def test_oracle_matches_frames_aposteriori_itself_wrote(name):
    """Picks up tests/golden/aposteriori_<code>.npz written by tools/validate_against_aposteriori.py --emit-fixture (needs
    aposteriori 2.4.0: not in this image).  With such a file the specification's restatement is held to aposteriori's own frames
    of the same structure — the pin row f-4 lacks.  Skipped, not passed, while no fixture exists."""
    if name is None:
        pytest.skip("no tests/golden/aposteriori_*.npz: run tools/validate_against_aposteriori.py --emit-fixture where aposteriori is installed")
    z = np.load(os.path.join(G, name))
    structure = os.path.join(G, str(z["structure"]))
    if not os.path.exists(structure):
        pytest.skip(f"{structure}: the structure of the fixture is not in tests/golden")
    encode_cb, gaussian = bool(z["encode_cb"]), bool(z["gaussian"])
    encoder = voxeliser.DEFAULT_ENCODER if encode_cb else tuple(a for a in voxeliser.DEFAULT_ENCODER if a != "CB")
    model = pdbio.read_pdb(structure)[0]
    xyz, ch, sg, frt, rows = voxeliser.prepare_structure(model, encode_cb=encode_cb, atom_encoder=encoder)
    assert [(c, n) for c, n, _l in rows] == [(str(r[1]), str(r[2])) for r in z["rows"]]
    ours = voxel_oracle.voxelise(xyz, ch, sg, frt, 21, 21.0, len(encoder), gaussian)
    assert ours.shape == z["frames"].shape
    assert float(np.abs(ours.astype(np.float64) - z["frames"].astype(np.float64)).max()) <= 1e-4
