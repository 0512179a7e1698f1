"""th_format_csv (host code in libtimedhip.so, no GPU needed) is byte-identical to np.savetxt(delimiter=",") —
the writer format of reference design_utils/utils.py:768-771 and predict.py:145-146."""
import io

import numpy as np
import pytest

from timed_hip import textio


def _np_bytes(a):
    buf = io.BytesIO()
    np.savetxt(buf, a, delimiter=",")
    return buf.getvalue()


def test_every_float16_value_formats_like_numpy():
    allh = np.arange(65536, dtype=np.uint16).view(np.float16).reshape(-1, 16)
    assert textio.format_csv(allh) == _np_bytes(allh)           # includes +-0, subnormals, +-inf, every NaN payload


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_full_precision_rows_format_like_numpy(dtype):
    rng = np.random.default_rng(5)
    a = rng.standard_normal((257, 7)).astype(dtype) * np.float64(10.0) ** rng.integers(-30, 30, (257, 7))
    a = a.astype(dtype)
    a[0, :4] = [0.0, -0.0, np.inf, -np.inf]
    a[1, 0] = np.nan
    a[2, :3] = [np.finfo(dtype).max, np.finfo(dtype).tiny, -np.finfo(dtype).eps]
    a[3, 0] = np.float64(5e-324) if dtype == np.float64 else np.float32(1e-45)    # smallest subnormal
    assert textio.format_csv(a) == _np_bytes(a)


def _py_lines(x):
    return "".join("%.18e\n" % v for v in np.asarray(x, dtype=np.float32).tolist()).encode()


def test_float32_fixed_point_formatter_equals_python_on_every_kind_of_value():
    """float32 values take th_fmt_e18_f32 (csrc/fmt_e18_f32.h: exact digits nine at a time from a fixed-point fraction, rounded to
    nearest / ties to even on the exact value) instead of the general double formatter — the same function the GPU kernel of
    th_format_csv_device runs.  Against Python's own '%.18e' (what np.savetxt applies): random bit patterns (every exponent,
    subnormals, NaN / inf / integers >= 2^24 through the fallback), every exponent with edge mantissas in both signs, and
    constructed TIES — odd m * 2^-k with exactly 20 significant digits, where the 19th digit is decided by round-half-even."""
    rng = np.random.default_rng(17)
    bits = rng.integers(0, 2 ** 32, 60000, dtype=np.uint64).astype(np.uint32)
    with np.errstate(all="ignore"):
        x = bits.view(np.float32)
        assert textio.format_csv(x) == _py_lines(x)
    p = np.concatenate([rng.random(30000, dtype=np.float32), (rng.random(30000) ** 8).astype(np.float32)])
    assert textio.format_csv(p) == _py_lines(p)
    sweep = [(s << 31) | (ex << 23) | man for ex in range(255) for man in (0, 1, 2, 3, 0x400000, 0x7fffff, 0x7ffffe, 0x555555, 0x2aaaaa)
             for s in (0, 1)]
    x = np.array(sweep, dtype=np.uint32).view(np.float32)
    assert textio.format_csv(x) == _py_lines(x)
    ties = [m / 2.0 ** k for k in range(1, 80) for m in range(1, 1 << 12, 2) if len(str(m * 5 ** k)) == 20]
    assert len(ties) > 2000
    x = np.array(ties, dtype=np.float64)
    assert np.array_equal(x.astype(np.float32).astype(np.float64), x)                # all of them ARE float32 values
    got = textio.format_csv(x.astype(np.float32))
    assert got == _py_lines(x)
    digits = [line.split(b"e")[0].replace(b".", b"") for line in got.split(b"\n")[:-1]]
    assert all(int(d[-1:]) % 2 == 0 for d in digits)                                  # every tie went to the even neighbour
    special = np.array([1, 10, 100, 0.5, 0.25, 9.5, 99.5, 16777215, 16777216, 33554432, 1e10, 3.4028235e38, 1e-45, 1.1754944e-38, -0.0, 0.0,
                        123456.789, 8388607.5, 4194303.75, 9.9999999e-5, 0.99999994], dtype=np.float32)
    assert textio.format_csv(special) == _py_lines(special)


def test_text_scratch_is_reused_across_calls_and_sizes():
    """textio.TextScratch: the writer's buffer kept between groups (predict.py hands one to every savetxt_csv call of a run); grows when
    a later matrix needs more room, never changes the bytes written, close() is idempotent"""
    rng = np.random.default_rng(9)
    sc = textio.TextScratch()
    for shape in ((3, 5), (1000, 338), (7, 20), (1, 1), (2000, 338)):
        a = rng.random(shape).astype(np.float32)
        b = io.BytesIO()
        textio.savetxt_csv(b, a, scratch=sc)
        assert b.getvalue() == _np_bytes(a)
        h = a.astype(np.float16)
        b = io.BytesIO()
        textio.savetxt_csv(b, h, scratch=sc)
        assert b.getvalue() == _np_bytes(h)
    sc.close(); sc.close()
    b = io.BytesIO(); textio.savetxt_csv(b, a[:2], scratch=sc)          # usable again after close()
    assert b.getvalue() == _np_bytes(a[:2])


def test_probability_matrices_and_shapes():
    rng = np.random.default_rng(6)
    p = rng.dirichlet(np.full(20, 0.3), size=3000).astype(np.float32)
    p16 = np.array(list(p), dtype=np.float16)                    # what utils.save_outputs_to_file builds
    assert textio.format_csv(p16) == _np_bytes(p16)
    rot = rng.dirichlet(np.full(338, 0.05), size=64).astype(np.float32)
    assert textio.format_csv(rot) == _np_bytes(rot)              # multithreaded path (> 20k values)
    big = rng.random((4096, 20)).astype(np.float16)
    assert textio.format_csv(big) == _np_bytes(big)
    assert textio.format_csv(np.zeros((0, 20), np.float16)) == b""
    one = rng.random(5).astype(np.float32)
    assert textio.format_csv(one) == _np_bytes(one)              # 1-D: one value per line
    with pytest.raises(TypeError):
        textio.format_csv(np.zeros((2, 2), np.int32))
    # text- and binary-mode handles
    s = io.StringIO(); textio.savetxt_csv(s, p16[:3]); assert s.getvalue().encode() == _np_bytes(p16[:3])
    b = io.BytesIO(); textio.savetxt_csv(b, p16[:3]); assert b.getvalue() == _np_bytes(p16[:3])


def test_loadtxt_f16_equals_genfromtxt(tmp_path):
    rng = np.random.default_rng(7)
    p16 = rng.dirichlet(np.full(20, 0.3), size=500).astype(np.float16)
    f = tmp_path / "m.csv"
    f.write_bytes(textio.format_csv(p16))
    want = np.genfromtxt(f, delimiter=",", dtype=np.float16)
    got = textio.loadtxt_f16(f)
    assert got.dtype == np.float16 and np.array_equal(got, want) and np.array_equal(got, p16)
    # the rotamer matrix is written at full fp32 precision and re-read as float16 (reference predict.py:145,163)
    rot = rng.dirichlet(np.full(338, 0.05), size=40).astype(np.float32)
    r = tmp_path / "rot.csv"
    r.write_bytes(textio.format_csv(rot))
    assert np.array_equal(textio.loadtxt_f16(r), np.genfromtxt(r, delimiter=",", dtype=np.float16))
    one = tmp_path / "one.csv"
    one.write_bytes(textio.format_csv(p16[:1]))
    assert textio.loadtxt_f16(one).shape == (1, 20)              # np.atleast_2d of the reference's re-read


# ---- native argmax -> letters and the dataset-map tokenizer (host code: run without a GPU) ------------------------------
def test_argmax_letters_follows_numpy_rules():
    """reference design_utils/utils.py:659 (np.argmax: first maximum, a NaN is the maximum) + :689-692"""
    rng = np.random.default_rng(0)
    letters = "ACDEFGHIKLMNPQRSTVWY"
    for dt in (np.float16, np.float32, np.float64):
        m = rng.random((5000, 20)).astype(dt)
        m[::7, 3] = m[::7].max(1)                    # ties: the first maximum wins
        m[5, 11] = np.nan
        m[6, [2, 9]] = np.nan                        # the first NaN wins
        m[7] = 0
        m[8, 0] = np.inf
        m[9] = -np.inf
        got = textio.argmax_letters(m, letters)
        want = np.array(list(letters))[np.argmax(m, axis=1)]
        assert got.dtype == np.dtype("S1") and np.array_equal(got.astype(str), want), dt
    wide = rng.random((300, 338)).astype(np.float32)
    cats = [letters[i % 20] for i in range(338)]
    assert np.array_equal(textio.argmax_letters(wide, cats).astype(str), np.array(cats)[wide.argmax(1)])
    assert textio.argmax_letters(np.empty((0, 20), np.float32), letters).shape == (0,)
    with pytest.raises(ValueError):
        textio.argmax_letters(wide, letters)


def test_string_table_reader_equals_genfromtxt(tmp_path):
    rows = [("1ubq", "A", str(i), "MET") for i in range(1, 77)] + [("2xyz_0", "B", "-5", "GLY"), ("7long_name", "AA", "1000", "TRP")]
    p = tmp_path / "datasetmap.txt"
    np.savetxt(p, np.array(rows), delimiter=",", fmt="%s")
    got = textio.read_string_table(p)
    want = np.atleast_2d(np.genfromtxt(p, delimiter=",", dtype="str"))
    assert got.dtype == want.dtype and np.array_equal(got, want)
    p.write_text("a,b,c,d")                                           # one row, no final newline
    assert np.array_equal(textio.read_string_table(p), np.atleast_2d(np.genfromtxt(p, delimiter=",", dtype="str")))
    # anything NumPy treats specially is declined (the caller falls back to NumPy)
    for text in ("a,b\n#c,d\n", "a,b\nc\n", "a,,b\n", "a,b\r\nc,d\r\n", "a, b\nc,d\n", "a,b\n\nc,d\n", "é,b\n", ""):
        p.write_text(text)
        assert textio.read_string_table(p) is None, repr(text)


def test_extract_sequences_plan_and_lazy_probabilities():
    """the vectorised / native extract_sequence_from_pred_matrix against a literal transcription of the reference's loop
    semantics (utils.py:660-692) on a map with a key that comes back later, and the lazy probability mapping"""
    from design_utils import utils
    rng = np.random.default_rng(3)
    fmap = np.array([("1abc", "A", str(i), "ALA") for i in range(4)] + [("1abc", "B", str(i), "GLY") for i in range(3)] +
                    [("1abc", "A", str(i), "TRP") for i in range(10, 12)])
    pm = rng.random((9, 20)).astype(np.float16)
    seq, prob, real, _, _ = utils.extract_sequence_from_pred_matrix(fmap, pm, None)
    letters = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    pick = letters[pm.argmax(1)]
    assert list(seq) == ["1abcA", "1abcB"] and seq["1abcA"] == "".join(pick[[0, 1, 2, 3, 7, 8]]) and seq["1abcB"] == "".join(pick[4:7])
    assert real == {"1abcA": "AAAAWW", "1abcB": "GGG"}
    assert list(prob) == ["1abcA", "1abcB"] and len(prob) == 2 and "1abcB" in prob and "zzz" not in prob
    assert prob["1abcB"] == [list(r) for r in pm[4:7]] and isinstance(prob["1abcB"][0], list)
    assert np.array_equal(prob.matrix("1abcA"), pm[[0, 1, 2, 3, 7, 8]])
    assert dict(prob.items()) == {"1abcA": [list(r) for r in pm[[0, 1, 2, 3, 7, 8]]], "1abcB": [list(r) for r in pm[4:7]]}
    with pytest.raises(KeyError):
        prob["nope"]
    # "<pdb> <count>" maps
    s2, p2, r2, _, _ = utils.extract_sequence_from_pred_matrix(np.array([("k1", "4"), ("k2", "3"), ("k1", "2")]), pm, None)
    assert s2 == {"k1": "".join(pick[[0, 1, 2, 3, 7, 8]]), "k2": "".join(pick[4:7])} and r2 == {"k1": "", "k2": ""}
