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
