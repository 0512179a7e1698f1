"""Sampler kernels through the C ABI against the reference-generated golden vectors (bit-exact
indices for identical (probs, r)) and the oracle's RNG restatements."""
import numpy as np
import pytest

from oracle import sampler_oracle as so
from timed_hip import _lib, sampler

pytestmark = pytest.mark.gpu
CASES = ["dir20_f64", "dir20_f16", "dir338_f64", "dir338_f16", "edge20"]


@pytest.mark.parametrize("name", CASES)
def test_indices_bit_exact_vs_reference(gpu, sampler_golden, name):
    g = sampler_golden
    p = g[f"probs_{name}"]
    for seed in (0, 42):
        r, want = g[f"r_{name}_s{seed}"], g[f"idx_{name}_s{seed}"]
        got = sampler.sample_indices(p, r.shape[0], uniforms=r)
        assert got.dtype == np.int32 and np.array_equal(got, want)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("t", [0.5, 2.0, 1.0])
def test_temperature_bit_exact_when_power_is_exact(gpu, sampler_golden, name, t):
    """t=0.5 -> x*x, t=2 -> sqrt(x), t=1 -> x: NumPy's fast paths are IEEE-exact, and the normaliser
    follows NumPy's pairwise order, so the tempered matrix is bit-identical to the reference's."""
    g = sampler_golden
    with np.errstate(all="ignore"):
        got = sampler.apply_temperature(g[f"probs_{name}"], t)
    assert np.array_equal(got, g[f"temp_{name}_t{t}"], equal_nan=True)


@pytest.mark.parametrize("name", CASES[:4])
@pytest.mark.parametrize("t", [0.1, 0.3, 0.7, 3.0])
def test_temperature_generic_exponent_bit_exact(gpu, sampler_golden, name, t):
    """Generic exponents: the Python layer raises the rows with NumPy's own `**` (what the reference executes,
    sampling_utils.py:159) and the GPU normalises in NumPy's pairwise order -> bit-identical to the NumPy restatement
    run on THIS host.  Against the fixture (generated on the build host, whose NumPy may pick another pow loop) the
    bound is 2 ulp."""
    g = sampler_golden
    p = g[f"probs_{name}"]
    got = sampler.apply_temperature(p, t)
    assert np.array_equal(got, so.apply_temp(p, t))
    if f"temp_{name}_t{t}" in g.files:
        np.testing.assert_allclose(got, g[f"temp_{name}_t{t}"], rtol=5e-16, atol=1e-300)


def test_abi_generic_exponent_is_host_libm_pow(gpu, lib, sampler_golden):
    """Through the bare C ABI (TH_TEMPER_POW) a generic exponent is raised with libm pow() on the host, never with the
    device's pow: q equals math.pow element by element followed by NumPy's normaliser, bit for bit."""
    import ctypes as C
    import math
    p = np.ascontiguousarray(sampler_golden["probs_dir20_f16"])
    for t in (0.1, 0.7):
        powered = np.array([math.pow(x, 1.0 / t) for x in p.ravel()]).reshape(p.shape)
        want = powered / np.sum(powered, axis=1)[:, None]
        out = np.empty_like(p)
        assert lib.th_apply_temp_on(gpu, p.ctypes.data, p.shape[0], p.shape[1], t, out.ctypes.data) == 0
        assert np.array_equal(out, want)
        out2 = np.empty_like(p)
        assert lib.th_apply_temp(p.ctypes.data, p.shape[0], p.shape[1], t, out2.ctypes.data) == 0
        assert np.array_equal(out2, want)


@pytest.mark.parametrize("name", CASES)
def test_indices_bit_exact_at_low_temperature(gpu, sampler_golden, name):
    """north star: bit-exact residue indices at every T, including T = 0.1 where the CDF is steep"""
    g = sampler_golden
    p, r = g[f"probs_{name}"], g[f"r_{name}_s42"]
    for t in (0.1, 0.3):
        with np.errstate(all="ignore"):
            want = so.choice_indices(so.apply_temp(p, t), r)
            got = sampler.sample_indices(p, r.shape[0], temperature=t, uniforms=r)
        assert np.array_equal(got, want)


def test_fused_temperature_draw_consistent(gpu, sampler_golden):
    g = sampler_golden
    p = g["probs_dir20_f16"]
    r = g["r_dir20_f16_s42"]
    for t in (0.1, 0.5, 2.0):
        q = sampler.apply_temperature(p, t)
        fused = sampler.sample_indices(p, r.shape[0], temperature=t, uniforms=r)
        assert np.array_equal(fused, so.choice_indices(q, r))


def test_letters_and_reference_sequences(gpu, sampler_golden):
    g = sampler_golden
    p, r = g["probs_dir20_f64"], g["r_dir20_f64_s0"]
    idx, letters = sampler.sample_indices(p, r.shape[0], uniforms=r, letters="ACDEFGHIKLMNPQRSTVWY")
    seqs = [b"".join(row).decode() for row in letters]
    assert seqs == list(g["seq_dir20_f64_s0"])


def test_device_mt19937_replays_numpy_legacy_stream(gpu, sampler_golden):
    g = sampler_golden
    p = g["probs_dir20_f16"]
    n_res = p.shape[0]
    for seed in (0, 42, 123456789):
        idx, r = sampler.sample_indices(p, 20, rng="mt19937", seed=seed, return_uniforms=True)
        assert np.array_equal(r.ravel()[:2000], g[f"mt_s{seed}"][: min(2000, r.size)][: r.size])
        assert np.array_equal(r.ravel(), so.legacy_uniforms(seed, 20 * n_res))
        assert np.array_equal(idx, so.choice_indices(p, r))
    # continuing the stream (second PDB in the reference's loop)
    _, r2 = sampler.sample_indices(p, 3, rng="mt19937", seed=42, rng_offset=20 * n_res, return_uniforms=True)
    assert np.array_equal(r2.ravel(), so.legacy_uniforms(42, 3 * n_res, skip=20 * n_res))
    # and it reproduces the reference draw-for-draw under np.random.seed(42)
    want = g["idx_dir20_f16_s42"]
    got = sampler.sample_indices(p, want.shape[0], rng="mt19937", seed=42)
    assert np.array_equal(got, want)


def test_device_philox_matches_rocrand_restatement(gpu, sampler_golden):
    p = sampler_golden["probs_dir338_f16"]
    n_res = p.shape[0]
    idx, r = sampler.sample_indices(p, 50, rng="philox", seed=42, return_uniforms=True)
    assert np.array_equal(r.ravel(), so.philox_uniforms(42, 50 * n_res))
    assert np.array_equal(r.ravel()[:64], sampler_golden["philox_s42_o0"])
    assert np.array_equal(idx, so.choice_indices(p, r))


def test_reference_statistical_test(gpu, sampler_golden):
    """reference tests/test_sampling_utils.py:31-44: 1e6 draws recover the distribution."""
    theo = sampler_golden["theoretical_prob"]
    idx = sampler.sample_indices(theo, 1_000_000, rng="philox", seed=7)
    real = np.bincount(idx.ravel(), minlength=20) / idx.size
    assert np.isclose(real.sum(), theo.sum(), rtol=0.01)
    assert np.allclose(theo[0], real, rtol=0.01, atol=0.01)


def test_config5_shape(gpu):
    """BASELINE config 5: 1k sequences x 300 residues at T in {0.1, 0.5, 1.0}."""
    rng = np.random.default_rng(7)
    p = rng.dirichlet(np.full(20, 0.3), size=300).astype(np.float16).astype(np.float64)
    for t in (0.1, 0.5, 1.0):
        idx, r = sampler.sample_indices(p, 1000, temperature=t, rng="mt19937", seed=42, return_uniforms=True)
        q = sampler.apply_temperature(p, t) if t != 1.0 else p
        assert np.array_equal(idx, so.choice_indices(q, r))


# ---- resident sampler: every key of a run in one launch sequence ---------------------------------------------
def _keys(rng, n_cls, sizes):
    mats = [rng.dirichlet(np.full(n_cls, 0.3), size=n).astype(np.float16).astype(np.float64) for n in sizes]
    off = np.concatenate([[0], np.cumsum(sizes)])
    return mats, off


@pytest.mark.parametrize("n_cls", [20, 338])
def test_batched_keys_equal_per_key_calls(gpu, n_cls):
    rng = np.random.default_rng(11)
    sizes = [7, 1, 64, 300, 13]
    mats, off = _keys(rng, n_cls, sizes)
    n_samples = 9
    r = rng.random(n_samples * int(off[-1]))
    sm = sampler.Sampler(gpu)
    sm.load(np.concatenate(mats))
    letters = "".join(chr(33 + (i % 90)) for i in range(n_cls))
    d = sm.draw(off, n_samples, uniforms=r, letters=letters, want_uniforms=True)
    assert np.array_equal(d["uniforms"], r)
    for k, (m, idx, let) in enumerate(zip(mats, sm.split(d["idx"], off, n_samples), sm.split(d["letters"], off, n_samples))):
        rk = r[n_samples * off[k]: n_samples * off[k + 1]].reshape(n_samples, -1)
        assert np.array_equal(idx, so.choice_indices(m, rk))                      # the oracle, key by key
        assert np.array_equal(idx, sampler.sample_indices(m, n_samples, uniforms=rk))
        assert np.array_equal(let, np.array(list(letters), dtype="S1")[idx])
    # a sub-range of keys draws from the same resident rows; the draw numbering restarts at its first key
    sub = sm.draw(off[2:5], n_samples, uniforms=r[: n_samples * int(off[4] - off[2])])
    assert np.array_equal(sm.split(sub["idx"], off[2:5], n_samples)[1],
                          so.choice_indices(mats[3], r[n_samples * 64: n_samples * 364].reshape(n_samples, 300)))
    # device generators number draws the same way: one stream across keys
    for rng_name, ref in (("mt19937", so.legacy_uniforms), ("philox", so.philox_uniforms)):
        dd = sm.draw(off, n_samples, rng=rng_name, seed=42, want_uniforms=True)
        assert np.array_equal(dd["uniforms"], ref(42, n_samples * int(off[-1])))
        for k, m in enumerate(mats):
            rk = dd["uniforms"][n_samples * off[k]: n_samples * off[k + 1]].reshape(n_samples, -1)
            assert np.array_equal(sm.split(dd["idx"], off, n_samples)[k], so.choice_indices(m, rk))
    sm.close()


def test_edge_rows_wide(gpu):
    """338-wide rows take the bisection path only when their running sum is finite and non-decreasing; NaN rows, zero rows,
    rows with a negative entry and rows summing to < r must give exactly NumPy's (cumsum > r).argmax() (0 when nothing
    exceeds r — SURVEY Appendix C-5)."""
    rng = np.random.default_rng(5)
    p = rng.dirichlet(np.full(338, 0.05), size=40)
    p[3] = 0.0
    p[5, 17] = np.nan
    p[7] *= 0.5                      # sums to 0.5: half of the draws fall through to index 0
    p[9, 100] = -0.2                 # decreasing running sum
    p[11, 337] = np.inf
    r = rng.random((50, 40))
    with np.errstate(all="ignore"):
        want = so.choice_indices(p, r)
        got = sampler.sample_indices(p, 50, uniforms=r)
    assert np.array_equal(got, want)
    assert (want[:, 7] == 0).sum() > 10


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
def test_running_sum_in_input_dtype(gpu, sampler_golden, dtype):
    """np.cumsum accumulates in the array's dtype: predict() returns float16 probabilities and the reference feeds them
    unconverted to random_choice_prob_index (sampling_utils.py:82,125)."""
    rng = np.random.default_rng(3)
    p = rng.dirichlet(np.full(20, 0.3), size=500).astype(dtype)
    r = rng.random((30, 500))
    want = (p.cumsum(axis=1)[None] > r[:, :, None]).argmax(axis=2)           # the reference expression on the typed array
    assert want.dtype == np.int64 and p.cumsum(axis=1).dtype == dtype
    got = sampler.sample_indices(p, 30, uniforms=r, cum_dtype=dtype)
    assert np.array_equal(got, want)
    f64 = sampler.sample_indices(p, 30, uniforms=r)
    if dtype == np.float16:
        assert not np.array_equal(f64, want)                                   # the dtype matters: float64 sums differ


def test_sequence_metrics_on_device_match_host_restatement(gpu):
    """f-2: (charge, pI, MW, eps280) per sampled sequence from the 20-bin histogram kernel == the host restatement of
    calculate_seq_metrics (design_utils/analyse_utils.py; ampal itself is absent: parity unpinned), bit for bit — both
    accumulate in class order."""
    from design_utils import analyse_utils as au
    rng = np.random.default_rng(8)
    sizes = [300, 5, 76, 1]
    mats, off = _keys(rng, 20, sizes)
    sm = sampler.Sampler(gpu)
    sm.load(np.concatenate(mats))
    d = sm.draw(off, 25, rng="philox", seed=1, letters="ACDEFGHIKLMNPQRSTVWY", want_metrics=True)
    assert d["metrics"].shape == (4 * 25, 4)
    for k in range(4):
        seqs = [row.tobytes().decode() for row in sm.split(d["letters"], off, 25)[k]]
        want = au.seq_metrics_batch(seqs)
        got = d["metrics"][k * 25:(k + 1) * 25]
        assert np.array_equal(got[:, 1:], want[:, 1:])                          # pI grid value, mass, extinction: exact
        np.testing.assert_allclose(got[:, 0], want[:, 0], rtol=1e-14, atol=1e-15)
        assert np.array_equal(got[:, 0], want[:, 0])
    # letters outside the 20 standard residues are not counted
    d2 = sm.draw(off[:2], 3, rng="philox", seed=1, letters="XCDEFGHIKLMNPQRSTVWY", want_metrics=True)
    seqs = [row.tobytes().decode() for row in sm.split(d2["letters"], off[:2], 3)[0]]
    assert np.array_equal(d2["metrics"], au.seq_metrics_batch(seqs))
    sm.close()


def test_sequence_metrics_on_device_match_independent_oracle(gpu):
    """f-2 against a checker that is NOT product code: oracle/seqmetrics_oracle.py (scalar, Counter order, its own constant
    tables; pinned to 50-digit values in tests/test_oracle_seqmetrics.py).  Accumulation order differs, so the bound is
    order noise, not bit equality; the pI grid point must be the same unless two grid points tie."""
    from oracle import seqmetrics_oracle as sq
    rng = np.random.default_rng(18)
    sizes = [300, 76, 12, 1]
    mats, off = _keys(rng, 20, sizes)
    sm = sampler.Sampler(gpu)
    sm.load(np.concatenate(mats))
    d = sm.draw(off, 20, rng="philox", seed=3, letters="ACDEFGHIKLMNPQRSTVWY", want_metrics=True)
    for k in range(len(sizes)):
        seqs = [row.tobytes().decode() for row in sm.split(d["letters"], off, 20)[k]]
        got = d["metrics"][k * 20:(k + 1) * 20]
        for s, g in zip(seqs, got):
            c, pi, mw, ext = sq.seq_metrics(s)
            assert abs(g[0] - c) <= 1e-12 * max(1.0, abs(c)), s
            if g[1] != pi:
                _, series = sq.charge_series(s)
                a = sorted(abs(x) for x in series)
                assert a[1] - a[0] < 1e-12, (s, g[1], pi)
            assert abs(g[2] - mw) <= 1e-9 * mw and g[3] == ext, s
    sm.close()


def test_sample_with_multiprocessing_is_one_stream_in_key_order(gpu):
    """reference sampling_utils.py:164-197 / :118-125 replayed under np.random.seed: for key: for sample: rand(n_res)"""
    from design_utils import sampling_utils as su
    rng = np.random.default_rng(2)
    sizes = dict(a=12, b=300, c=1)
    p2p = {k: [list(row) for row in rng.dirichlet(np.full(20, 0.3), size=n)] for k, n in sizes.items()}
    np.random.seed(42)
    out = su.sample_with_multiprocessing(8, list(p2p), 5, p2p, None)
    stream = so.legacy_uniforms(42, 5 * sum(sizes.values()))
    letters = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    pos = 0
    for k, n in sizes.items():
        r = stream[pos:pos + 5 * n].reshape(5, n)
        pos += 5 * n
        want = ["".join(letters[i]) for i in so.choice_indices(np.array(p2p[k]), r)]
        assert [t[0] for t in out[k]] == want
        assert all(len(t) == 5 and isinstance(t[4], int) for t in out[k])
    assert list(out) == list(p2p)
    np.random.seed(42)
    one = su.sample_from_sequences("a", 5, p2p, None)
    assert one["a"] == out["a"]
    with pytest.raises(ValueError):
        su.sample_with_multiprocessing(1, ["z"], 2, {"z": []}, None)


# ---- th_sampler_run: one submission (round 5) --------------------------------------------------------------------------------
@pytest.mark.parametrize("n_cls,cum", [(20, np.float64), (20, np.float16), (338, np.float32), (338, np.float64)])
def test_one_submission_run_equals_load_plus_draw(gpu, n_cls, cum):
    """th_sampler_run (rows + offsets + letters up in one copy, k_cumsum_draw + k_seq_metrics, one page-locked block back) against
    th_sampler_load + th_sampler_draw on the same rows and uniforms: indices, letters and metrics bit for bit — ragged keys (1 ..
    300 residues: several keys inside one 8-row workgroup, keys across workgroup boundaries), float16-rounded rows that sum to
    less than 1 (fall-through to 0), an all-zero row and a NaN row, running sums in float16 / float32 / float64"""
    rng = np.random.default_rng(n_cls + np.dtype(cum).itemsize)
    sizes = [3, 1, 300, 7, 76, 2]
    mats, off = _keys(rng, n_cls, sizes)
    rows = np.concatenate(mats).astype(np.float16).astype(np.float64)
    rows[5] = 0.0
    rows[11] = np.nan
    letters = "".join("ACDEFGHIKLMNPQRSTVWY"[i % 20] for i in range(n_cls))
    n_s = 9
    r = rng.random(n_s * rows.shape[0])
    sm = sampler.Sampler(gpu)
    sm.load(rows, cum_dtype=cum)
    a = sm.draw(off, n_s, uniforms=r, letters=letters, want_idx=True, want_metrics=True)
    a = {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in a.items()}
    b = sm.run(rows, off, n_s, uniforms=r, letters=letters, want_idx=True, want_letters=True, want_metrics=True, cum_dtype=cum)
    assert np.array_equal(b["idx"], a["idx"]) and np.array_equal(b["letters"], a["letters"])
    assert np.array_equal(b["metrics"], a["metrics"], equal_nan=True)
    # the device generators number their draws the same way in both paths
    for mode, seed in (("philox", 12345), ("mt19937", 7)):
        a2 = sm.draw(off, n_s, rng=mode, seed=seed, letters=letters)
        b2 = sm.run(rows, off, n_s, rng=mode, seed=seed, letters=letters, want_metrics=False, cum_dtype=cum)
        assert np.array_equal(np.array(b2["idx"]), a2["idx"]) and np.array_equal(np.array(b2["letters"]), a2["letters"]), mode
    # parts that are not requested are absent, the others unchanged
    c = sm.run(rows, off, n_s, uniforms=r, letters=letters, want_idx=False, want_letters=True, want_metrics=True, cum_dtype=cum)
    assert c["idx"] is None and np.array_equal(c["letters"], a["letters"]) and np.array_equal(c["metrics"], a["metrics"], equal_nan=True)
    d = sm.run(rows, off, n_s, uniforms=r, want_idx=True, want_letters=False, want_metrics=False, cum_dtype=cum)
    assert d["letters"] is None and d["metrics"] is None and np.array_equal(d["idx"], a["idx"])
    sm.close()


def test_run_rejects_bad_arguments(gpu):
    rng = np.random.default_rng(0)
    rows = rng.dirichlet(np.full(20, 0.3), size=10)
    sm = sampler.Sampler(gpu)
    with pytest.raises(_lib.TimedHipError):
        sm.run(rows, [0, 4], 2, rng="philox", seed=1, letters="ACDEFGHIKLMNPQRSTVWY")          # keys do not cover the rows
    with pytest.raises(_lib.TimedHipError):
        sm.run(rows, [0, 10, 10], 2, rng="philox", seed=1, letters="ACDEFGHIKLMNPQRSTVWY")     # an empty key
    with pytest.raises(ValueError):
        sm.run(rows, [0, 10], 2, uniforms=np.zeros(3), letters="ACDEFGHIKLMNPQRSTVWY")
    with pytest.raises(ValueError):
        sm.run(rows, [0, 10], 2, rng="philox", seed=1, want_metrics=True)                      # metrics without letters
    sm.close()


@pytest.mark.parametrize("mode", ["philox", "mt19937"])
def test_device_generators_from_the_reference_entry_points(gpu, mode):
    """sample_with_multiprocessing(..., rng=, seed=) — the keyword sample.py's --rng passes: uniforms drawn ON the device (rocRAND
    Philox4x32-10, or MT19937(seed) = np.random.seed(seed); np.random.rand), nothing generated or uploaded by the host.  The
    sequences are the inverse-CDF draws of exactly those uniforms (checked with the oracle against the generator's own stream),
    the result depends on the seed only, and NumPy's global generator is left alone."""
    from design_utils import sampling_utils as su
    rng = np.random.default_rng(5)
    sizes = dict(a=17, b=300, c=1)
    p2p = {k: rng.dirichlet(np.full(20, 0.3), size=n).astype(np.float16).astype(np.float64) for k, n in sizes.items()}
    np.random.seed(99)
    before = np.random.get_state()[1].copy()
    out = su.sample_with_multiprocessing(8, list(p2p), 6, p2p, None, rng=mode, seed=4242)
    assert np.array_equal(np.random.get_state()[1], before)
    su.reset_device_rng()
    out = su.sample_with_multiprocessing(8, list(p2p), 6, p2p, None, rng=mode, seed=4242)
    # a second call with the same seed CONTINUES the (rng, seed) stream — like two calls on the global NumPy generator — and
    # reset_device_rng() (sample.py's --seed) starts it over
    second = su.sample_with_multiprocessing(8, ["b"], 2, p2p, None, rng=mode, seed=4242)
    su.reset_device_rng()
    again = su.sample_with_multiprocessing(8, list(p2p), 6, p2p, None, rng=mode, seed=4242)
    other = su.sample_with_multiprocessing(8, list(p2p), 6, p2p, None, rng=mode, seed=4243)
    assert out == again and out != other
    total = 6 * sum(sizes.values())
    more = total + 2 * sizes["b"]
    if mode == "mt19937":
        tail = so.legacy_uniforms(4242, more)[total:]
    else:
        tail = sampler.sample_indices(np.full((more, 2), 0.5), 1, rng="philox", seed=4242, return_uniforms=True)[1].ravel()[total:]
    want2 = ["".join(np.array(list("ACDEFGHIKLMNPQRSTVWY"))[i]) for i in so.choice_indices(p2p["b"], tail.reshape(2, sizes["b"]))]
    assert [t[0] for t in second["b"]] == want2
    if mode == "mt19937":
        stream = so.legacy_uniforms(4242, total)
    else:
        stream = sampler.sample_indices(np.full((total, 2), 0.5), 1, rng="philox", seed=4242, return_uniforms=True)[1].ravel()
    letters = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    pos = 0
    for k, n in sizes.items():
        r = stream[pos:pos + 6 * n].reshape(6, n)
        pos += 6 * n
        want = ["".join(letters[i]) for i in so.choice_indices(p2p[k], r)]
        assert [t[0] for t in out[k]] == want, k
    with pytest.raises(ValueError):
        su.sample_with_multiprocessing(8, list(p2p), 6, p2p, None, rng="xorshift")


def test_raw_mt19937_words_mode_is_np_random_rand(gpu):
    """TH_RNG_MT_WORDS: the host walks the MT19937 recurrence only (th_mt19937_words), the draw kernel tempers the raw words and
    forms genrand_res53 — the doubles are np.random.rand's, the generator's state afterwards is NumPy's, and a run drawn this way
    equals the run drawn from the doubles.  Start positions around the 624-word block boundary, odd positions (a pair straddles two
    blocks), lengths above and below the replay threshold."""
    from design_utils import sampling_utils as su
    rng = np.random.default_rng(1)
    rows = rng.dirichlet(np.full(20, 0.3), size=37).astype(np.float16).astype(np.float64)
    letters = "ACDEFGHIKLMNPQRSTVWY"
    sm = sampler.Sampler(gpu)
    for seed, burn, n_s in ((0, 0, 120), (5, 311, 200), (42, 623, 130), (7, 1, 111)):
        np.random.seed(seed)
        np.random.rand(burn) if burn else None
        if burn % 2 == 0 and burn:
            np.random.randint(0, 10)              # an odd word position: the next double's pair starts on an odd index
        state = np.random.get_state()
        want = np.random.rand(n_s * 37)
        after = np.random.get_state()
        np.random.set_state(state)
        words = su._legacy_words(n_s * 37)
        now = np.random.get_state()
        assert now[2] == after[2] and np.array_equal(now[1], after[1])
        assert np.random.rand() == (np.random.set_state(after), np.random.rand())[1]
        a = sm.run(rows, [0, 37], n_s, uniforms=want, rng="host", letters=letters)
        a = {k: np.array(v) for k, v in a.items() if isinstance(v, np.ndarray)}
        b = sm.run(rows, [0, 37], n_s, uniforms=words, rng="mt_words", letters=letters)
        assert np.array_equal(b["idx"], a["idx"]) and np.array_equal(b["letters"], a["letters"])
    sm.close()
