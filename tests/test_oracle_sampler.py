"""The sampler oracle against vectors produced by importing the reference's own sampling_utils.py
(tests/golden/make_sampler_golden.py) and against rocRAND's host-compiled Philox generator."""
import numpy as np
import pytest

from oracle import sampler_oracle as so

CASES = ["dir20_f64", "dir20_f16", "dir338_f64", "dir338_f16", "edge20"]


@pytest.mark.parametrize("name", CASES)
def test_choice_indices_match_reference(sampler_golden, name):
    g = sampler_golden
    p = g[f"probs_{name}"]
    for seed in (0, 42):
        r, want = g[f"r_{name}_s{seed}"], g[f"idx_{name}_s{seed}"]
        with np.errstate(invalid="ignore"):
            assert np.array_equal(so.choice_indices(p, r), want)
            for s in range(r.shape[0]):
                assert np.array_equal(so.choice_indices(p, r[s]), want[s])


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("t", [0.1, 0.5, 1.0, 2.0])
def test_apply_temp_matches_reference(sampler_golden, name, t):
    g = sampler_golden
    with np.errstate(all="ignore"):
        got = so.apply_temp(g[f"probs_{name}"], t)
    assert np.array_equal(got, g[f"temp_{name}_t{t}"], equal_nan=True)


def test_reference_own_temperature_test(sampler_golden):
    """reference tests/test_sampling_utils.py:47-62 restated on the oracle."""
    theo = sampler_golden["theoretical_prob"]
    assert np.allclose(so.apply_temp(theo, 1), theo)
    cold = so.apply_temp(theo, 0.01)
    assert cold.argmax() == theo.argmax() and np.isclose(cold[0, cold.argmax()], 1.0)
    assert np.allclose(so.apply_temp(theo, 100), 0.05, rtol=0.01, atol=0.01)


@pytest.mark.parametrize("seed", [0, 42, 123456789])
def test_legacy_stream(sampler_golden, seed):
    want = sampler_golden[f"mt_s{seed}"]
    assert np.array_equal(so.legacy_uniforms(seed, 2000), want)
    assert np.array_equal(so.legacy_uniforms(seed, 500, skip=700), want[700:1200])


def test_philox_matches_rocrand_host_build(sampler_golden):
    g = sampler_golden
    assert np.array_equal(so.philox_uniforms(42, 64, 0), g["philox_s42_o0"])
    seed, off = 0xDEADBEEFCAFE, 5_000_000_000
    assert np.array_equal(so.philox_uniforms(seed, 64, off), g[f"philox_s{seed}_o{off}"])
    r = so.philox_uniforms(1, 100000)
    assert r.min() > 0.0 and r.max() <= 1.0 and abs(r.mean() - 0.5) < 0.01
