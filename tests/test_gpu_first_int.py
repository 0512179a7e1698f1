"""uint8 / bool frames on the one-piece form of the split first layer (csrc/conv_first_b3.hip, template INT):
what the reference builds for voxels_as_gaussian=False (design_utils/utils.py:518-521) and hands to Model.predict (predict.py:142).
Integers 0 .. 255 and three of the four F(2,3) points are exact in ONE bf16 piece, the sum point in two: 3 (5) products instead of 6
and no split arithmetic.  The products that remain run in the general kernel's order and the dropped ones are exact zeros there, so
the logits are BIT-IDENTICAL to the run of the same values as float32 frames and to the general kernel on the uint8 frames."""
import numpy as np
import pytest

from oracle import cnn_oracle
from timed_hip import engine, synth

pytestmark = pytest.mark.gpu


def _frames(n, cin, seed, top=255):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, top + 1, size=(n, 21, 21, 21, cin)) * (rng.random((n, 21, 21, 21, cin)) < 0.3)
    x[0, :3, :3, :3] = top                                   # a dense corner of maxima: every sum point at 2 x top
    return x.astype(np.uint8)


@pytest.mark.parametrize("cin,classes", [(6, 20), (5, 20), (6, 338)])
def test_integer_frames_give_the_bits_of_the_float32_run(gpu, monkeypatch, cin, classes):
    cfg, weights = synth.timed_synth(classes, widths=(32, 16), in_channels=cin, seed=7, bias_std=0.1)
    m = engine.HipFrameModel.from_keras(cfg, weights, device=gpu)
    assert any("k_conv_first_b3" in s["label"] for s in m.steps())
    for top, as_bool in ((255, False), (1, True), (1, False)):
        u8 = _frames(7, cin, 3 + top, top)
        x = u8.astype(bool) if as_bool else u8
        got = m.predict(x, logits=True)
        as_f32 = m.predict(u8.astype(np.float32), logits=True)
        assert got.tobytes() == as_f32.tobytes(), (top, as_bool)
        want = cnn_oracle.forward(cfg, weights, u8.astype(np.float32), np.float64, return_all=True)
        logit = [k for k in want if "global_average" in k][-1]
        scale = max(1.0, float(np.abs(want[logit]).max()))
        assert float(np.abs(got - want[logit]).max()) <= 2e-5 * scale
    m.set_chunk(3)                                            # several frames per workgroup, ragged chunks
    u8 = _frames(11, cin, 9)
    a = m.predict(u8)
    monkeypatch.setenv("TH_FIRST_INT", "0")                   # the general six-product kernel on the same uint8 frames
    g = engine.HipFrameModel.from_keras(cfg, weights, device=gpu)
    g.set_chunk(3)
    assert "TH_FIRST_INT=0" in g.knobs() and g.predict(u8).tobytes() == a.tobytes()
    m.close(); g.close()
