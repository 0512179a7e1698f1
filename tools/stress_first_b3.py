import os, sys, numpy as np
sys.path.insert(0, "timed-design_amd")
from timed_hip import engine, synth
cfg, w = synth.timed_synth(20)
rng = np.random.default_rng(3)
frames = (rng.standard_normal((3000, 21, 21, 21, 6)) * (rng.random((3000, 21, 21, 21, 6)) < 0.3)).astype(np.float32)
m = engine.HipFrameModel.from_keras(cfg, w)
assert any("k_conv_first_b3" in s["label"] for s in m.steps())
outs = []
for chunk in (4096, 1000, 257):
    m.set_chunk(chunk)
    for rep in range(3):
        outs.append(m.predict(frames, logits=True))
m.close()
for o in outs[1:]:
    assert np.array_equal(o, outs[0]), float(np.abs(o - outs[0]).max())
os.environ["TH_FIRST_SPLIT"] = "0"; os.environ["TH_WINO_SPLIT"] = "0"
m = engine.HipFrameModel.from_keras(cfg, w)
ref = m.predict(frames, logits=True)
m.close()
print("bit-identical across 9 runs / 3 chunk sizes; max |dlogit| vs all-fp32 plan %.3g of scale %.3g" % (float(np.abs(outs[0] - ref).max()), float(np.abs(ref).max())))
