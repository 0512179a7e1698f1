"""Logits of TIMED-synth on harsh frames (sparse standard-normal voxels) under four plans against the float64 oracle: python tools/err_plans_vs_float64.py (GPU box)"""
import os, sys, numpy as np
sys.path.insert(0, "timed-design_amd"); sys.path.insert(0, ".")
from timed_hip import engine, synth
from oracle import cnn_oracle
cfg, w = synth.timed_synth(338 if '--rotamer' in sys.argv else 20)
rng = np.random.default_rng(3)
harsh = "--fixture" not in sys.argv
classes = 338 if "--rotamer" in sys.argv else 20
frames = (rng.standard_normal((8, 21, 21, 21, 6)) * (rng.random((8, 21, 21, 21, 6)) < 0.3)).astype(np.float32) if harsh else synth.synthetic_frames(8)
print("frames:", "sparse standard-normal voxels" if harsh else "synth.synthetic_frames (Gaussian splats in [0, 1], the benchmark's)", "| classes:", classes)
import copy
cfgl = copy.deepcopy(cfg)
want = None
def run(env):
    for k in ("TH_FIRST_SPLIT", "TH_WINO_SPLIT", "TH_WINOGRAD", "TH_WFUSED", "TH_FIRST_WINO"):
        os.environ.pop(k, None)
    os.environ.update(env)
    m = engine.HipFrameModel.from_keras(cfg, w)
    o = m.predict(frames, logits=True); m.close(); return o
vals = cnn_oracle.forward(cfg, w, frames, np.float64, return_all=True)
layers = cfg["config"]["layers"]; out = cfg["config"]["output_layers"][0][0]
last = next(l for l in layers if l["name"] == out)
want = vals[last["inbound_nodes"][0][0][0]]
outs = {"opt-in 7-point scheme (TH_WINOGRAD=2)": run({"TH_WINOGRAD": "2"}), "default": run({}), "first fp32": run({"TH_FIRST_SPLIT": "0"}), "all fp32 (Winograd)": run({"TH_FIRST_SPLIT": "0", "TH_WINO_SPLIT": "0"}),
        "direct": run({"TH_FIRST_SPLIT": "0", "TH_WINO_SPLIT": "0", "TH_WINOGRAD": "0", "TH_WFUSED": "0", "TH_FIRST_WINO": "0"})}
ref = want if want is not None else outs["direct"].astype(np.float64)
for k, v in outs.items():
    print("%-42s max |dlogit| vs %s %.3g  (scale %.3g)" % (k, "float64 oracle" if want is not None else "direct plan", float(np.abs(v - ref).max()), float(np.abs(ref).max())))
