"""Single-layer error of the direct and the Cook-Toom kernels against the float64 oracle on dense standard-normal inputs (GPU box;
test infrastructure: uses the CPU oracle).  python tests/wino_layer_error.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "timed-design_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cnn_oracle
from timed_hip import engine
import test_gpu_wino as T
for cin, cout, n in [(256, 338, 5), (128, 256, 8), (64, 128, 8)]:
    cfg, w, layer = T._one_layer(cin, cout, seed=cin + cout)
    rng = np.random.default_rng(n)
    frames = (rng.standard_normal((n, 5, 5, 5, cin)) * (rng.random((n, 5, 5, 5, cin)) < 0.5)).astype(np.float32)
    ref = cnn_oracle.forward(cfg, w, frames, np.float64, return_all=True)[layer]
    ref32 = cnn_oracle.forward(cfg, w, frames, np.float32, return_all=True)[layer]
    for wg in ("0", "1"):
        os.environ["TH_WINOGRAD"] = wg
        m = engine.HipFrameModel.from_keras(cfg, w)
        m.predict(frames)
        got = m.fetch(layer, n, (5, 5, 5, cout))
        e = np.abs(got - ref)
        print(cin, cout, "wino" if wg == "1" else "direct", "max|err| %.3e  rms %.3e  max|ref| %.2f" % (e.max(), np.sqrt((e**2).mean()), np.abs(ref).max()))
        m.close()
    e = np.abs(ref32 - ref); print(cin, cout, "numpy fp32", "max|err| %.3e rms %.3e" % (e.max(), np.sqrt((e**2).mean())))
