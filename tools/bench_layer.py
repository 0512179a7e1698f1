#!/usr/bin/env python
"""Time ONE convolution layer (as a one-layer model) through the planner: which kernel it picks and its rate.
    python tools/bench_layer.py side cin cout k [frames] [pool]      e.g.  5 256 20 3 8192"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import _lib, engine, synth
import ctypes as C
side, cin, cout, k = (int(x) for x in sys.argv[1:5])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 8192
pool = int(sys.argv[6]) if len(sys.argv) > 6 else 0
b = synth.KerasGraphBuilder((side, side, side, cin), seed=1)
x = b.conv3d(b.input_name, cout, k, padding="same")
x = b.elu(x); x = b.batchnorm(x)
if pool:
    x = b.maxpool(x, 2)
x = b.gap(x); x = b.softmax(x)
cfg, w = b.finish(x)
m = engine.HipFrameModel.from_keras(cfg, w)
m.set_chunk(4096)
lib = _lib.load()
fr = np.random.default_rng(0).random((n, side, side, side, cin), dtype=np.float32)
d_in = engine.DeviceBuffer(fr.nbytes); d_in.upload(fr)
d_out = engine.DeviceBuffer(n * cout * 4)
m.predict_device(d_in.ptr, n, d_out.ptr)
m.profile(1)
for _ in range(3):
    m.predict_device(d_in.ptr, n, d_out.ptr)
for s in m.steps():
    if s["launches"] and not s["flops"] and s["bytes"] and ("wino" in s["label"]):
        ms = s["ms"] / 3
        print(json.dumps(dict(label=s["label"], ms_per_4096=ms * 4096 / n, GBps=s["bytes"] * n / (ms * 1e-3) / 1e9, frac_of_8TBps=s["bytes"] * n / (ms * 1e-3) / 8e12)))
    if s["launches"] and s["flops"]:
        ms = s["ms"] / 3
        print(json.dumps(dict(label=s["label"], ms_per_4096=ms * 4096 / n, tflops_algo=s["flops"] * n / (ms * 1e-3) / 1e12,
                              frac=s["flops"] * n / (ms * 1e-3) / 1e12 / 157.3, env={k: v for k, v in os.environ.items() if k.startswith("TH_")})))
