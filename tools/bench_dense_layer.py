#!/usr/bin/env python
"""Time ONE DenseCPD dense layer  BN -> ReLU -> Conv1^3(-> mid) -> BN -> ReLU -> Conv3^3('same', -> growth)  through the
planner (the 1x1x1 bottleneck on k_conv_pw2, the growth convolution on k_conv_n16), reading a channel slice of a wider arena
like the real concat path does:
    python tools/bench_dense_layer.py side cin [mid=64] [growth=16] [frames=8192] [arena_channels=96]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import _lib, engine, synth
side, cin = int(sys.argv[1]), int(sys.argv[2])
mid = int(sys.argv[3]) if len(sys.argv) > 3 else 64
growth = int(sys.argv[4]) if len(sys.argv) > 4 else 16
n = int(sys.argv[5]) if len(sys.argv) > 5 else 8192
b = synth.KerasGraphBuilder((side, side, side, cin), seed=1)
x0 = b.input_name
x = b.conv3d(b.relu(b.batchnorm(x0)), mid, 1, padding="same")
x = b.conv3d(b.relu(b.batchnorm(x)), growth, 3, padding="same")
x = b.concat([x0, x])                      # the growth channels land in a slice of the concat arena
x = b.gap(x); x = b.softmax(b.dense(x, 20))
cfg, w = b.finish(x)
m = engine.HipFrameModel.from_keras(cfg, w)
m.set_chunk(4096)
fr = np.random.default_rng(0).random((n, side, side, side, cin), dtype=np.float32)
d_in = engine.DeviceBuffer(fr.nbytes); d_in.upload(fr)
d_out = engine.DeviceBuffer(n * 20 * 4)
m.predict_device(d_in.ptr, n, d_out.ptr)
m.profile(1)
for _ in range(3):
    m.predict_device(d_in.ptr, n, d_out.ptr)
for s in m.steps():
    if s["launches"] and s["flops"] and "conv" in s["label"]:
        ms = s["ms"] / 3
        print(json.dumps(dict(label=s["label"][:90], ms_per_4096=round(ms * 4096 / n, 4), tflops=round(s["flops"] * n / (ms * 1e-3) / 1e12, 1),
                              TBps=round(s["bytes"] * n / (ms * 1e-3) / 1e12, 2), env={k: v for k, v in os.environ.items() if k.startswith("TH_")})))
