# parity + rate of the one-piece first layer on uint8 frames:  gpurun -- 'bash tools/jobs/first_int.sh'
mkdir -p gpurun_out/first
timeout 600 python -m pytest tests/test_gpu_first_int.py "tests/test_gpu_conv_sweep.py" -x -q -k "first or int" 2>&1 | tail -6
timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/first/u8_rate.txt
import os, sys, time
sys.path.insert(0, "timed-design_amd")
import numpy as np, ctypes as C
from timed_hip import _lib, engine, synth
lib = _lib.load()
cfg, w = synth.timed_synth(20)
n = 16384
d_f = engine.DeviceBuffer(n * 55566 * 4); d_p = engine.DeviceBuffer(n * 20 * 4)
_lib.check(lib.th_dev_synth_frames(0, C.c_void_p(d_f.ptr), n, 21, 6, 200, 1234))
x = d_f.download((n, 55566), np.float32)
u8 = (x > 0).astype(np.uint8); d_u = engine.DeviceBuffer(u8.nbytes); d_u.upload(u8)
for env in ({}, {"TH_FIRST_INT": "0"}):
    os.environ.update(env)
    m = engine.HipFrameModel.from_keras(cfg, w)
    for name, ptr, dt in (("f32", d_f.ptr, _lib.TH_F32), ("u8", d_u.ptr, _lib.TH_U8)):
        m.profile(0); m.predict_device(ptr, n, d_p.ptr, dtype=dt)
        t0 = time.perf_counter()
        for _ in range(3): m.predict_device(ptr, n, d_p.ptr, dtype=dt)
        dt_s = time.perf_counter() - t0
        m.profile(1); m.predict_device(ptr, n, d_p.ptr, dtype=dt)
        first = [s for s in m.steps() if "first" in s["label"]][0]
        print(env, name, "%.0f frames/s; first layer %.3f ms per 4096" % (3 * n / dt_s, first["ms"] * 4096 / n))
    m.close()
PY
