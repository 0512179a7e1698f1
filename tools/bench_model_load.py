import sys, time
sys.path.insert(0, "/root/repo/timed-design_amd")
from timed_hip import engine, synth, pack
import numpy as np
for name in ("timed", "timed_rotamer"):
    cfg, w = synth.TOPOLOGIES[name]()
    blob = pack.keras_to_pack(cfg, w)
    open("/tmp/m.pack", "wb").write(blob)
    for k in range(3):
        t0 = time.perf_counter(); m = engine.load_model("/tmp/m.pack"); t1 = time.perf_counter()
        x = synth.synthetic_frames(8, seed=1); p = m.predict(x); t2 = time.perf_counter()
        m.close()
        print(name, "load %.1f ms, first predict %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
