#!/usr/bin/env python
"""Which kernels the seeded random graphs of tests/test_gpu_fuzz.py exercise (run on the GPU box)."""
import sys, collections, re
sys.path.insert(0,'.'); sys.path.insert(0,'timed-design_amd')
import importlib.util
spec = importlib.util.spec_from_file_location('fz','tests/test_gpu_fuzz.py'); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from timed_hip import engine
c = collections.Counter()
for seed in range(48):
    cfg,w,fr = fz._random_net(seed)
    m = engine.HipFrameModel.from_keras(cfg,w)
    for s in m.steps():
        lab = s['label'].split(': ',1)[1]
        c[re.sub(r'\s.*','',lab)] += 1
    m.close()
for k,v in c.most_common(): print(v, k)
