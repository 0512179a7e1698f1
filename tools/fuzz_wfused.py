#!/usr/bin/env python
"""Soak of tests/test_gpu_conv_wfused.py::test_wfused_random_blocks over many more seeds (GPU box):
    python tools/fuzz_wfused.py            (seeds 10..129; prints the worst relative error and the failing seeds)"""
import sys, os, importlib.util
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
spec = importlib.util.spec_from_file_location("tw", os.path.join(ROOT, "tests", "test_gpu_conv_wfused.py")); tw = importlib.util.module_from_spec(spec); spec.loader.exec_module(tw)
from oracle import cnn_oracle
bad=[]; worst=0
for seed in range(10, 130):
    cfg, w, fr = tw._random_block_net(seed)
    want = cnn_oracle.forward(cfg, w, fr, np.float32)
    got, labels = tw._run(cfg, w, fr, chunk=1 + seed % 4)
    rel = float(np.abs(got-want).max())/max(1.0,float(np.abs(want).max())); worst=max(worst,rel)
    if not rel <= 2e-5 or not any("k_conv_wf" in l for l in labels): bad.append((seed, rel))
print("worst", worst, "bad", bad)
