#!/usr/bin/env python
"""BASELINE config 5: MC sampling, 1k sequences x 300 residues at T in {0.1, 0.5, 1.0} — the same leg bench.py reports
as `sampler` (tools/bench_legs.py sampler_config5), standalone so that it can run under rocprofv3."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "timed-design_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_legs
print(json.dumps(bench_legs.sampler_config5(0)))
