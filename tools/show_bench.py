#!/usr/bin/env python
"""Pretty-print the per-kernel table of a bench.py JSON line: python tools/show_bench.py out.json"""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print(f, "%.0f %s  %.1f ms/step  dominant %.1f TF" % (d["value"], d["unit"], d["ms_per_step"], r.get("achieved", 0)))
    tot = sum(k["ms_total"] for k in d.get("kernels", []))
    for k in d.get("kernels", []):
        print("  %-92s %8.2f ms %5.1f%%  %6.1f TF" % (k["label"][:92], k["ms_total"], 100 * k["ms_total"] / tot, k["tflops_algo"]))
