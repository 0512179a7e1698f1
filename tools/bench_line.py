"""The ONE stdout line of bench.py, kept small.

bench.py gathers a large record (per-kernel tables for three topologies, e2e legs, sampler legs).  The driver
reads only the tail of stdout, so the contract line must stay well under 8 KB: `compact()` keeps the contract
keys and scalars, everything else goes to the side file (`bench_detail.json`) and to stderr.
tests/test_bench_line.py guards the size and the required keys on canned records.
"""
from __future__ import annotations

import json
import re

MAX_LINE_BYTES = 4096

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")


def workload_string(topology: str, dims, frames_per_gpu: int, n_classes: int) -> str:
    """`config.workload` of the bench line.  A function of the per-GPU job only — never of the number of ranks — so that the N = 1
    line of a scaling run (SCALE_rNN.json) names the same workload as the headline run (BENCH_rNN.json): scaling is weak, every
    rank runs this job on its own GPU (tests/test_bench_line.py holds the two equal)."""
    d, h, w, c = (int(v) for v in dims)
    return (f"{topology}-synth forward, {d}x{h}x{w}x{c} fp32 frames resident in HBM, {int(frames_per_gpu)} frames per GPU per step, "
            f"{int(n_classes)} classes, random-init weights")


def _r(x, nd=4):
    """Round floats to `nd` significant digits (ints and None pass through)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float(f"{float(x):.{nd}g}")
    except (TypeError, ValueError):
        return x


def short_kernel(label: str) -> str:
    """'conv3d_4: conv_mfma<...> FB2 ... [k_conv_mfma<4,4,2,4,16,2,0,7>]' -> 'conv3d_4 k_conv_mfma<4,4,2,4,16,2,0,7>'."""
    if not label:
        return label
    layer = label.split(":", 1)[0].strip()
    m = re.search(r"\[([^\]]+)\]\s*$", label)
    kern = m.group(1) if m else label.split(":", 1)[-1].strip().split(" ")[0]
    return f"{layer} {kern}"[:96]


def _roofline(rl: dict) -> dict:
    out = {"bound": rl.get("bound"), "achieved": _r(rl.get("achieved"), 5), "peak": rl.get("peak"), "unit": rl.get("unit"),
           "frac": _r(rl.get("frac")), "traffic": rl.get("traffic"), "kernel": short_kernel(rl.get("kernel", "")),
           "avg_launch_ms": _r(rl.get("avg_launch_ms"), 5), "launches": rl.get("launches")}
    if rl.get("pipe"):
        out["pipe"] = rl["pipe"]
    for k in ("algorithmic_bytes", "traffic_frames", "algorithmic_flops_per_launch", "frames_per_launch"):
        if rl.get(k) is not None:
            out[k] = rl[k]
    if rl.get("share_of_device_time") is not None:
        out["share_of_device_time"] = _r(rl["share_of_device_time"], 3)
    return out


def _cpu(cb: dict, sample_chars: int = 120) -> dict:
    return {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
            "sample": (cb.get("sample") or "")[:sample_chars]}


def _other(o: dict) -> dict:
    mr, rl = o.get("model_roofline") or {}, o.get("roofline") or {}
    if "(" in (o.get("topology") or ""):           # a non-default plan of a topology listed above: its rate and its distance only
        return {"topology": o.get("topology"), "frames_per_s": _r(o.get("frames_per_s"), 5),
                "max_dlogit_vs_default": _r(o.get("max_abs_dlogit_vs_default_plan"), 3)}
    out = {"topology": o.get("topology"), "frames": o.get("frames"), "frames_per_s": _r(o.get("frames_per_s"), 5),
           "model_frac": _r(mr.get("frac")), "dominant": short_kernel(rl.get("kernel", "")), "dominant_frac": _r(rl.get("frac"))}
    if mr.get("traffic") and mr.get("algorithmic_bytes"):
        out["traffic_over_algorithmic"] = _r(mr["traffic"] / mr["algorithmic_bytes"], 3)
    if o.get("cpu_baseline"):
        out["cpu_frames_per_s"] = _r(o["cpu_baseline"].get("value"))
    if o.get("max_abs_dlogit_vs_default_plan") is not None:
        out["max_dlogit_vs_default"] = _r(o["max_abs_dlogit_vs_default_plan"], 3)
    return out


_E2E_KEYS = ("device_resident_fps", "device_resident_u8_fps", "th_predict_sync_pageable_fps", "th_predict_async_pinned_fps",
             "predict_py_framepack_f32_fps", "predict_py_framepack_sparse_f32_fps", "predict_py_framepack_sparse_f32_pcie_bytes_per_frame",
             "predict_py_framepack_u8_fps", "predict_py_rotamer_fps",
             "predict_py_hdf5_gzip_f64_first_call_fps", "predict_py_hdf5_gzip_f64_fps", "predict_py_hdf5_cold_process_fps",
             "config1_predict_py_from_pdb_s",
             "config1_cpu_oracle_forward_s")


def _sampler(s: dict) -> dict:
    out = {"n_residues": s.get("n_residues"), "n_samples": s.get("n_samples")}
    for t, v in (s.get("temperatures") or {}).items():
        out[f"T{t}"] = {"api_ms": _r(v.get("api_ms")), "api_philox_ms": _r(v.get("api_philox_ms")), "kernel_ms": _r(v.get("kernel_ms")),
                        "seq_per_s": _r(v.get("api_sequences_per_s")), "cpu_numpy_ms": _r(v.get("cpu_numpy_ms")),
                        "bit_exact": v.get("indices_bit_exact_vs_oracle")}
    if s.get("api_breakdown_ms"):
        out["api_breakdown_ms"] = {k: _r(v, 3) for k, v in s["api_breakdown_ms"].items()}
    return out


def compact(full: dict, detail_path: str | None = None) -> dict:
    """The driver-facing record: contract keys + roofline + cpu_baseline + scalars of the secondary legs."""
    line = {k: full[k] for k in REQUIRED if k in full}
    line["value"] = _r(full["value"], 7)
    line["ms_per_step"] = _r(full["ms_per_step"], 6)
    cfg = dict(full.get("config") or {})
    for k in ("algo_mflop_per_frame", "exec_mflop_per_frame", "kernel_mflop_per_frame"):
        if k in cfg:
            cfg[k] = _r(cfg[k], 6)
    if isinstance(cfg.get("exchange"), str):
        cfg["exchange"] = cfg["exchange"][:120]
    if isinstance(cfg.get("arithmetic"), str):
        cfg["arithmetic"] = cfg["arithmetic"][:160]
    if isinstance(cfg.get("workload"), str):
        cfg["workload"] = cfg["workload"][:200]
    line["config"] = cfg
    for k in ("schema", "model_tflops", "model_kernel_tflops_fp32_equiv", "model_pipe_time_frac", "model_frac_of_fp32_mfma_peak", "model_direct_equiv_tflops"):
        if k in full:
            line[k] = _r(full[k])
    if "hbm" in full:
        line["hbm"] = {k: _r(v) if isinstance(v, float) and k in ("achieved_GBps", "frac") else v for k, v in full["hbm"].items()}
    if "roofline" in full:
        line["roofline"] = _roofline(full["roofline"])
    if "cpu_baseline" in full:
        line["cpu_baseline"] = _cpu(full["cpu_baseline"])
    if "other_configs" in full:
        line["other_configs"] = [_other(o) for o in full["other_configs"]]
    if "e2e" in full:
        line["e2e"] = {k: _r(full["e2e"][k]) for k in _E2E_KEYS if k in full["e2e"]}
    if "sampler" in full:
        line["sampler"] = _sampler(full["sampler"])
    if "pmc" in full and full["pmc"].get("error"):
        line["pmc_error"] = str(full["pmc"]["error"])[:120]
    if "extras_wall_s" in full:
        line["extras_wall_s"] = _r(full["extras_wall_s"], 3)
    if detail_path:
        line["detail"] = detail_path
    # over the budget: first the non-default-plan legs and the sampler's breakdown, then whole optional blocks — never a contract key
    if len(json.dumps(line)) > MAX_LINE_BYTES and "other_configs" in line:
        line["other_configs"] = [o for o in line["other_configs"] if "(" not in (o.get("topology") or "")]
    if len(json.dumps(line)) > MAX_LINE_BYTES and "sampler" in line:
        line["sampler"].pop("api_breakdown_ms", None)
    for drop in ("sampler", "e2e", "other_configs", "hbm"):
        if len(json.dumps(line)) <= MAX_LINE_BYTES:
            break
        line.pop(drop, None)
    return line


def dumps(full: dict, detail_path: str | None = None) -> str:
    s = json.dumps(compact(full, detail_path))
    assert "\n" not in s and len(s) <= MAX_LINE_BYTES, f"bench line is {len(s)} bytes"
    return s
