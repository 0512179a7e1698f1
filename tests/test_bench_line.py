"""The stdout line of bench.py must stay parseable by the driver: one JSON line, well under 8 KB, contract keys present.
Round 3's line grew to 28.9 KB (per-kernel tables inline) and the driver's stdout tail cut it: `parsed: null`.
Canned records are the full bench records of earlier rounds kept under profiles/."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_line  # noqa: E402

CANNED = ["profiles/r03_d_bench_default.json", "profiles/r02_e_bench_default.json",
          "profiles/r03_a_bench_2ranks_1gpu_fallback.json", "profiles/r01_f_bench_default.json"]


@pytest.mark.parametrize("path", CANNED)
def test_line_is_small_valid_and_complete(path):
    full = json.load(open(os.path.join(ROOT, path)))
    s = bench_line.dumps(full, "bench_detail.json")
    assert "\n" not in s
    assert len(s.encode()) <= bench_line.MAX_LINE_BYTES < 8192
    line = json.loads(s)
    for k in bench_line.REQUIRED:
        assert k in line, k
    assert line["metric"] == full["metric"] and line["unit"] == "frames/s"
    assert abs(line["value"] - full["value"]) <= 1e-6 * full["value"]
    assert abs(line["ms_per_step"] - full["ms_per_step"]) <= 1e-5 * full["ms_per_step"]
    assert "workload" in line["config"] and "model" not in line["config"]
    if "roofline" in full:
        rl = line["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches"):
            assert k in rl, k
        assert rl["bound"] in ("hbm", "mfma") and 0 < rl["frac"] <= 1.0
        assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
        assert len(rl["kernel"]) <= 96
    if "cpu_baseline" in full:
        cb = line["cpu_baseline"]
        assert set(cb) == {"value", "unit", "cores", "kind", "sample"} and cb["kind"] in ("port", "reference")
    assert "kernels" not in line


def test_other_configs_are_scalars_only():
    full = json.load(open(os.path.join(ROOT, "profiles/r03_d_bench_default.json")))
    line = bench_line.compact(full)
    assert [o["topology"] for o in line["other_configs"]] == ["densecpd", "timed_rotamer"]
    for o in line["other_configs"]:
        assert all(not isinstance(v, (list, dict)) for v in o.values())
        assert o["frames_per_s"] > 0 and 0 < o["model_frac"] <= 1


def test_oversized_record_sheds_optional_blocks_not_contract_keys():
    full = json.load(open(os.path.join(ROOT, "profiles/r03_d_bench_default.json")))
    full["other_configs"] = full["other_configs"] * 40          # a pathological record
    s = bench_line.dumps(full)
    line = json.loads(s)
    assert len(s) <= bench_line.MAX_LINE_BYTES
    for k in bench_line.REQUIRED + ("roofline", "cpu_baseline"):
        assert k in line


def test_short_kernel_names():
    lab = "conv3d_4: conv_mfma<w4,4x2,nt4,ci16,stream,pool0> FB2 ZB5/5 rows128 lds58K [k_conv_mfma<4,4,2,4,16,2,0,7>]"
    assert bench_line.short_kernel(lab) == "conv3d_4 k_conv_mfma<4,4,2,4,16,2,0,7>"
    assert bench_line.short_kernel("dense: k_dense") == "dense k_dense"


def test_short_kernel_names_with_a_remark_in_front_of_the_bracket():
    """runtime.hip puts remarks such as '(input chunk-blocked)' in FRONT of the trailing [kernel] tag that the tools parse."""
    lab = ("conv3d_1: conv_wf<F(2,3)^2 in-plane fused in LDS, z direct; pool1> 16c x 4, K32, lds159K (16x16x4 MFMA) "
           "(input chunk-blocked) [k_conv_wf<10,10,10,1,0,0>]")
    assert bench_line.short_kernel(lab) == "conv3d_1 k_conv_wf<10,10,10,1,0,0>"


def test_workload_string_does_not_depend_on_the_number_of_ranks():
    """the N = 1 line of the driver's scaling run and the headline run name the same workload: the string is a function of the
    per-GPU job (weak scaling), and it is the one the canned round-4 record carries"""
    w = bench_line.workload_string("timed", (21, 21, 21, 6), 100000, 20)
    assert w == "timed-synth forward, 21x21x21x6 fp32 frames resident in HBM, 100000 frames per GPU per step, 20 classes, random-init weights"
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "bench_line.workload_string(model.name, (D, H, W, Cc), n, model.n_classes)" in src      # nothing rank-dependent goes in
    r4 = json.load(open(os.path.join(ROOT, "profiles/r04_k_bench_line.json")))
    assert r4["config"]["workload"].startswith("timed-synth forward, 21x21x21x6 fp32 frames resident in HBM, 100000 frames per GPU per step, 20 classes")


def test_bf16x3_steps_are_priced_on_the_bf16_pipe():
    import bench_legs
    fp32 = {"label": "conv3d_1: conv_wf<...> [k_conv_wf<10,10,10,1,0,0>]", "flops": 45.88e6}
    split = {"label": "conv3d_4: conv_wino<...; bf16x3 split operands, 6 products, fp32 accumulate> [k_wino_gemm_b3]", "flops": 69.01e6}
    assert bench_legs.step_pipe(fp32) == ("fp32", 157.3, 45.88e6)
    assert bench_legs.step_pipe(split) == ("bf16", 2500.0, 6 * 69.01e6)
    # 400 k frames/s: the two steps need 0.1167 + 0.0662 of the wall time of their pipes at peak
    f = bench_legs.pipe_time_frac([fp32, split, {"label": "k_wino_in", "flops": 0.0}], 400e3)
    assert abs(f - (400e3 * 45.88e6 / 157.3e12 + 400e3 * 6 * 69.01e6 / 2500e12)) < 1e-12
