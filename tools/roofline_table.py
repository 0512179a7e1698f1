#!/usr/bin/env python
"""Per-kernel roofline table from a rocprofv3 kernel trace alone: every plan step of a topology with its kernel, its average
launch duration over REPS passes of one chunk, the FLOPs / bytes of ONE launch (the planner's per-frame figures x chunk) and
the resulting fraction of the fp32-MFMA peak (157.3 TFLOP/s) or of the HBM peak (8 TB/s) — so that every `frac` quoted in
DESIGN.md / the bench line can be recomputed from profiles/ without the bench.

    python tools/roofline_table.py [--chunk 4096] [--reps 5] timed densecpd timed_rotamer      (on the GPU box)

Runs tools/pmc_child.py under `rocprofv3 --kernel-trace` (no counters), splits the dispatch stream on the k_synth_frames
marker launches and matches the [k_...]-tagged plan steps to dispatches in order."""
import argparse, json, os, re, shutil, sqlite3, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_TF, PEAK_GBS = 157.3, 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunk", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("topologies", nargs="*", default=["timed"])
    args = ap.parse_args()
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    env = dict(os.environ, TMPDIR="/tmp", PMC_CHILD_REPS=str(args.reps), PMC_CHILD_COSTS="1")
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        r = subprocess.run([rocprof, "--kernel-trace", "-d", td, "-o", "kt", "--", sys.executable, os.path.join(ROOT, "tools", "pmc_child.py"),
                            str(args.chunk), *args.topologies], capture_output=True, text=True, cwd="/tmp", env=env, timeout=600)
        line = next((l for l in r.stdout.splitlines() if l.startswith("{")), None)
        dbs = [os.path.join(dp, f) for dp, _d, fn in os.walk(td) for f in fn if f.endswith(".db")]
        if r.returncode or line is None or not dbs:
            sys.exit("rocprofv3 --kernel-trace failed: " + (r.stderr or r.stdout)[-400:])
        info = json.loads(line)
        c = sqlite3.connect(dbs[0])
        disp = c.execute("select coalesce(s.display_name, s.kernel_name), d.end - d.start from rocpd_kernel_dispatch d "
                         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    segments, cur = [], None
    for name, ns in disp:
        if "k_synth_frames" in name:
            cur = []
            segments.append(cur)
        elif cur is not None and "k_clip01" not in name:
            cur.append((re.sub(r"\(anonymous namespace\)::", "", name), ns / 1e3))
    per = 1 + args.reps                                # warm-up pass + reps measured passes per topology
    assert len(segments) == per * len(args.topologies), (len(segments), per, args.topologies)
    for t, topo in enumerate(args.topologies):
        labels, costs = info[topo], info[topo + "/costs"]
        sums = [[] for _ in labels]
        for seg in segments[t * per + 1:(t + 1) * per]:
            pos = 0
            for i, label in enumerate(labels):
                m = re.search(r"\[(k_[a-z0-9_]+)(<[^>]*>)?\]", label)
                if not m:
                    continue
                want = m.group(1) + (m.group(2) or "").replace(",", ", ")
                while pos < len(seg) and want not in seg[pos][0]:
                    pos += 1
                if pos < len(seg):
                    sums[i].append(seg[pos][1])
                    pos += 1
        total = sum(sum(v) / len(v) for v in sums if v)
        print(f"# {topo}: one chunk of {args.chunk} frames, average of {args.reps} passes; matched kernels {total / 1e3:.3f} ms per chunk "
              f"= {args.chunk / total * 1e6:,.0f} frames/s")
        print(f"{'plan step':40s} {'avg_us':>9s} {'share':>6s} {'GFLOP/launch':>13s} {'TFLOP/s':>8s} {'of peak':>8s} {'MB/launch':>10s} {'GB/s':>8s} {'of 8000':>8s}  bound")
        for label, (fl, _ef, by), v in zip(labels, costs, sums):
            if not v:
                continue
            us = sum(v) / len(v)
            split = "bf16x3" in label            # six bf16 piece products per multiply-add: priced on the bf16 pipe (2500 TFLOP/s dense)
            peak = 2500.0 if split else PEAK_TF
            tf = (6.0 if split else 1.0) * fl * args.chunk / (us * 1e-6) / 1e12
            gbs = by * args.chunk / (us * 1e-6) / 1e9
            bound = "mfma" if fl and by and fl / by >= PEAK_TF * 1e12 / (PEAK_GBS * 1e9) else "hbm"
            short = label.split(":", 1)[0] + " " + (re.search(r"\[([^\]]+)\]\s*$", label).group(1))
            print(f"{short[:40]:40s} {us:9.1f} {100 * us / total:5.1f}% {(6.0 if split else 1.0) * fl * args.chunk / 1e9:13.2f} {tf:8.1f} {tf / peak:8.3f} "
                  f"{by * args.chunk / 1e6:10.1f} {gbs:8.0f} {gbs / PEAK_GBS:8.3f}  {bound}{' (bf16 pipe, peak 2500: 6 products per fp32 multiply-add)' if split else ''}")
        print()


if __name__ == "__main__":
    main()
