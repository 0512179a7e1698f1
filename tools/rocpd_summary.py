#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (.db) outputs into the text tables kept under profiles/.

    python tools/rocpd_summary.py --kernel-trace kt_results.db [--pmc FETCH_SIZE=fetch_results.db ...]

Kernel instantiations are shared between layers (e.g. three TIMED blocks run the same
k_conv_mfma<8,4,2,4,16,0,0>), so rows are keyed by (kernel, grid, LDS bytes): that separates the
layers whose launch geometry differs.  PMC values are averaged per dispatch; FETCH_SIZE/WRITE_SIZE
are reported in the counter's own unit (KiB) and, corrected per
/opt/skills/guides/MI355X_MICROARCH.md §HBM (FETCH_SIZE under-reports wide coalesced reads by 2x on
gfx950), as bytes.
"""
import argparse
import re
import sqlite3
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\.kd$", "", name)
    m = re.match(r"(?:void )?(k_[a-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def dispatches(db):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select d.id, d.event_id, coalesce(s.display_name, s.kernel_name), d.start, d.end, d.grid_size_x, d.workgroup_size_x, "
        "d.group_segment_size, s.arch_vgpr_count, s.accum_vgpr_count, s.sgpr_count "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    return c, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel-trace")
    ap.add_argument("--pmc", action="append", default=[], help="NAME=path.db")
    ap.add_argument("--json", help="write per-kernel {avg_us, FETCH/WRITE bytes per launch} for bench.py's roofline.traffic")
    args = ap.parse_args()
    js = {}
    if args.kernel_trace:
        _, rows = dispatches(args.kernel_trace)
        agg = defaultdict(list)
        meta = {}
        for _id, _ev, name, st, en, grid, wg, lds, vg, ag, sg in rows:
            key = (short(name), grid // max(wg, 1), lds)
            agg[key].append((en - st) / 1e3)
            meta[key] = (wg, vg, ag, sg)
        total = sum(sum(v) for v in agg.values())
        print(f"# kernel trace: {len(rows)} dispatches, {total/1e3:.3f} ms total device time")
        print(f"{'kernel':58s} {'WGs':>7s} {'LDS':>7s} {'wg':>4s} {'vgpr':>5s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} "
              f"{'min_us':>10s} {'max_us':>10s} {'%':>6s}")
        for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            wg, vg, ag, sg = meta[key]
            print(f"{key[0]:58s} {key[1]:7d} {key[2]:7d} {wg:4d} {vg + ag:5d} {len(v):6d} {sum(v)/1e3:10.3f} {sum(v)/len(v):10.1f} "
                  f"{min(v):10.1f} {max(v):10.1f} {100*sum(v)/total:6.2f}")
            js.setdefault("|".join(map(str, key)), {})["avg_us"] = sum(v) / len(v)
    for spec in args.pmc:
        cname, path = spec.split("=", 1)
        c, rows = dispatches(path)
        vals = dict(c.execute("select e.event_id, sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                              "where p.name = ? group by e.event_id", (cname,)).fetchall())
        agg = defaultdict(list)
        for _id, ev, name, st, en, grid, wg, lds, *_ in rows:
            if ev in vals:
                agg[(short(name), grid // max(wg, 1), lds)].append(vals[ev])
        print(f"\n# PMC {cname} per dispatch (counter unit: KiB)")
        # calibration on k_convert_frames (known byte counts: reads n*V*C*4, writes n*V*8*4): FETCH_SIZE
        # reads exactly 1/2 of a wide coalesced stream (x2 per the guide), WRITE_SIZE reads exact (x1).
        corr = 2.0 if cname == "FETCH_SIZE" else 1.0
        print(f"{'kernel':58s} {'WGs':>7s} {'LDS':>7s} {'calls':>6s} {'avg_KiB':>14s} {'raw_MB':>10s} {'corrected_MB(x%g)' % corr:>20s}")
        for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            avg = sum(v) / len(v)
            print(f"{key[0]:58s} {key[1]:7d} {key[2]:7d} {len(v):6d} {avg:14.1f} {avg*1024/1e6:10.3f} {corr*avg*1024/1e6:20.3f}")
            js.setdefault("|".join(map(str, key)), {})[cname + "_bytes"] = corr * avg * 1024
    if args.json:
        import json
        out = [dict(kernel=k.split("|")[0].replace(" ", ""), wgs=int(k.split("|")[1]), lds=int(k.split("|")[2]), **v) for k, v in js.items()
               if k.startswith("k_conv")]
        json.dump(dict(note="per-launch averages; FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE x1; separate --pmc passes", kernels=out),
                  open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
