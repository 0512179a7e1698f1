#!/usr/bin/env python
"""Raw per-kernel averages of every counter in rocprofv3 rocpd .db files: python tools/pmc_raw.py [--match text] a.db b.db ..."""
import collections, re, sqlite3, sys
args = sys.argv[1:]
match = ""
if args and args[0] == "--match":
    match = args[1]; args = args[2:]
for db in args:
    c = sqlite3.connect(db)
    names = dict(c.execute("select id,name from rocpd_info_pmc"))
    disp = c.execute("select d.event_id, coalesce(s.display_name,s.kernel_name), d.start, d.end, d.grid_size_x/d.workgroup_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id").fetchall()
    vals = collections.defaultdict(dict)
    for ev, pid, v in c.execute("select event_id,pmc_id,value from rocpd_pmc_event"):
        vals[ev][names[pid]] = vals[ev].get(names[pid], 0) + v
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for ev, name, st, en, wgs in disp:
        if match and match not in name: continue
        mm = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", name)
        key = (mm.group(0) if mm else name[:40], wgs)
        for k, v in vals[ev].items(): agg[key][k].append(v)
        agg[key]["dur_us"].append((en - st) / 1e3)
    for key, d in sorted(agg.items(), key=lambda kv: -sum(kv[1]["dur_us"]))[:8]:
        m = {k: sum(v) / len(v) for k, v in d.items()}
        print(key, " ".join("%s=%.4g" % kv for kv in sorted(m.items())))
