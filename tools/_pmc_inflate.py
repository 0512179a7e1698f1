import collections, re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
names = dict(c.execute("select id,name from rocpd_info_pmc"))
disp = c.execute("select d.event_id, coalesce(s.display_name,s.kernel_name), d.start,d.end,d.grid_size_x/d.workgroup_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id").fetchall()
vals = collections.defaultdict(dict)
for ev, pid, v in c.execute("select event_id,pmc_id,value from rocpd_pmc_event"):
    vals[ev][names[pid]] = vals[ev].get(names[pid], 0) + v
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for ev, name, st, en, wgs in disp:
    if "inflate" not in name and "lz_res" not in name: continue
    key = (name[-40:], wgs)
    for k, v in vals[ev].items(): agg[key][k].append(v)
    agg[key]["dur_us"].append((en - st) / 1e3)
for key, d in agg.items():
    m = {k: sum(v) / len(v) for k, v in d.items()}
    print(key, {k: "%.4g" % v for k, v in m.items()})
