#!/usr/bin/env python
"""Per-kernel averages of every counter found in rocprofv3 rocpd .db files (conv kernels only)."""
import collections, re, sqlite3, sys
MIN_US = 100
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    names = dict(c.execute("select id,name from rocpd_info_pmc"))
    disp = c.execute("select d.event_id, coalesce(s.display_name,s.kernel_name), d.start,d.end,d.grid_size_x/d.workgroup_size_x, d.group_segment_size from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id").fetchall()
    vals = collections.defaultdict(dict)
    for ev, pid, v in c.execute("select event_id,pmc_id,value from rocpd_pmc_event"):
        vals[ev][names[pid]] = vals[ev].get(names[pid], 0) + v
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for ev, name, st, en, wgs, lds in disp:
        if "conv" not in name and "wino" not in name: continue
        key = (re.sub(r".*(k_(?:conv|wino)_[a-z0-9_]+(?:<[^>]*>)?).*", r"\1", name), wgs, lds)
        for k, v in vals[ev].items(): agg[key][k].append(v)
        agg[key]["dur_us"].append((en - st) / 1e3)
    for key, d in sorted(agg.items(), key=lambda kv: -sum(kv[1]["dur_us"])):
        m = {k: sum(v) / len(v) for k, v in d.items()}
        print(key, "dur_us=%.1f" % m["dur_us"])
        if "GRBM_GUI_ACTIVE" in m and m["dur_us"] < MIN_US:
            # the derived clock of such short dispatches reads 2.8-3.5 GHz (the counter window is wider than the kernel):
            # MFMA-busy / TF/s computed from it are artefacts and are not printed
            print("    (shorter than %d us: derived clock / MFMA-busy / TF/s not meaningful, omitted)" % MIN_US)
        elif "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8  # summed over 8 XCDs
            print("    clock %.3f GHz  MFMA-busy %.1f%%  exec TF/s %.1f" % (cyc / m["dur_us"] / 1e3, 100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), m.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) * 512 / m["dur_us"] / 1e6))
            wc = m["SQ_WAVE_CYCLES"]
            print("    of wave-cycles: WAIT_ANY %.1f%%  WAIT_INST_ANY %.1f%%  ACTIVE_INST_ANY %.1f%%" % (100 * m["SQ_WAIT_ANY"] / wc, 100 * m["SQ_WAIT_INST_ANY"] / wc, 100 * m["SQ_ACTIVE_INST_ANY"] / wc))
        else:
            print("   ", {k: "%.4g" % v for k, v in m.items()})
