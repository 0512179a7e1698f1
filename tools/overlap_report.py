#!/usr/bin/env python
"""How much do kernels of different streams overlap in a rocprofv3 --kernel-trace database?
    python tools/overlap_report.py results.db
Prints wall time covered by >= 1 and by >= 2 running kernels, and the overlap per pair of kernel families."""
import re, sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select coalesce(s.display_name, s.kernel_name), d.start, d.end, d.queue_id from rocpd_kernel_dispatch d "
                 "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
def fam(n):
    m = re.search(r"(k_[a-z0-9_]+)", n)
    return m.group(1) if m else n[:24]
ev = []
for n, s, e, q in rows:
    ev.append((s, 1, fam(n))); ev.append((e, -1, fam(n)))
ev.sort()
active = defaultdict(int); cov1 = cov2 = 0; last = None; pair = defaultdict(int)
for t, d, f in ev:
    if last is not None and t > last:
        n = sum(active.values())
        if n >= 1: cov1 += t - last
        if n >= 2:
            cov2 += t - last
            fs = sorted(k for k, v in active.items() if v > 0)
            pair["+".join(fs)] += t - last
    active[f] += d; last = t
print(f"{len(rows)} dispatches on queues {sorted(set(r[3] for r in rows))}: busy {cov1/1e6:.2f} ms, >=2 kernels running {cov2/1e6:.2f} ms ({100*cov2/max(cov1,1):.1f} %)")
for k, v in sorted(pair.items(), key=lambda kv: -kv[1])[:12]:
    print(f"   {v/1e6:8.3f} ms  {k}")
