#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc sqlite result per kernel: python tools/pmc_db.py <results.db> [name-filter]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
sfx = [t for t in tabs if t.startswith("rocpd_pmc_event_")][0][len("rocpd_pmc_event_"):]
q = f"""select k.display_name, d.dispatch_id, (d.end-d.start), p.name, sum(e.value)
from rocpd_kernel_dispatch_{sfx} d join rocpd_info_kernel_symbol_{sfx} k on d.kernel_id=k.id
join rocpd_pmc_event_{sfx} e on e.event_id=d.event_id join rocpd_info_pmc_{sfx} p on e.pmc_id=p.id
group by d.dispatch_id, p.name"""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.Counter(); seen = set()
for name, did, du, pn, v in cur.execute(q):
    key = name.split("(")[0][-44:]
    agg[key][pn] += v
    if did not in seen:
        seen.add(did); cnt[key] += 1; dur[key] += du
for k, a in agg.items():
    if flt and flt not in k: continue
    print(f"{k:44s} n={cnt[k]:4d} avg_us={dur[k]/cnt[k]/1e3:8.1f} " + " ".join(f"{n}={v/cnt[k]:.4g}" for n, v in sorted(a.items())))
