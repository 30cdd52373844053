#!/usr/bin/env python3
"""Merge rocprofv3 --pmc result databases (one per counter pass) into one per-kernel, per-launch CSV.

    python tools/pmc_collect.py out.csv pass1.db pass2.db ...

Every counter is averaged over the launches of a kernel in its pass.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in
KB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide streaming reads, hence the x2 column
(MI355X_MICROARCH.md, HBM)."""
import collections
import csv
import sqlite3
import sys

out, dbs = sys.argv[1], sys.argv[2:]
rows = collections.defaultdict(dict)
for path in dbs:
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    sfx = [t for t in tabs if t.startswith("rocpd_pmc_event_")][0][len("rocpd_pmc_event_"):]
    q = f"""select k.display_name, d.dispatch_id, (d.end-d.start), d.grid_size_x, p.name, sum(e.value)
    from rocpd_kernel_dispatch_{sfx} d join rocpd_info_kernel_symbol_{sfx} k on d.kernel_id=k.id
    join rocpd_pmc_event_{sfx} e on e.event_id=d.event_id join rocpd_info_pmc_{sfx} p on e.pmc_id=p.id
    group by d.dispatch_id, p.name"""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, dur, grid, seen = collections.Counter(), collections.Counter(), {}, set()
    for name, did, du, gx, pn, v in cur.execute(q):
        key = name.split("(")[0]
        agg[key][pn] += v
        if did not in seen:
            seen.add(did); cnt[key] += 1; dur[key] += du; grid[key] = gx
    for k, a in agg.items():
        r = rows[k]
        r["kernel"] = k; r["grid_size"] = grid[k]; r["launches"] = cnt[k]
        r.setdefault("avg_us_under_pmc", round(dur[k] / cnt[k] / 1e3, 2))
        for n, v in a.items():
            r[n + "_avg"] = round(v / cnt[k], 3)
cols = ["kernel", "grid_size", "launches", "avg_us_under_pmc"]
extra = sorted({c for r in rows.values() for c in r} - set(cols))
for r in rows.values():
    if "FETCH_SIZE_avg" in r:
        r["FETCH_SIZE_KB_raw_avg"] = r.pop("FETCH_SIZE_avg")
        r["fetch_MB_corrected_x2"] = round(2 * r["FETCH_SIZE_KB_raw_avg"] / 1024, 3)
    if "WRITE_SIZE_avg" in r:
        r["WRITE_SIZE_KB_avg"] = r.pop("WRITE_SIZE_avg")
    if "TCC_HIT_sum_avg" in r and "TCC_MISS_sum_avg" in r and r["TCC_HIT_sum_avg"] + r["TCC_MISS_sum_avg"] > 0:
        r["L2_hit_pct"] = round(100 * r["TCC_HIT_sum_avg"] / (r["TCC_HIT_sum_avg"] + r["TCC_MISS_sum_avg"]), 1)
extra = sorted({c for r in rows.values() for c in r} - set(cols))
with open(out, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=cols + extra)
    w.writeheader()
    for k in sorted(rows, key=lambda k: -rows[k].get("avg_us_under_pmc", 0) * rows[k]["launches"]):
        w.writerow(rows[k])
print("wrote", out, len(rows), "kernels")
