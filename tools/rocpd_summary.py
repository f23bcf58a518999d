#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (sqlite) outputs into the small text files committed under profiles/.

  rocpd_summary.py stats  <results.db> <out.csv>          per-kernel calls / total / avg / min / max ns (the --stats table)
  rocpd_summary.py pmc    <results.db> [...] <out.json>   per-kernel average of each collected counter, in bytes, with the
                                                          MI355X_MICROARCH.md corrections (WRITE_SIZE / FETCH_SIZE are KB;
                                                          FETCH_SIZE under-reports 2x on gfx950)
"""
import json
import sqlite3
import sys


def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                     "max(scratch_size), max(lds_size), max(grid_x*grid_y*grid_z), max(workgroup_x) from kernels group by name "
                     "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs,ScratchBytes,LDSBytes,GridSize,WorkgroupSize\n")
        for r in rows:
            f.write('"%s",%d,%d,%.1f,%d,%d,%.2f,%d,%d,%d,%d,%d\n' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, *r[6:]))


def pmc(dbs, out):
    res = {}
    for db in dbs:
        c = sqlite3.connect(db)
        for name, ctr, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                           "group by kernel_name, counter_name"):
            if ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                scale = 1024.0 * (2.0 if ctr == "FETCH_SIZE" else 1.0)
                res.setdefault(name, {})[ctr] = {"dispatches": n, "avg_raw_KB": avg, "avg_bytes_corrected": avg * scale}
            else:
                res.setdefault(name, {})[ctr] = {"dispatches": n, "avg_per_dispatch": avg}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
