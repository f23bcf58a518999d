#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (sqlite) outputs into the small text files committed under profiles/.

  rocpd_summary.py stats  <results.db> <out.csv>          per-kernel calls / total / avg / min / max ns (the --stats table)
  rocpd_summary.py pmc    <results.db> [...] <out.json>   per-kernel average of each collected counter, in bytes, with the
                                                          MI355X_MICROARCH.md corrections (WRITE_SIZE / FETCH_SIZE are KB;
                                                          FETCH_SIZE under-reports 2x on gfx950)
  rocpd_summary.py overlap <results.db> <out.txt>         steady-state concurrency of the hot path's kernels: share of the wall time
                                                          with k expansion kernels / k advice-writing kernels / nothing running
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


def overlap(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    hot = [(n, s, e) for n, s, e in rows if "expand_fast" in n or "k_merkle" in n or "k_prologue" in n or "k_strands" in n]
    ex = [r for r in hot if "expand_fast" in r[0]]
    if len(ex) < 12:
        open(out, "w").write("too few launches\n"); return
    t0, t1 = ex[len(ex) // 4][1], ex[-len(ex) // 4][2]          # the middle half of the run: steady state
    cls = {"expand": lambda n: "expand_fast" in n, "chains": lambda n: "k_merkle_bn" in n, "writers": lambda n: "expand_fast" in n or "k_merkle_bn" in n,
           "prologue": lambda n: "k_prologue" in n, "glue": lambda n: "k_strands" in n or "k_merkle_gl" in n, "any": lambda n: True}
    lines = ["window %.1f ms, %d expansion kernels inside" % ((t1 - t0) / 1e6, sum(1 for r in ex if r[1] >= t0 and r[2] <= t1))]
    for key, f in cls.items():
        ev = []
        for n, s, e in hot:
            if f(n) and e > t0 and s < t1:
                ev.append((max(s, t0), 1)); ev.append((min(e, t1), -1))
        ev.sort()
        hist, cur, last = {}, 0, t0
        for t, d in ev:
            hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
        hist[cur] = hist.get(cur, 0) + (t1 - last)
        tot = float(t1 - t0)
        lines.append("%-9s " % key + "  ".join("%d running: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items()) if v > 0))
    for key in ("expand_fast", "k_merkle_bn", "k_prologue", "k_strands", "k_merkle_gl"):
        d = [(e - s) / 1e6 for n, s, e in hot if key in n and s >= t0 and e <= t1]
        if d: lines.append("%-12s n=%d avg %.2f ms  min %.2f  max %.2f" % (key, len(d), sum(d) / len(d), min(d), max(d)))
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "overlap":
        overlap(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
