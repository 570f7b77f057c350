#!/usr/bin/env python3
"""Summarise rocprofv3's rocpd sqlite output (kernel trace and PMC passes) as text tables for profiles/.
    python tools/rocpd_summary.py gpurun_out/prof_r1 > profiles/r01_summary.txt"""
import glob
import os
import re
import sqlite3
import sys


def short(n, width):
    m = re.search(r"(k_\w+(?:<[\w, ]+>)?)", n)
    if m:
        return m.group(1)
    return n.replace("(anonymous namespace)::", "").split("(")[0][-width:]


def q(con, sql):
    return con.execute(sql).fetchall()


def kernel_stats(db):
    con = sqlite3.connect(db)
    cols = [r[1] for r in q(con, "pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = q(con, f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name} order by 3 desc")
    total = sum(r[2] for r in rows) or 1
    out = [f"{'kernel':<60} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}"]
    for n, c, tot, avg, mn, mx in rows:
        out.append(f"{short(n, 58):<60} {c:>7} {tot / 1e6:>10.3f} {avg / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {100 * tot / total:>6.2f}")
    return "\n".join(out)


def pmc_stats(db):
    con = sqlite3.connect(db)
    cols = [r[1] for r in q(con, "pragma table_info(counters_collection)")]
    kname = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    val = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    rows = q(con, f"select {kname}, {cname}, count(*), sum({val}), avg({val}) from counters_collection group by {kname}, {cname} order by 1, 2")
    out = [f"{'kernel':<50} {'counter':<30} {'dispatches':>10} {'sum':>20} {'avg_per_dispatch':>20}"]
    for k, c, n, s, a in rows:
        out.append(f"{short(k, 48):<50} {c:<30} {n:>10} {s:>20.1f} {a:>20.1f}")
    return "\n".join(out)


if __name__ == "__main__":
    root = sys.argv[1]
    for db in sorted(glob.glob(os.path.join(root, "*", "*_results.db"))):
        tag = os.path.basename(os.path.dirname(db))
        print(f"==== {tag} ({os.path.relpath(db, root)})")
        try:
            print(kernel_stats(db) if tag == "trace" else pmc_stats(db))
        except Exception as e:  # keep going: one malformed pass should not hide the others
            print(f"(could not summarise: {e})")
        print()
