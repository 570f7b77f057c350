#!/usr/bin/env python3
"""Summarise rocprofv3's rocpd sqlite output (kernel trace and PMC passes) as text tables for profiles/.
    python tools/rocpd_summary.py gpurun_out/prof_r1 > profiles/r01_summary.txt"""
import glob
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.pmc_traffic import STREAMING  # noqa: E402  kernels whose FETCH_SIZE needs the x2 of a coalesced stream


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


def derived(root):
    """Per kernel: duration, HBM GB/s (FETCH_SIZE doubled for the streaming kernels only: gfx950 counts a 128-byte read
    request of a coalesced stream as 64 bytes and the scattered 64-byte reads of the traversal kernels exactly; the same
    per-kernel factor as tools/pmc_traffic.py, so that both files give one number per kernel), VALU
    utilisation (SQ_ACTIVE_INST_VALU is in 4-cycle units, summed over 1024 SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs),
    how a wavefront's resident time splits, cache hit rates."""
    def load(tag, fn):
        path = os.path.join(root, tag, fn)
        if not os.path.exists(path):
            return {}
        con = sqlite3.connect(path)
        out = {}
        if tag == "trace":
            cols = [r[1] for r in q(con, "pragma table_info(kernels)")]
            name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
            for n, c, avg in q(con, f"select {name}, count(*), avg(end-start) from kernels group by {name}"):
                out[short(n, 58)] = {"calls": c, "avg_ns": avg}
        else:
            cols = [r[1] for r in q(con, "pragma table_info(counters_collection)")]
            kname = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
            cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
            for k, c, n, sm in q(con, f"select {kname}, {cname}, count(*), sum(value) from counters_collection group by {kname}, {cname}"):
                out.setdefault(short(k, 48), {})[c] = sm / n
        return out
    tr, fe, wr = load("trace", "trace_results.db"), load("pmc_fetch", "fetch_results.db"), load("pmc_write", "write_results.db")
    sq, ca = load("pmc_sq", "sq_results.db"), load("pmc_cache", "cache_results.db")
    rows = [f"{'kernel':<22} {'avg_us':>9} {'HBM_GB/s':>9} {'of_8TB/s':>8} {'VALU_busy':>9} {'wave:wait_mem':>13} {'wait_issue':>10} {'executing':>9} {'L2_hit':>7}"]
    for k in sorted(tr, key=lambda k: -tr[k]["avg_ns"] * tr[k]["calls"]):
        us = tr[k]["avg_ns"] / 1e3
        f, w = fe.get(k, {}).get("FETCH_SIZE"), wr.get(k, {}).get("WRITE_SIZE")
        kshort = re.sub(r"<.*", "", k)
        factor = 2 if kshort in STREAMING else 1
        gbs = (factor * f + w) * 1024 / (us * 1e-6) / 1e9 if f is not None and w is not None and us > 0 else None
        s_ = sq.get(k, {})
        valu = (s_["SQ_ACTIVE_INST_VALU"] * 4 / 1024) / (s_["GRBM_GUI_ACTIVE"] / 8) if "SQ_ACTIVE_INST_VALU" in s_ and s_.get("GRBM_GUI_ACTIVE") else None
        wc = s_.get("SQ_WAVE_CYCLES")
        c_ = ca.get(k, {})
        l2 = c_["TCC_HIT_sum"] / (c_["TCC_HIT_sum"] + c_["TCC_MISS_sum"]) if c_.get("TCC_HIT_sum") is not None and (c_.get("TCC_HIT_sum", 0) + c_.get("TCC_MISS_sum", 0)) > 0 else None
        fmt = lambda v, p="{:.2f}": "-" if v is None else p.format(v)
        rows.append(f"{k:<22} {us:>9.1f} {fmt(gbs, '{:.0f}'):>9} {fmt(None if gbs is None else gbs / 8000):>8} {fmt(valu):>9} "
                    f"{fmt(s_['SQ_WAIT_ANY'] / wc if wc else None):>13} {fmt(s_['SQ_WAIT_INST_ANY'] / wc if wc else None):>10} "
                    f"{fmt(s_['SQ_ACTIVE_INST_ANY'] / wc if wc else None):>9} {fmt(l2):>7}")
    return "\n".join(rows)


if __name__ == "__main__":
    root = sys.argv[1]
    if os.path.exists(os.path.join(root, "trace", "trace_results.db")):
        print("==== derived (per launch averages)")
        print(derived(root))
        print()
    for db in sorted(glob.glob(os.path.join(root, "*", "*_results.db"))):
        tag = os.path.basename(os.path.dirname(db))
        print(f"==== {tag} ({os.path.relpath(db, root)})")
        try:
            print(kernel_stats(db) if tag == "trace" else pmc_stats(db))
        except Exception as e:  # keep going: one malformed pass should not hide the others
            print(f"(could not summarise: {e})")
        print()
