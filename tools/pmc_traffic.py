#!/usr/bin/env python3
"""HBM traffic per launch of the traversal kernels, from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_on_box.sh.

    python tools/pmc_traffic.py gpurun_out/prof_<tag> "<bench arguments of that profile>" > profiles/pmc_traffic.json

FETCH_SIZE and WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE counts a 128-byte memory-side read request as 64 bytes, so a
read stream is under-counted by up to 2x: `hbm_bytes_per_launch` doubles the read side (upper bound), `..._uncorrected`
does not (lower bound). bench.py reports the corrected figure as roofline.traffic when it runs the profiled workload."""
import json
import os
import re
import sqlite3
import sys


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    kname = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    out = {}
    for k, n, s in con.execute(f"select {kname}, count(*), sum(value) from counters_collection where {cname} = ? group by {kname}", (counter,)):
        m = re.search(r"(k_\w+)(<(\w+)>)?", k)
        if not m or (m.group(3) == "true"):  # the counting variants run only in the counter pass of bench.py
            continue
        out[m.group(1)] = (n, s)
    return out


def main():
    root, args = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    fetch = per_kernel(os.path.join(root, "pmc_fetch", "fetch_results.db"), "FETCH_SIZE")
    write = per_kernel(os.path.join(root, "pmc_write", "write_results.db"), "WRITE_SIZE")
    res = {"bench_args": args, "source": os.path.basename(root.rstrip("/")), "kernels": {}}
    for k in sorted(fetch):
        n, f = fetch[k]
        _, w = write.get(k, (n, 0.0))
        f_kb, w_kb = f / n, w / n
        res["kernels"][k] = {"launches": n, "fetch_kib_per_launch": round(f_kb, 1), "write_kib_per_launch": round(w_kb, 1),
                             "hbm_bytes_per_launch": int((2 * f_kb + w_kb) * 1024),
                             "hbm_bytes_per_launch_uncorrected": int((f_kb + w_kb) * 1024)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
