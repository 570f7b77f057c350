#!/usr/bin/env python3
"""Per-launch counters of every kernel from the passes of tools/profile_on_box.sh, as JSON for bench.py.

    python tools/pmc_traffic.py gpurun_out/prof_<tag> "<bench arguments of that profile>" > profiles/pmc_traffic.json

Units and corrections:
  * FETCH_SIZE and WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE counts a 128-byte memory-side read request of a wide
    coalesced stream as 64 bytes, and counts the scattered 64-byte record reads of the traversal kernels exactly
    (tools/micro/fetch_size_probe.hip, profiles/r02_micro_probes.txt: ratio 1.000 for a random 64-byte gather,
    0.500 for a 16-byte-per-lane stream). `hbm_bytes_per_launch` therefore doubles the read side only for the
    streaming kernels (STREAMING below); the traversal kernels use the counter as it is.
  * SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs:
    valu_busy = SQ_ACTIVE_INST_VALU * 4 / 1024 / (GRBM_GUI_ACTIVE / 8).
  * TCP_TOTAL_CACHE_ACCESSES_sum is summed over the 256 L1 caches and counts one access per lane and load instruction
    for scattered loads (tools/micro/l1_access_probe.hip under the same counter: 4.00 per 64-byte record read with four
    16-byte loads): l1_access_busy = accesses / 256 / (GRBM_GUI_ACTIVE / 8), in accesses per clock and CU; the probe
    reaches 1.46 with every access a hit. ta_busy = TA_TA_BUSY_sum / 256 / cycles.
  * TCP_TCC_READ_REQ_sum counts the L1 misses: 1.05 per scattered 64-byte record in the probe.
`source_sha1` is the hash of the kernel sources the profiled library was built from: bench.py reports these figures only
while the sources in the tree still have that hash."""
import hashlib
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMING = {"k_raygen", "k_accumulate", "k_direct", "k_resolve", "k_shade"}


def source_sha1():
    h = hashlib.sha1()
    d = os.path.join(ROOT, "raytracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def short(name):
    m = re.search(r"(k_\w+)(<(\w+)>)?", name)
    if not m or m.group(3) == "true":  # the counting variants run only in the counter pass of bench.py
        return None
    return m.group(1)


def counters(db):
    if not os.path.exists(db):
        return {}
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    kname = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    out = {}
    for k, c, n, s in con.execute(f"select {kname}, {cname}, count(*), sum(value) from counters_collection group by {kname}, {cname}"):
        k = short(k)
        if k:
            out.setdefault(k, {})[c] = (n, s / n)
    return out


def durations(db):
    if not os.path.exists(db):
        return {}
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    out = {}
    for n, c, avg in con.execute(f"select {name}, count(*), avg(end-start) from kernels group by {name}"):
        k = short(n)
        if k:
            out[k] = (c, avg / 1e3)
    return out


def main():
    root, args = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    fetch = counters(os.path.join(root, "pmc_fetch", "fetch_results.db"))
    write = counters(os.path.join(root, "pmc_write", "write_results.db"))
    sq = counters(os.path.join(root, "pmc_sq", "sq_results.db"))
    cache = counters(os.path.join(root, "pmc_cache", "cache_results.db"))
    dur = durations(os.path.join(root, "trace", "trace_results.db"))
    res = {"bench_args": args, "source": os.path.basename(root.rstrip("/")), "source_sha1": source_sha1(), "kernels": {}}
    for k in sorted(fetch):
        n, f_kb = fetch[k]["FETCH_SIZE"]
        w_kb = write.get(k, {}).get("WRITE_SIZE", (n, 0.0))[1]
        read_factor = 2 if k in STREAMING else 1
        e = {"launches": n, "fetch_kib_per_launch": round(f_kb, 1), "write_kib_per_launch": round(w_kb, 1),
             "fetch_size_factor": read_factor,
             "hbm_bytes_per_launch": int((read_factor * f_kb + w_kb) * 1024)}
        if k in dur:
            e["trace_avg_us"] = round(dur[k][1], 1)
        s = {c: v[1] for c, v in sq.get(k, {}).items()}
        c_ = {c: v[1] for c, v in cache.get(k, {}).items()}
        cycles = s.get("GRBM_GUI_ACTIVE", 0) / 8
        if cycles > 0:
            e["gpu_cycles_per_launch"] = int(cycles)
            if "SQ_ACTIVE_INST_VALU" in s:
                e["valu_busy"] = round(s["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cycles, 4)
            if "SQ_INSTS_VALU" in s:
                e["valu_instr_per_launch"] = int(s["SQ_INSTS_VALU"])
            cyc_c = c_.get("GRBM_GUI_ACTIVE", 0) / 8 or cycles
            if "TCP_TOTAL_CACHE_ACCESSES_sum" in c_:
                e["l1_accesses_per_launch"] = int(c_["TCP_TOTAL_CACHE_ACCESSES_sum"])
                e["l1_access_busy"] = round(c_["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256 / cyc_c, 4)
            if "TA_TA_BUSY_sum" in c_:
                e["ta_busy"] = round(c_["TA_TA_BUSY_sum"] / 256 / cyc_c, 4)
            if "TCP_TCC_READ_REQ_sum" in c_:
                e["l2_read_requests_per_launch"] = int(c_["TCP_TCC_READ_REQ_sum"])  # L1 misses: one per 64-byte record fetched from L2
            if "TCC_MISS_sum" in c_:
                e["l2_misses_per_launch"] = int(c_["TCC_MISS_sum"])
            if c_.get("TCC_HIT_sum") is not None and c_.get("TCC_MISS_sum") is not None and c_["TCC_HIT_sum"] + c_["TCC_MISS_sum"] > 0:
                e["l2_hit"] = round(c_["TCC_HIT_sum"] / (c_["TCC_HIT_sum"] + c_["TCC_MISS_sum"]), 4)
        res["kernels"][k] = e
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
