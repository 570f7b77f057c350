#!/bin/bash
# GPU box: the streaming kernels (k_shade, k_direct) come out in one of two modes from process to process. Four consecutive
# processes of the bench command under the same counter pass (L2 <-> memory side requests and stalls, L1 translation
# misses); per process: GPU cycles and counters of k_shade / k_direct, so that what differs between the modes can be read off.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/shade_modes
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4; do
  timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_GUI_ACTIVE \
    -d "$OUT/p$i" -o p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/p$i.log"; exit 1; }
done
python3 - <<PY
import glob, sqlite3
for i in (1, 2, 3, 4):
    db = glob.glob("$OUT/p%d/*_results.db" % i)[0]
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    kname = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    for kern in ("k_shade", "k_direct"):
        rows = con.execute(f"select {cname}, count(*), sum(value) from counters_collection where {kname} like '%{kern}%' group by {cname}").fetchall()
        print(f"process {i} {kern}: " + ", ".join(f"{c} {s / n:.4g}" for c, n, s in sorted(rows)))
PY
rm -rf "$OUT"/p*/
