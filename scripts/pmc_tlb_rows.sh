#!/bin/bash
# GPU box: UTCL1 translation counters per launch for scripts/tlb_rows.py (tile size and placement of a read+write pass at 34 qubits)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_tlb_rows; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
W=${W:-34}
python3 $ROOT/scripts/tlb_rows.py $W 3 > $OUT/timing.log 2>&1 || { cat $OUT/timing.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum -d $OUT/g1 -o run -- python3 $ROOT/scripts/tlb_rows.py $W 1 > $OUT/g1.log 2>&1 || echo "pmc run failed: $(tail -3 $OUT/g1.log)"
cd $ROOT
python3 - <<'PY'
import sqlite3, glob
out = "gpurun_out/pmc_tlb_rows"
cases = [l.split("CASE ")[1][:20].strip() for l in open(out + "/timing.log") if l.startswith("CASE ")]
for db in glob.glob(out + "/g1/run_results.db"):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection where kernel_name like '%k_multi%' order by dispatch_id").fetchall()
    per = {}
    for did, kn, cn, v in rows:
        per.setdefault(did, {"k": kn.split("(")[0]})[cn] = per.get(did, {}).get(cn, 0) + v
    ids = sorted(per)
    with open(out + "/summary.txt", "w") as f:
        # every case launches twice (one warm-up, one timed): report the second
        for i, did in enumerate(ids):
            d = per[did]
            label = cases[i // 2] if i // 2 < len(cases) else "?"
            req = d.get("TCP_UTCL1_REQUEST_sum", 0) or 1
            line = "%-20s launch %d  %-28s miss %.4g (%.3f %% of requests)  miss-under-miss %.4g (%.2f %%)  requests %.4g" % (
                label, i % 2, d["k"].replace("void ", ""), d.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0), 100 * d.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / req,
                d.get("TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum", 0), 100 * d.get("TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum", 0) / req, req)
            print(line); f.write(line + "\n")
PY
cat $OUT/timing.log
