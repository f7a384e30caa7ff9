#!/bin/bash
# GPU box: why does the 34-qubit read+write pass sit at 0.755 of peak where 28..32 qubits reach 0.80, and why is a 32 GiB shard
# written at 6.66 TB/s where 128..256 GiB ones reach 7.1-7.5?  Address-translation (UTCL1), L2 <-> fabric stall and wave
# wait counters per kernel at 31, 32 and 34 qubits, one --pmc group per run (program directly after --).
# Output: gpurun_out/pmc_tlb/summary.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_tlb; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for W in ${WIDTHS:-31 32 34}; do
  python3 $ROOT/scripts/tlb_case.py $W > $OUT/timing_$W.log 2>&1 || exit 1
  i=0
  for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
             "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
             "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum" \
             "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
             "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d $OUT/W${W}_g$i -o run -- python3 $ROOT/scripts/tlb_case.py $W > $OUT/W${W}_g$i.log 2>&1 || echo "W=$W group $i failed: $(tail -2 $OUT/W${W}_g$i.log)"
  done
done
cd $ROOT
python3 - <<'PY'
import sqlite3, glob, os, re
out = "gpurun_out/pmc_tlb"
res = {}
for db in sorted(glob.glob(out + "/W*_g*/run_results.db")):
    W = int(re.search(r"/W(\d+)_g", db).group(1))
    cur = sqlite3.connect(db).cursor()
    try:
        rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%k_multi%' or kernel_name like '%k_init_prod%' group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(db, e); continue
    for kn, cn, v, c in rows:
        short = re.sub(r"^void ", "", kn.split("(")[0])
        res.setdefault((short, W), {})[cn] = v
with open(out + "/summary.txt", "w") as f:
    for (kn, W) in sorted(res):
        A = 2.0 ** W
        d = res[(kn, W)]
        line = "%-34s W=%d  " % (kn, W) + "  ".join("%s=%.4g (%.3g /KiB)" % (c.replace("_sum", ""), v, v / (16 * A / 1024)) for c, v in sorted(d.items()))
        print(line); f.write(line + "\n")
PY
cat $OUT/timing_*.log
