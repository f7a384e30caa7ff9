#!/bin/bash
# GPU box: what holds the dense 5-qubit gate at 0.64 of the stream when its loads and stores are whole 1 KiB runs?  Wave,
# LDS, L1 <-> L2 latency and L2 <-> fabric counters of the kernels of scripts/kq_lds_case.py, one --pmc group per run
# (program directly after --).  Output: gpurun_out/kq_lds_pmc/summary.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kq_lds_pmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
python3 $ROOT/scripts/kq_lds_case.py > $OUT/timing.log 2>&1 || { cat $OUT/timing.log; exit 1; }
cat $OUT/timing.log
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum" \
           "TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" \
           "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_NC_READ_REQ_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o run -- python3 $ROOT/scripts/kq_lds_case.py > $OUT/g$i.log 2>&1 || echo "group $i failed: $(tail -2 $OUT/g$i.log)"
done
cd $ROOT
python3 - <<'PY'
import sqlite3, glob, re
out = "gpurun_out/kq_lds_pmc"
res = {}
for db in sorted(glob.glob(out + "/g*/run_results.db")):
    cur = sqlite3.connect(db).cursor()
    try:
        rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%k_kq%' or kernel_name like '%k_pair%' group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(db, e); continue
    for kn, cn, v, c in rows:
        res.setdefault(re.sub(r"^void ", "", kn.split("(")[0]), {})[cn.replace("_sum", "")] = v
with open(out + "/summary.txt", "w") as f:
    for kn in sorted(res):
        line = "%-32s " % kn + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(res[kn].items()))
        print(line); f.write(line + "\n")
PY
