#!/bin/bash
# GPU box: where do the waves of the dense k-qubit kernels wait?  SQ counters per kernel (separate --pmc passes, program
# directly after --), for the three micro-benchmark cases and every kq_variant.  Output: gpurun_out/kq_pmc/summary.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kq_pmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
python3 $ROOT/scripts/kq_variants.py > $OUT/timing.log 2>&1 || exit 1
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/$tag -o run -- python3 $ROOT/scripts/kq_variants.py pmc > $OUT/$tag.log 2>&1 || echo "group $tag failed"
done
cd $ROOT
python3 - <<'PY'
import sqlite3, glob, os
out = "gpurun_out/kq_pmc"
res = {}
for db in sorted(glob.glob(out + "/*/run_results.db")):
    cur = sqlite3.connect(db).cursor()
    try:
        rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%k_kq%' group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(db, e); continue
    for kn, cn, v, c in rows:
        res.setdefault(kn.split("(")[0], {})[cn] = (v, c)
with open(out + "/summary.txt", "w") as f:
    for kn in sorted(res):
        line = kn + "  " + "  ".join("%s=%.4g(n=%d)" % (c, v[0], v[1]) for c, v in sorted(res[kn].items()))
        print(line); f.write(line + "\n")
PY
