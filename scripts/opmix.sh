#!/bin/bash
# GPU box: instruction mix per op of the general k_multi pass (difference between 60-op and 4-op passes)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/opmix; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for kind in ${KINDS:-ccx_lane ccx_block ccx_reg cp_regreg u_dense}; do for n in 4 60; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $OUT/a_${kind}_$n -o run -- python3 $ROOT/scripts/opmix.py $kind $n > $OUT/a_${kind}_$n.log 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAVES -d $OUT/b_${kind}_$n -o run -- python3 $ROOT/scripts/opmix.py $kind $n > $OUT/b_${kind}_$n.log 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU -d $OUT/c_${kind}_$n -o run -- python3 $ROOT/scripts/opmix.py $kind $n > $OUT/c_${kind}_$n.log 2>&1 || echo "counter group c unavailable"
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM_NORM -d $OUT/d_${kind}_$n -o run -- python3 $ROOT/scripts/opmix.py $kind $n > $OUT/d_${kind}_$n.log 2>&1 || echo "counter group d unavailable"
done; done
cd $ROOT
python3 - <<'PY'
import sqlite3, glob, os
out = "gpurun_out/opmix"
res = {}
for d in sorted(glob.glob(out + "/[abcd]_*")):
    if not os.path.isdir(d): continue
    db = os.path.join(d, "run_results.db")
    if not os.path.exists(db): continue
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%k_multi%' group by kernel_name, counter_name").fetchall()
    key = os.path.basename(d)[2:]
    for kn, cn, v, c in rows:
        res.setdefault(key, {})[cn] = v
for kind in ("ccx_lane", "ccx_block", "ccx_reg", "cp_regreg", "u_dense"):
    a, b = res.get(kind + "_4", {}), res.get(kind + "_60", {})
    print(kind, {k: round((b[k] - a[k]) / 56.0, 1) for k in b if k in a}, "| 60-op pass:", {k: round(v) for k, v in b.items()})
PY
