#!/bin/bash
# GPU box: rocprofv3 kernel stats and HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, program directly after --) of the
# gate micro-benchmark (bench.py --gates: one dedicated kernel per gate on a 28-qubit state).   usage: scripts/profile_gates.sh TAG
TAG=${1:-rXX}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp; cd /tmp || exit 1
CMD="bench.py --gates --steps 10"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_gates_$TAG -o run -- python3 $ROOT/$CMD > $OUT/prof_gates_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_gates_$TAG -o run -- python3 $ROOT/$CMD > $OUT/pmc_fetch_gates_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_gates_$TAG -o run -- python3 $ROOT/$CMD > $OUT/pmc_write_gates_$TAG.log 2>&1 || exit 1
cd $ROOT
python3 - "$TAG" <<'PY'
import sqlite3, sys, re
tag = sys.argv[1]
out = "gpurun_out"
def q(db, sql):
    return sqlite3.connect(db).cursor().execute(sql).fetchall()
stats = {}
try:
    rows = q("%s/prof_gates_%s/run_results.db" % (out, tag), "select name, count(*), avg(end - start) from kernels group by name")
except Exception:
    rows = []
    db = sqlite3.connect("%s/prof_gates_%s/run_results.db" % (out, tag))
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if "kernel_dispatch" in t]
    ks = [t for t in tabs if "kernel_symbol" in t]
    if kd and ks:
        rows = db.execute("select s.kernel_name, count(*), avg(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name" % (kd[0], ks[0])).fetchall()
for name, n, avg in rows:
    stats[re.sub(r"^void ", "", name.split("(")[0])] = [n, avg / 1e6, None, None]
for key, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    try:
        for kn, v in q("%s/pmc_%s_gates_%s/run_results.db" % (out, sub, tag), "select kernel_name, avg(value) from counters_collection where counter_name = '%s' group by kernel_name" % key):
            s = stats.setdefault(re.sub(r"^void ", "", kn.split("(")[0]), [0, 0.0, None, None])
            s[2 if sub == "fetch" else 3] = v
    except Exception as e:
        print("pmc", sub, e)
with open("%s/%s_gates_kernel_stats.csv" % (out, tag), "w") as f:
    f.write("kernel,calls,avg_ms,hbm_read_GB(2 x FETCH_SIZE KiB),hbm_write_GB\n")
    for k in sorted(stats):
        n, ms, fe, wr = stats[k]
        f.write("\"%s\",%d,%.4f,%s,%s\n" % (k, n, ms, "" if fe is None else "%.4f" % (2 * fe * 1024 / 1e9), "" if wr is None else "%.4f" % (wr * 1024 / 1e9)))
print(open("%s/%s_gates_kernel_stats.csv" % (out, tag)).read())
PY
