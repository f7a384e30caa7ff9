"""Turn the rocprofv3 result databases of scripts/profile_round.sh into the committed summaries:

    profiles/<tag>_kernel_stats.csv            per-kernel calls / total / average duration
    profiles/<tag>_pmc.csv                     per-kernel average FETCH_SIZE / WRITE_SIZE (KiB)
    profiles/pmc_traffic.json                  HBM bytes per launch of the dominant kernels

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE reports exactly half of a 16-B-per-lane streaming
read on gfx950 (x2 before comparing with a byte count); WRITE_SIZE is exact for 16-B stores.

    python scripts/pmc_summary.py TAG [gpurun_out]
"""
import csv
import json
import os
import re
import sqlite3
import sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    return [(short(n), c, s / 1e3, a / 1e3, lo / 1e3, hi / 1e3, 100.0 * s / tot) for n, c, s, a, lo, hi in rows]


def counter_avg(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, count(*), avg(value), max(grid_size) from counters_collection where counter_name = ? "
                       "group by kernel_name", (counter,)).fetchall()
    return {short(n): (c, v, g) for n, c, v, g in rows}


def run(label, sub):
    out = {}
    st = os.path.join(src, "prof_%s%s" % (sub, tag), "run_results.db")
    if os.path.exists(st):
        with open(os.path.join(PROF, "%s_%skernel_stats.csv" % (tag, sub)), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
            for r in kernel_stats(st):
                w.writerow([r[0], r[1]] + ["%.3f" % x for x in r[2:]])
                out.setdefault(r[0], {})["avg_ms_rocprof"] = r[3] / 1e3
    fe = os.path.join(src, "pmc_fetch_%s%s" % (sub, tag), "run_results.db")
    wr = os.path.join(src, "pmc_write_%s%s" % (sub, tag), "run_results.db")
    if os.path.exists(fe) and os.path.exists(wr):
        F, Wc = counter_avg(fe, "FETCH_SIZE"), counter_avg(wr, "WRITE_SIZE")
        with open(os.path.join(PROF, "%s_%spmc.csv" % (tag, sub)), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "dispatches", "avg_FETCH_SIZE_KiB", "avg_WRITE_SIZE_KiB", "hbm_bytes_per_launch(2*FETCH+WRITE)"])
            for k in sorted(set(F) | set(Wc)):
                fv = F.get(k, (0, 0.0, 0))[1]
                wv = Wc.get(k, (0, 0.0, 0))[1]
                hb = 2.0 * fv * 1024 + wv * 1024
                w.writerow([k, F.get(k, Wc.get(k))[0], "%.3f" % fv, "%.3f" % wv, "%.0f" % hb])
                out.setdefault(k, {}).update({"FETCH_SIZE_KiB": fv, "WRITE_SIZE_KiB": wv, "hbm_bytes": hb})
    return out


main = run("default path", "")
sweeps = run("full-width sweeps", "sweeps_")
sweeps34 = run("full-width sweeps, 34 qubits", "sweeps34_")
modes = run("general passes (MODE 1 / MODE 0), 28 qubits", "modes_")
unfused = run("the reference's unfused stream (fusion 0), 28 qubits", "unfused_")
tfile = os.path.join(PROF, "pmc_traffic.json")
T = json.load(open(tfile)) if os.path.exists(tfile) else {}
T["_method_" + tag] = ("scripts/profile_round.sh %s: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                       "`python3 bench.py --steps 5 --warmup 2 --no-cpu --no-variants` (34-qubit default path) and the same with "
                       "`--config 2 --no-fold` (28-qubit full-width sweeps); KiB; read bytes = 2 x FETCH_SIZE x 1024 on gfx950 "
                       "(MI355X_MICROARCH.md), WRITE_SIZE x 1024 exact; per-kernel averages in profiles/%s_*pmc.csv" % (tag, tag))


def put(key, wkey, table, prefix):
    for k, v in table.items():
        if k.startswith(prefix) and "hbm_bytes" in v:
            T.setdefault(key, {})[wkey] = v["hbm_bytes"]
            T[key]["kernel_" + wkey] = k


put("k_init_prod", "hbm_bytes_per_launch_W34", main, "k_init_prod")
put("k_multi", "hbm_bytes_per_launch_W28", sweeps, "k_multi<5, false")
put("k_multi_init", "hbm_bytes_per_launch_W28", sweeps, "k_multi<5, true")
put("k_multi", "hbm_bytes_per_launch_W34", sweeps34, "k_multi<5, false")
put("k_multi_init", "hbm_bytes_per_launch_W34", sweeps34, "k_multi<5, true")
put("k_multi_mode1", "hbm_bytes_per_launch_W28", modes, "k_multi<5, false, 1")
put("k_multi_mode0", "hbm_bytes_per_launch_W28", modes, "k_multi<4, false, 0")
put("k_multi_unfused_stream", "hbm_bytes_per_launch_W28", unfused, "k_multi<4, false, 0")
json.dump(T, open(tfile, "w"), indent=1)
for name, table in (("default", main), ("sweeps", sweeps), ("sweeps34", sweeps34), ("modes", modes), ("unfused", unfused)):
    for k, v in table.items():
        if "hbm_bytes" in v and v["hbm_bytes"] > 1e8:
            print("%-8s %-28s avg %.3f ms  HBM %.4f GB / launch" % (name, k, v.get("avg_ms_rocprof", float("nan")), v["hbm_bytes"] / 1e9))
