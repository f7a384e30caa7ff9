"""GPU box: the reference's unfused gate stream (fusion 0) through the general k_multi passes, with this round's switches:
   general_combos (controls on workgroup-uniform bits resolved per workgroup), fold_init_h, planner rule for pure controls.
   python scripts/unfused_variants.py W [trace]"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = int(sys.argv[1]) if len(sys.argv) > 1 else 28
if len(sys.argv) > 2 and sys.argv[2] == "child":
    from qcmrf_amd import QCMRF, workloads as wl
    from qcmrf_amd.backend import QsvBackend
    C = wl.for_width(W)
    qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
    be = QsvBackend()
    eo = json.loads(sys.argv[3])
    be.run(qc, shots=16, fusion=0, engine_options=eo)
    trace = dict(eo, trace_passes=1) if os.environ.get("QSV_TRACE") else eo
    r = be.run(qc, shots=16, fusion=0, profile=True, engine_options=trace).result()
    k = r.metadata(0)["stats"]["kinds"]
    print("RESULT evolve %.1f ms | " % (r.metadata(0)["time_evolve"] * 1e3) +
          "  ".join("%s: %d x %.2f ms (%.3f)" % (n, v["launches"], v["ms"] / v["launches"], v["bytes"] / v["ms"] / 1e6 / 8000) for n, v in k.items() if v["launches"]), flush=True)
    be.close()
    sys.exit(0)
for ctrl_top in ("1", "0"):
    for combos in (1, 0):
        for fold in (1, 0):
            if fold == 0 and (combos == 0 or ctrl_top == "0"):
                continue
            env = dict(os.environ, QSV_PLANNER_CTRL_TOP=ctrl_top)
            if len(sys.argv) > 2 and sys.argv[2] == "trace":
                env["QSV_TRACE"] = "1"
            p = subprocess.run([sys.executable, os.path.abspath(__file__), str(W), "child", json.dumps({"general_combos": combos, "fold_init_h": fold})],
                               env=env, capture_output=True, text=True, timeout=500)
            res = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
            print("W=%d pure controls on %s bits, combos %d, fold_init_h %d: %s" % (W, "block" if ctrl_top == "1" else "lane", combos, fold, res[0][7:] if res else "FAILED " + p.stderr[-400:]), flush=True)
            if env.get("QSV_TRACE"):
                print("\n".join(l for l in p.stderr.splitlines() if "[qsv pass]" in l), flush=True)
