"""Build libqsv.so for gfx950 in-tree:  python -m qcmrf_amd.build [--force]"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libqsv.so")
SOURCES = ["qsv_kmulti_m0_r5.hip", "qsv_kmulti_m0_r4.hip", "qsv_kmulti_m0_r3.hip", "qsv_kmulti_m0_low.hip", "qsv.hip",
           "qsv_kmulti_m1.hip", "qsv_kmulti_m2.hip"]      # slowest first
DEPENDS = SOURCES + ["qsv_common.h", "qsv_kernels.h", "qsv_kmulti.h", "qsv_kmulti_inst.h", "qsv_gates.inc", "qsv_multi.inc", "qsv_layout.inc", "qsv_measure.inc",
                     "qsv_exec.inc", os.path.join("..", "..", "include", "qsv.h")]
CFLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
          "-Wno-unused-value", "-Wno-unused-result"]
OBJDIR = os.path.join(CSRC, "_obj")


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPENDS)


def build(force=False, verbose=True):
    """one object per translation unit, compiled side by side (the k_multi instantiations are most of
    the work: qsv_kmulti_inst.h), then one link"""
    if not force and not stale():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "hipcc")
    os.makedirs(OBJDIR, exist_ok=True)

    def compile_one(src):
        import time
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + CFLAGS + ["-c", "-o", obj, src]
        if verbose:
            print("[qcmrf_amd.build]", " ".join(cmd), flush=True)
        t0 = time.time()
        subprocess.check_call(cmd, cwd=CSRC)
        if verbose:
            print("[qcmrf_amd.build] %s: %.0f s" % (src, time.time() - t0), flush=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"]
    if verbose:
        print("[qcmrf_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
