"""Build libqsv.so for gfx950 in-tree:  python -m qcmrf_amd.build [--force]"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libqsv.so")
SOURCES = ["qsv.hip"]
DEPENDS = ["qsv.hip", "qsv_kernels.h", "qsv_gates.inc", "qsv_multi.inc", "qsv_layout.inc", "qsv_measure.inc",
           "qsv_exec.inc", os.path.join("..", "..", "include", "qsv.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
         "-Wno-unused-value", "-Wno-unused-result"]


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPENDS)


def build(force=False, verbose=True):
    if not force and not stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc] + FLAGS + ["-o", OUT] + SOURCES + ["-ldl"]
    if verbose:
        print("[qcmrf_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
