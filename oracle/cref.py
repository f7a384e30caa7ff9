"""ctypes binding of oracle/qsv_ref.c.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libqsv_ref.so")
_lib = None


def build(force=False):
    src = os.path.join(HERE, "qsv_ref.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(src) > os.path.getmtime(LIB):
        subprocess.check_call(["make", "-C", HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.ref_norm.restype = C.c_double
        _lib.ref_num_threads.restype = C.c_int
    return _lib


def _ip(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.complex128)).view(np.float64).ravel()
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def host_threads():
    """threads this process may really use: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // p))
            break
        except Exception:
            continue
    return n


def set_threads(n):
    load().ref_set_threads(int(n))


class RefState:
    """2^nq complex128 on the host, gates by OpenMP sweeps."""

    def __init__(self, nq):
        self.nq = nq
        self.lib = load()
        self.state = np.empty(2 ** nq, dtype=np.complex128)
        self._p = self.state.ctypes.data_as(C.c_void_p)
        self.lib.ref_init_zero(self._p, nq)

    def threads(self):
        return int(self.lib.ref_num_threads())

    def apply_1q(self, t, m, ctrls=(), vals=None):
        ca, cp = _ip(ctrls)
        va, vp = _ip([1] * len(ca) if vals is None else vals)
        ma, mp = _dp(m)
        self.lib.ref_apply_1q(self._p, self.nq, int(t), mp, len(ca), cp, vp, 0)

    def apply_mcx(self, ctrls, t, vals=None):
        ca, cp = _ip(ctrls)
        va, vp = _ip([1] * len(ca) if vals is None else vals)
        self.lib.ref_apply_1q(self._p, self.nq, int(t), None, len(ca), cp, vp, 1)

    def apply_mcphase(self, qubits, angle, vals=None):
        qa, qp = _ip(qubits)
        va, vp = _ip([1] * len(qa) if vals is None else vals)
        self.lib.ref_apply_mcphase(self._p, self.nq, len(qa), qp, vp, C.c_double(angle))

    def apply_diag(self, qubits, table):
        qa, qp = _ip(qubits)
        ta, tp = _dp(table)
        self.lib.ref_apply_diag(self._p, self.nq, len(qa), qp, tp)

    def apply_mux(self, ctrls, t, mats):
        ca, cp = _ip(ctrls)
        ma, mp = _dp(mats)
        self.lib.ref_apply_mux(self._p, self.nq, len(ca), cp, int(t), mp)

    def norm(self):
        return float(self.lib.ref_norm(self._p, self.nq))

    def marginal(self, qubits, fmask=0, fval=0):
        qa, qp = _ip(qubits)
        out = np.zeros(2 ** len(qa))
        self.lib.ref_marginal(self._p, self.nq, len(qa), qp, C.c_uint64(fmask), C.c_uint64(fval),
                              out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def sample(self, shots, seed):
        u = np.sort(np.random.RandomState(seed).random_sample(shots)) * self.norm()
        out = np.zeros(shots, dtype=np.uint64)
        self.lib.ref_sample(self._p, self.nq, C.c_uint64(shots), u.ctypes.data_as(C.POINTER(C.c_double)),
                            out.ctypes.data_as(C.POINTER(C.c_uint64)))
        return out

    def run_stream(self, ops):
        """oracle.gate_stream primitives"""
        from .sv_numpy import MATS
        for op in ops:
            k = op[0]
            if k == "x":
                self.apply_mcx([], op[1])
            elif k in MATS:
                self.apply_1q(op[1], MATS[k])
            elif k == "mcx":
                self.apply_mcx(list(op[1]), op[2])
            elif k == "cp":
                self.apply_mcphase([op[2], op[3]], op[1])
            elif k in ("measure", "barrier"):
                pass
            else:
                raise ValueError("oracle: unknown primitive %r" % (k,))
        return self.state
