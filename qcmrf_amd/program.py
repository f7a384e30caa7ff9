"""Encode planned (physical) IR ops as the flat ``qsv_op`` records ``qsv_exec`` consumes
(include/qsv.h): one ctypes crossing per circuit instead of one per gate."""
from __future__ import annotations

import numpy as np

from . import _lib

_X = np.array([[0, 1], [1, 0]], dtype=np.complex128)


def encode(ops):
    """list of ir.Op (physical qubits) -> (records: ndarray[OP_DTYPE], data: float64 pool)"""
    rec = np.zeros(len(ops), dtype=_lib.OP_DTYPE)
    pool = []
    top = 0

    def put(arr):
        nonlocal top
        flat = np.ascontiguousarray(arr, dtype=np.complex128).view(np.float64).ravel()
        pool.append(flat)
        off = top
        top += flat.size
        return off

    def fill(r, qs, vals=None):
        n = len(qs)
        if n > _lib.MAX_CTRL:
            raise ValueError("gate touches %d qubits; the engine's limit per gate is %d" % (n, _lib.MAX_CTRL))
        r["n"] = n
        r["qubits"][:n] = qs
        if vals is not None:
            r["vals"][:n] = vals

    for r, op in zip(rec, ops):
        k = op.kind
        if op.new_pass:
            r["flags"] = _lib.OPF_NEW_PASS
        if k == "init":
            r["kind"] = _lib.OP_INIT_UNIFORM if op.mask else _lib.OP_INIT_ZERO
            r["mask"] = op.mask
        elif k == "u":
            r["kind"] = _lib.OP_1Q
            r["target"] = op.target
            fill(r, op.ctrls, op.vals)
            r["data_off"] = put(op.mat)
        elif k == "x":
            r["kind"] = _lib.OP_MCX
            r["target"] = op.target
            fill(r, op.ctrls, op.vals)
        elif k == "diag":
            r["kind"] = _lib.OP_DIAG
            fill(r, op.qubits)
            r["data_off"] = put(op.table)
        elif k == "mcphase":
            r["kind"] = _lib.OP_MCPHASE
            fill(r, op.qubits, op.vals)
            r["angle"] = op.angle
        elif k == "mux":
            if len(op.ctrls) > 10:
                raise ValueError("multiplexer with %d controls exceeds the engine limit of 10" % len(op.ctrls))
            r["kind"] = _lib.OP_MUX
            r["target"] = op.target
            fill(r, op.ctrls)
            r["data_off"] = put(op.mats)
        elif k == "kq":
            if len(op.qubits) > _lib.MAX_KQ:
                raise ValueError("dense gate on %d qubits exceeds the engine limit of %d" % (len(op.qubits), _lib.MAX_KQ))
            r["kind"] = _lib.OP_KQ
            fill(r, op.qubits)
            r["data_off"] = put(op.mat)
        elif k == "swap":
            r["kind"] = _lib.OP_SWAP
            fill(r, op.a, op.b)
        else:
            raise ValueError("cannot encode op kind %r" % k)
    data = np.concatenate(pool) if pool else np.zeros(0, dtype=np.float64)
    return rec, data


def run_stepwise(engine, ops):
    """Same program through the one-call-per-gate entry points (parity tests exercise both)."""
    for op in ops:
        k = op.kind
        if k == "init":
            engine.init_uniform(op.mask) if op.mask else engine.init_zero()
        elif k == "u":
            engine.apply_1q(op.target, op.mat, op.ctrls, op.vals)
        elif k == "x":
            engine.apply_mcx(op.ctrls, op.target, op.vals)
        elif k == "diag":
            engine.apply_diag(op.qubits, op.table)
        elif k == "mcphase":
            engine.apply_mcphase(op.qubits, op.angle, op.vals)
        elif k == "mux":
            engine.apply_mux(op.ctrls, op.target, op.mats)
        elif k == "kq":
            engine.apply_kq(op.qubits, op.mat)
        elif k == "swap":
            engine.swap_layout(op.a, op.b)
        else:
            raise ValueError("cannot run op kind %r" % k)
