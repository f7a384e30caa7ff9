"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/qsv.h declares (no compute calls: there is no GPU here).  Also: no CPU fallback exists."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib_path():
    from qcmrf_amd import build
    return build.build(verbose=False)


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "qsv.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qsv_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported_and_bound(lib_path):
    from qcmrf_amd import _lib
    names = declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(lib_path)
    for n in names:
        assert hasattr(lib, n), "libqsv.so does not export %s" % n
        assert n in _lib.SIGNATURES, "python binding lacks %s" % n
    assert sorted(_lib.SIGNATURES) == names
    _lib.load()
    assert _lib.load().qsv_version().startswith(b"qsv")


def test_op_record_layout_matches_header():
    from qcmrf_amd import _lib
    assert ctypes.sizeof(_lib.QsvOp) == 4 * 4 + 2 * 16 * 4 + 8 + 8 + 8 == _lib.OP_DTYPE.itemsize
    assert _lib.K_COUNT == 13 and ctypes.sizeof(_lib.Stats) == 13 * 24 + 8 + 8 + 8


def test_code_object_is_gfx950_only(lib_path):
    blob = open(lib_path, "rb").read()
    archs = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert archs == {b"gfx950"}, archs


def test_no_gpu_means_loud_failure_not_a_fallback(lib_path):
    from qcmrf_amd import _lib, QCMRF, Aer
    if _lib.load().qsv_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        _lib.Engine(4)
    with pytest.raises(RuntimeError):
        Aer.get_backend("qasm_simulator").run(QCMRF([[0, 1]], [-0.1] * 4), shots=10)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "qcmrf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "qsv_ref" not in txt, f
                # host side = Python + numpy + ctypes (north star): no PyTorch anywhere in the package,
                # not even for the multi-process rendezvous (qcmrf_amd.comm is standard library only)
                assert not re.search(r"^\s*(from|import)\s+torch\b", txt, flags=re.M), f
