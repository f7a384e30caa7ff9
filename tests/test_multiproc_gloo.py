"""CPU: the N>1 path across REAL processes (world_size 2, gloo), one shard per rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_two_ranks_over_gloo(tmp_path):
    out = tmp_path / "result.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_gloo_worker.py"), str(out), "gloo"]
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.load(open(out))
    assert set(res) == {"reference/0", "reference/2", "auto/0", "auto/2"}
    for key, v in res.items():
        assert v["err"] < 1e-13, (key, v)
        assert v["shots"] == 20000 and v["outside_support"] == 0
        assert 0.7 < v["chi2"] < 1.4, (key, v)
        assert v["n_exchanges"] == v["engine_exchanges"]
    assert res["auto/2"]["n_exchanges"] == 0            # exchange-free layout for the fused circuit
    assert res["reference/2"]["n_exchanges"] >= 1       # ancilla targets sit on the shard bit
    assert res["reference/0"]["n_exchanges"] >= 1


def _check(res):
    assert set(res) == {"reference/0", "reference/2", "auto/0", "auto/2"}
    for key, v in res.items():
        assert v["err"] < 1e-13, (key, v)
        assert v["shots"] == 20000 and v["outside_support"] == 0
        assert 0.7 < v["chi2"] < 1.4, (key, v)
        assert v["n_exchanges"] == v["engine_exchanges"]
    assert res["auto/2"]["n_exchanges"] == 0 and res["reference/2"]["n_exchanges"] >= 1


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.timeout(600)
def test_ranks_over_the_package_socket_comm(tmp_path, world):
    """the same worker over qcmrf_amd.comm.SocketComm (standard library only: the shipped process
    group), started as plain processes with the launcher's environment variables -- no torch at all"""
    out = tmp_path / "result.json"
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r),
                   WORLD_SIZE=str(world), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(world))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), str(out), "socket"],
                                      cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=540) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-4000:]
    _check(json.load(open(out)))


def test_socket_comm_collectives_in_threads():
    """SocketComm semantics on 5 ranks (threads of this process, TCP endpoint): object and raw
    all-gathers in rank order, broadcast from the hub and from another rank, barrier"""
    import threading
    import numpy as np
    from qcmrf_amd.comm import SocketComm
    world, port = 5, free_port()
    res, errs = {}, []

    def body(r):
        try:
            env = {"RANK": str(r), "WORLD_SIZE": str(world), "QSV_COMM_ENDPOINT": "tcp:127.0.0.1:%d" % port}
            c = SocketComm(timeout_s=60, env=env)
            out = {"ag": c.allgather({"r": r, "x": [r] * r}), "b0": c.bcast("hub" if r == 0 else None),
                   "b3": c.bcast(("from", 3) if r == 3 else None, src=3),
                   "u64": c.allgather_u64(np.arange(4, dtype=np.uint64) + np.uint64(10 * r)),
                   "f64": c.allgather_f64(0.5 * r), "bytes": c.allgather_bytes(bytes([r]) * r)}
            c.barrier()
            for _ in range(50):                                   # many small collectives back to back stay in step
                assert c.allgather_f64(r).tolist() == [float(i) for i in range(world)]
            c.close()
            res[r] = out
        except Exception as e:                                    # noqa: BLE001
            errs.append((r, repr(e)))
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not errs, errs
    for r in range(world):
        o = res[r]
        assert o["ag"] == [{"r": i, "x": [i] * i} for i in range(world)]
        assert o["b0"] == "hub" and o["b3"] == ("from", 3)
        assert o["u64"].shape == (world, 4) and o["u64"][3].tolist() == [30, 31, 32, 33]
        assert o["f64"].tolist() == [0.0, 0.5, 1.0, 1.5, 2.0]
        assert o["bytes"] == [bytes([i]) * i for i in range(world)]


def test_socket_comm_gather_and_data_only_frames():
    """gather_bytes: only the hub receives (the tail of the sampling merge); frames carry data, never pickles: an object
    the codec does not know is refused on the sending side, an unknown tag on the receiving side"""
    import threading
    import numpy as np
    from qcmrf_amd import comm as qc
    world, port = 3, free_port()
    res, errs = {}, []

    def body(r):
        try:
            env = {"RANK": str(r), "WORLD_SIZE": str(world), "QSV_COMM_ENDPOINT": "tcp:127.0.0.1:%d" % port}
            c = qc.SocketComm(timeout_s=60, env=env)
            res[r] = (c.gather_bytes(bytes([65 + r]) * (r + 1)), c.allgather(np.arange(3) * (1 + 1j) * r))
            c.barrier()
            c.close()
        except Exception as e:                                    # noqa: BLE001
            errs.append((r, repr(e)))
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(60) for t in th]
    assert not errs, errs
    assert res[0][0] == [b"A", b"BB", b"CCC"] and res[1][0] is None and res[2][0] is None
    assert all(np.array_equal(res[r][1][2], np.arange(3) * (2 + 2j)) for r in range(world))
    with pytest.raises(TypeError):
        qc.encode({"f": open})                                    # a callable is not data
    with pytest.raises(TypeError):
        qc.encode(np.array([object()]))
    import pickle
    with pytest.raises(ValueError):
        qc.decode(pickle.dumps({"a": 1}))                         # a pickle is not a frame of this codec
    assert "pickle" not in open(qc.__file__).read().replace("never pickle", "").replace("unpickled", "").replace("pickles", "")


def test_socket_comm_refuses_a_peer_without_the_launch_key():
    """the hello carries a MAC under a key derived from the launcher's environment (or QSV_COMM_TOKEN): a process that
    merely reaches the rendezvous address is turned away"""
    import threading
    from qcmrf_amd.comm import SocketComm
    port = free_port()
    out = {}

    def hub():
        try:
            SocketComm(timeout_s=10, env={"RANK": "0", "WORLD_SIZE": "2", "QSV_COMM_ENDPOINT": "tcp:127.0.0.1:%d" % port,
                                          "QSV_COMM_TOKEN": "right"})
            out["hub"] = "accepted"
        except ConnectionError as e:
            out["hub"] = str(e)

    def intruder():
        try:
            SocketComm(timeout_s=10, env={"RANK": "1", "WORLD_SIZE": "2", "QSV_COMM_ENDPOINT": "tcp:127.0.0.1:%d" % port,
                                          "QSV_COMM_TOKEN": "wrong"})
            out["peer"] = "accepted"
        except Exception as e:                                    # noqa: BLE001
            out["peer"] = repr(e)
    th = [threading.Thread(target=hub), threading.Thread(target=intruder)]
    [t.start() for t in th]
    [t.join(30) for t in th]
    assert "not a rank of this launch" in out["hub"] and out["peer"] != "accepted", out


def test_child_endpoint_follows_the_parent_rule():
    from qcmrf_amd.comm import child_endpoint
    one = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29511", "WORLD_SIZE": "8", "LOCAL_WORLD_SIZE": "8"}
    assert child_endpoint(one, "legs").startswith("unix:") and child_endpoint(one, "legs").endswith("-legs")
    two = dict(one, WORLD_SIZE="16", MASTER_ADDR="node0")
    assert child_endpoint(two, "legs") == "tcp:node0:29513"
    assert child_endpoint(dict(one, QSV_COMM_ENDPOINT="unix:whatever"), "legs") == child_endpoint(one, "legs")


def test_spmd_ingest_gives_every_rank_the_same_program():
    """one process per GPU: each rank reads 1/world of the composite blocks of a QCMRF circuit and ONE all-gather completes
    the picture -- identical ops (tables bit for bit) on every rank; a circuit whose blocks are not phase blocks, and one
    that raises, fall back to everyone reading everything without leaving a rank behind in the collective"""
    import threading
    import numpy as np
    from qcmrf_amd import QCMRF, workloads as wl
    from qcmrf_amd.circuit import QuantumCircuit
    from qcmrf_amd.comm import SocketComm
    from qcmrf_amd.ingest import ingest
    C = wl.grid(2, 4)
    qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
    odd = QuantumCircuit(4, 4)                                   # a composite block that is NOT AND . cp . AND triples
    sub = QuantumCircuit(3, name="blk")
    sub.h(0); sub.cx(0, 1); sub.rz(0.3, 2)
    odd.h(3); odd.append(sub, [0, 1, 2]); odd.append(sub.inverse(), [1, 2, 3]); odd.measure(0, 0)
    bad = QuantumCircuit(3, 3)
    bad.append(sub, [0, 1, 2]); bad.measure(0, 0); bad.append(sub, [0, 1, 2])       # gate after measurement: ValueError
    refs = [ingest(c, peephole=True) for c in (qc, odd)]
    world, port = 3, free_port()
    res, errs = {}, []

    def body(r):
        try:
            c = SocketComm(timeout_s=60, env={"RANK": str(r), "WORLD_SIZE": str(world), "QSV_COMM_ENDPOINT": "tcp:127.0.0.1:%d" % port})
            out = [ingest(x, peephole=True, comm=c) for x in (qc, odd)]
            try:
                ingest(bad, peephole=True, comm=c)
                out.append("no error")
            except ValueError as e:
                out.append(str(e))
            c.barrier()
            c.close()
            res[r] = out
        except Exception as e:                                    # noqa: BLE001
            errs.append((r, repr(e)))
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not errs and len(res) == world, errs
    for r in range(world):
        for got, ref in zip(res[r][:2], refs):
            assert len(got.ops) == len(ref.ops) and got.n_source_ops == ref.n_source_ops and got.measure == ref.measure
            for a, b in zip(got.ops, ref.ops):
                assert a.kind == b.kind and a.support() == b.support()
                if a.kind == "diag":
                    assert np.array_equal(a.table, b.table)
        assert "after it was measured" in res[r][2]
