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
