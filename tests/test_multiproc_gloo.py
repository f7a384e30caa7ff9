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
           os.path.join(ROOT, "tests", "_gloo_worker.py"), str(out)]
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
