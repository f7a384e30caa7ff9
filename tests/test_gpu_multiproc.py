"""GPU: the one-process-per-rank path with REAL libqsv engines -- 2 and 4 ranks sharing the test
box's single GPU (gloo rendezvous on the host, HIP IPC between the ranks' shards)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.timeout(600)
def test_ranks_as_processes_on_one_gpu(tmp_path, world):
    out = tmp_path / "result.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_gpu_rank_worker.py"), str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.load(open(out))
    assert set(res) == {"reference/2", "reference/0", "auto/3", "auto/2"}
    for key, v in res.items():
        assert v["err"] < 1e-12, (key, v)
        assert v["shots"] == 20000 and v["outside_support"] == 0
        assert 0.7 < v["chi2"] < 1.4, (key, v)
        assert all(x == v["n_exchanges"] for x in v["engine_exchanges"]), (key, v)
    assert res["auto/3"]["n_exchanges"] == 0 and res["auto/2"]["n_exchanges"] == 0     # exchange-free layouts
    assert res["reference/2"]["n_exchanges"] >= 1 and res["reference/0"]["n_exchanges"] >= 1
    assert res["reference/2"]["transport"] == "p2p"          # two ranks on one device: RCCL is not an option
    # fused circuit, ancillas on the shard bits: ALL shard bits are swapped in by one batched exchange
    assert res["reference/2"]["n_exchanges"] == 1


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.timeout(900)
def test_config4_w31_ranks_exchange_real_half_shards(tmp_path, world):
    """BASELINE config 4 at full size: 31 qubits = 32 GiB over 2 then 4 rank processes (16 / 8 GiB
    shards).  layout='reference' puts the ancillas on the shard bits (QCMRF.py:231-236), so the full-width
    sweeps need half-shard exchanges between the ranks; every rank checks slices of its own shard."""
    out = tmp_path / "result.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_gpu_rank_worker.py"), str(out), "config4"]
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.load(open(out))
    assert set(res) == {"reference/sweeps", "auto/sweeps", "auto/fold"}
    for key, v in res.items():
        assert v["err"] < 1e-12, (key, v)
        assert abs(v["norm"] - 1.0) < 1e-12, (key, v)
        assert v["shots"] == 4096 and v["outside_support"] == 0
        assert abs(v["success"] - v["delta"]) < 5 * (v["delta"] * (1 - v["delta"]) / 4096) ** 0.5 + 1e-3, (key, v)
        assert all(x == v["n_exchanges"] for x in v["engine_exchanges"]), (key, v)
    assert res["reference/sweeps"]["n_exchanges"] >= 1 and res["reference/sweeps"]["transport"] == "p2p"
    assert res["auto/sweeps"]["n_exchanges"] == 0 and res["auto/fold"]["n_exchanges"] == 0
