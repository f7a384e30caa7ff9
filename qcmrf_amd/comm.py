"""Host-side process group used when one process drives one GPU (torchrun-style launch).

Only tiny host values travel here (per-shard probability mass, sampled outcomes, the RCCL
unique id, IPC handles); amplitudes move between GPUs inside libqsv over RCCL/xGMI, never through
Python.  Standard library only -- sockets, struct, pickle -- as the north star asks of the host
side ("Python, numpy + ctypes, no PyTorch"): the launcher may well be ``torch.distributed.run``
(bench.py's contract), but all this module takes from it is the environment it exports
(RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR, MASTER_PORT).

Topology: a star.  Rank 0 listens, every other rank keeps one stream socket to it; a collective
is "everyone sends its part to the hub, the hub answers everyone with the assembled result".
Messages are a few bytes to a few KiB, so what matters is the number of system calls on the
critical path (2 per rank), not bandwidth.

Rendezvous address
  * all ranks on this host (LOCAL_WORLD_SIZE == WORLD_SIZE, the only case a single xGMI node
    needs): an ABSTRACT unix socket named after MASTER_ADDR / MASTER_PORT -- nothing to clean up,
    nothing that can collide with the TCP store the launcher itself keeps on MASTER_PORT;
  * otherwise TCP on MASTER_ADDR, port QSV_COMM_PORT (default MASTER_PORT + 1).
  ``QSV_COMM_ENDPOINT=unix:<name>`` / ``tcp:<host>:<port>`` overrides both.
"""
from __future__ import annotations

import os
import pickle
import socket
import struct
import time

import numpy as np

_HDR = struct.Struct("<Q")
_MAGIC = b"qsvcomm1"


class SingleProcess:
    rank = 0
    world = 1

    def allgather(self, obj):
        return [obj]

    def bcast(self, obj, src=0):
        return obj

    def barrier(self):
        pass

    def allgather_u64(self, arr):
        return np.ascontiguousarray(arr, dtype=np.uint64)[None, :]

    def allgather_f64(self, value):
        return np.array([value], dtype=np.float64)

    def allgather_bytes(self, payload):
        return [bytes(payload)]

    def close(self):
        pass


def _recv_exact(sock, n):
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("qcmrf_amd.comm: peer closed the connection in the middle of a collective")
        got += k
    return buf


def _send_frame(sock, payload):
    sock.sendall(_HDR.pack(len(payload)) + payload)


def _recv_frame(sock):
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    return _recv_exact(sock, n) if n else bytearray()


def _endpoint(env):
    ep = env.get("QSV_COMM_ENDPOINT")
    if ep:
        kind, _, rest = ep.partition(":")
        if kind == "unix":
            return "unix", rest
        if kind == "tcp":
            host, _, port = rest.rpartition(":")
            return "tcp", (host, int(port))
        raise ValueError("QSV_COMM_ENDPOINT must be unix:<name> or tcp:<host>:<port>, not %r" % ep)
    addr = env.get("MASTER_ADDR", "127.0.0.1")
    port = int(env.get("MASTER_PORT", "29500"))
    world = int(env.get("WORLD_SIZE", "1"))
    local = int(env.get("LOCAL_WORLD_SIZE", str(world)))
    if local == world and hasattr(socket, "AF_UNIX"):
        return "unix", "qsv-comm-%s-%d-%s-%s" % (addr, port, env.get("TORCHELASTIC_RUN_ID", "none"),
                                                 env.get("TORCHELASTIC_RESTART_COUNT", "0"))
    return "tcp", (addr, int(env.get("QSV_COMM_PORT", str(port + 1))))


class SocketComm:
    """The process group of a one-process-per-GPU launch: ``allgather`` / ``bcast`` / ``barrier`` of
    small picklable objects plus raw fixed-width variants for the per-step sampling merge."""

    def __init__(self, rank=None, world=None, timeout_s=600.0, env=None):
        env = os.environ if env is None else env
        self.rank = int(env.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(env.get("WORLD_SIZE", "1")) if world is None else int(world)
        if not 0 <= self.rank < self.world:
            raise ValueError("rank %d not in [0, %d)" % (self.rank, self.world))
        self._timeout = float(timeout_s)
        self._peers = []              # hub: socket of rank r at index r (None for itself)
        self._hub = None              # other ranks: the socket to rank 0
        self._listener = None
        if self.world == 1:
            return
        kind, where = _endpoint(env)
        family = socket.AF_UNIX if kind == "unix" else socket.AF_INET
        address = ("\0" + where) if kind == "unix" else where
        hello = _MAGIC + struct.pack("<II", self.rank, self.world)
        if self.rank == 0:
            ls = socket.socket(family, socket.SOCK_STREAM)
            if kind == "tcp":
                ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            ls.bind(address)
            ls.listen(self.world)
            ls.settimeout(self._timeout)
            self._listener = ls
            self._peers = [None] * self.world
            for _ in range(self.world - 1):
                c, _ = ls.accept()
                c.settimeout(self._timeout)
                msg = bytes(_recv_exact(c, len(hello)))
                r, w = struct.unpack("<II", msg[len(_MAGIC):])
                if msg[:len(_MAGIC)] != _MAGIC or w != self.world or not 0 < r < self.world or self._peers[r] is not None:
                    raise ConnectionError("qcmrf_amd.comm: unexpected hello from a peer (rank %d of %d)" % (r, w))
                self._tune(c, kind)
                self._peers[r] = c
            for c in self._peers[1:]:
                c.sendall(b"\1")       # everyone is in: release them together
        else:
            deadline = time.monotonic() + self._timeout
            while True:
                c = socket.socket(family, socket.SOCK_STREAM)
                try:
                    c.connect(address)
                    break
                except (ConnectionRefusedError, FileNotFoundError, OSError):
                    c.close()
                    if time.monotonic() > deadline:
                        raise TimeoutError("qcmrf_amd.comm: rank %d could not reach rank 0 at %r" % (self.rank, where))
                    time.sleep(0.01)
            c.settimeout(self._timeout)
            self._tune(c, kind)
            c.sendall(hello)
            _recv_exact(c, 1)
            self._hub = c

    @staticmethod
    def _tune(sock, kind):
        if kind == "tcp":
            sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    # ---- the one primitive: every rank contributes bytes, every rank receives all of them ----
    def allgather_bytes(self, payload):
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [bytes(_recv_frame(c)) for c in self._peers[1:]]
            blob = b"".join([_HDR.pack(len(p)) + p for p in parts])
            frame = _HDR.pack(len(blob)) + blob
            for c in self._peers[1:]:
                c.sendall(frame)
            return parts
        _send_frame(self._hub, payload)
        blob = _recv_frame(self._hub)
        parts, off = [], 0
        for _ in range(self.world):
            (n,) = _HDR.unpack_from(blob, off)
            off += _HDR.size
            parts.append(bytes(blob[off:off + n]))
            off += n
        return parts

    def allgather(self, obj):
        return [pickle.loads(p) for p in self.allgather_bytes(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))]

    def bcast(self, obj, src=0):
        if self.world == 1:
            return obj
        if src != 0:                                          # via the hub: rare (bootstrap only)
            return self.allgather(obj if self.rank == src else None)[src]
        if self.rank == 0:
            frame = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
            frame = _HDR.pack(len(frame)) + frame
            for c in self._peers[1:]:
                c.sendall(frame)
            return obj
        return pickle.loads(_recv_frame(self._hub))

    def barrier(self):
        self.allgather_bytes(b"")

    def allgather_u64(self, arr):
        """equal-length uint64 vectors of every rank -> (world, n) array, one collective, no pickling"""
        a = np.ascontiguousarray(arr, dtype=np.uint64)
        parts = self.allgather_bytes(a.tobytes())
        return np.frombuffer(b"".join(parts), dtype=np.uint64).reshape(self.world, a.size).copy()

    def allgather_f64(self, value):
        """one double per rank -> float64[world]"""
        parts = self.allgather_bytes(struct.pack("<d", float(value)))
        return np.frombuffer(b"".join(parts), dtype=np.float64).copy()

    def close(self):
        for c in self._peers:
            if c is not None:
                c.close()
        self._peers = []
        if self._hub is not None:
            self._hub.close()
            self._hub = None
        if self._listener is not None:
            self._listener.close()
            self._listener = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def from_environment(timeout_s=600.0):
    """the process group the launcher's environment describes (SingleProcess without one)"""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return SocketComm(timeout_s=timeout_s)
    return SingleProcess()
