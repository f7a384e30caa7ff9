"""Host-side process group used when one process drives one GPU (torchrun-style launch).

Only tiny host values travel here (per-shard probability mass, sampled outcomes, the RCCL
unique id, IPC handles); amplitudes move between GPUs inside libqsv over RCCL/xGMI, never through
Python.  Standard library only -- sockets, struct, hmac -- as the north star asks of the host
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

import hashlib
import hmac
import os
import socket
import struct
import time

import numpy as np

_HDR = struct.Struct("<Q")
_MAGIC = b"qsvcomm2"
_MAC_LEN = 16


# ---- wire format of the few bootstrap objects: a tagged encoding that can only ever produce data ----------------
# (None, bool, int, float, str, bytes, list / tuple, dict with str keys, numeric numpy arrays).  Nothing received from
# a socket is ever unpickled: whoever reaches the rendezvous address first must not thereby run code in rank 0.
_NUMERIC_KINDS = "biufc"


def _enc(obj, out):
    if obj is None:
        out.append(b"N")
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b"T" if obj else b"F")
    elif isinstance(obj, (int, np.integer)):
        b = str(int(obj)).encode("ascii")
        out.append(b"I" + _HDR.pack(len(b)) + b)
    elif isinstance(obj, (float, np.floating)):
        out.append(b"D" + struct.pack("<d", float(obj)))
    elif isinstance(obj, str):
        b = obj.encode("utf-8")
        out.append(b"S" + _HDR.pack(len(b)) + b)
    elif isinstance(obj, (bytes, bytearray, memoryview)):
        b = bytes(obj)
        out.append(b"B" + _HDR.pack(len(b)) + b)
    elif isinstance(obj, (list, tuple)):
        out.append((b"L" if isinstance(obj, list) else b"U") + _HDR.pack(len(obj)))
        for x in obj:
            _enc(x, out)
    elif isinstance(obj, dict):
        out.append(b"M" + _HDR.pack(len(obj)))
        for k, v in obj.items():
            if not isinstance(k, str):
                raise TypeError("qcmrf_amd.comm: dict keys must be str, not %r" % type(k).__name__)
            _enc(k, out)
            _enc(v, out)
    elif isinstance(obj, np.ndarray):
        if obj.dtype.kind not in _NUMERIC_KINDS:
            raise TypeError("qcmrf_amd.comm: only numeric arrays travel, not dtype %s" % obj.dtype)
        a = np.ascontiguousarray(obj)
        dt = a.dtype.str.encode("ascii")
        out.append(b"A" + _HDR.pack(len(dt)) + dt + _HDR.pack(a.ndim) + b"".join(_HDR.pack(d) for d in a.shape))
        b = a.tobytes()
        out.append(_HDR.pack(len(b)) + b)
    else:
        raise TypeError("qcmrf_amd.comm cannot send a %s" % type(obj).__name__)


def encode(obj):
    out = []
    _enc(obj, out)
    return b"".join(out)


def _dec(buf, off):
    tag = bytes(buf[off:off + 1])
    off += 1
    if tag == b"N":
        return None, off
    if tag in (b"T", b"F"):
        return tag == b"T", off
    if tag == b"D":
        return struct.unpack_from("<d", buf, off)[0], off + 8
    if tag in (b"I", b"S", b"B"):
        (n,) = _HDR.unpack_from(buf, off)
        off += _HDR.size
        raw = bytes(buf[off:off + n])
        if len(raw) != n:
            raise ValueError("qcmrf_amd.comm: truncated frame")
        off += n
        return (int(raw) if tag == b"I" else raw.decode("utf-8") if tag == b"S" else raw), off
    if tag in (b"L", b"U"):
        (n,) = _HDR.unpack_from(buf, off)
        off += _HDR.size
        items = []
        for _ in range(n):
            x, off = _dec(buf, off)
            items.append(x)
        return (items if tag == b"L" else tuple(items)), off
    if tag == b"M":
        (n,) = _HDR.unpack_from(buf, off)
        off += _HDR.size
        d = {}
        for _ in range(n):
            k, off = _dec(buf, off)
            v, off = _dec(buf, off)
            d[k] = v
        return d, off
    if tag == b"A":
        (n,) = _HDR.unpack_from(buf, off)
        off += _HDR.size
        dt = np.dtype(bytes(buf[off:off + n]).decode("ascii"))
        off += n
        if dt.kind not in _NUMERIC_KINDS:
            raise ValueError("qcmrf_amd.comm: array dtype %s refused" % dt)
        (nd,) = _HDR.unpack_from(buf, off)
        off += _HDR.size
        shape = []
        for _ in range(nd):
            (d,) = _HDR.unpack_from(buf, off)
            off += _HDR.size
            shape.append(d)
        (nb,) = _HDR.unpack_from(buf, off)
        off += _HDR.size
        a = np.frombuffer(bytes(buf[off:off + nb]), dtype=dt).reshape(shape).copy()
        return a, off + nb
    raise ValueError("qcmrf_amd.comm: unknown tag %r in a frame" % tag)


def decode(buf):
    obj, off = _dec(buf, 0)
    if off != len(buf):
        raise ValueError("qcmrf_amd.comm: %d stray bytes after a frame" % (len(buf) - off))
    return obj


def _group_key(env):
    """what a peer has to know to be let in: QSV_COMM_TOKEN if the launcher set one, else the rendezvous facts every
    rank of THIS launch shares (run id, restart count, master address and port, world size)"""
    tok = env.get("QSV_COMM_TOKEN")
    if tok:
        return hashlib.sha256(b"qsv-token:" + tok.encode("utf-8")).digest()
    facts = "|".join(env.get(k, "") for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE"))
    return hashlib.sha256(b"qsv-launch:" + facts.encode("utf-8")).digest()


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

    def gather_bytes(self, payload):
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


def child_endpoint(env, tag):
    """``QSV_COMM_ENDPOINT`` for a second process group of the same ranks (bench.py's exchange-leg children): the rule
    of ``_endpoint`` with ``tag`` mixed in -- an abstract unix socket on one node, TCP two ports further otherwise"""
    kind, where = _endpoint({k: v for k, v in env.items() if k != "QSV_COMM_ENDPOINT"})
    if kind == "unix":
        return "unix:%s-%s" % (where, tag)
    return "tcp:%s:%d" % (where[0], where[1] + 1)


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
    small plain-data objects (``encode``: numbers, strings, bytes, lists, dicts, numeric arrays -- never pickle) plus raw
    fixed-width variants for the per-step sampling merge.  A peer is admitted only with a hello that carries a MAC under
    the launch's key (``_group_key``)."""

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
        key = _group_key(env)
        hello = _MAGIC + struct.pack("<II", self.rank, self.world)
        hello += hmac.new(key, hello, hashlib.sha256).digest()[:_MAC_LEN]
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
                try:
                    msg = bytes(_recv_exact(c, len(hello)))
                except (ConnectionError, socket.timeout):
                    c.close()
                    raise ConnectionError("qcmrf_amd.comm: a peer connected and did not say hello")
                body, mac = msg[:-_MAC_LEN], msg[-_MAC_LEN:]
                r, w = struct.unpack("<II", body[len(_MAGIC):])
                if (body[:len(_MAGIC)] != _MAGIC or not hmac.compare_digest(mac, hmac.new(key, body, hashlib.sha256).digest()[:_MAC_LEN])
                        or w != self.world or not 0 < r < self.world or self._peers[r] is not None):
                    c.close()
                    raise ConnectionError("qcmrf_amd.comm: a peer that is not a rank of this launch tried to join (bad hello)")
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

    def gather_bytes(self, payload):
        """every rank contributes bytes, ONLY rank 0 receives them (list by rank; None elsewhere): one send per rank
        and no answer to wait for -- the tail of the per-step sampling merge"""
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            return [payload] + [bytes(_recv_frame(c)) for c in self._peers[1:]]
        _send_frame(self._hub, payload)
        return None

    def allgather(self, obj):
        return [decode(p) for p in self.allgather_bytes(encode(obj))]

    def bcast(self, obj, src=0):
        if self.world == 1:
            return obj
        if src != 0:                                          # via the hub: rare (bootstrap only)
            return self.allgather(obj if self.rank == src else None)[src]
        if self.rank == 0:
            frame = encode(obj)
            frame = _HDR.pack(len(frame)) + frame
            for c in self._peers[1:]:
                c.sendall(frame)
            return obj
        return decode(_recv_frame(self._hub))

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
