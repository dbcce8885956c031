"""Rendezvous of the ranks of ONE node without any framework: what `bench.py --gpus N` and a multi-rank caller of the
package need before RCCL exists -- hand the 128-byte ncclUniqueId from rank 0 to the others, a barrier, a maximum over
ranks, a gather of small values.  The reference has no launcher at all (hmm.py:276-283 is a plain call); the data path's
collectives are RCCL inside liblnasr_hip.so (csrc/comm.hip), this module never carries data of the hot path.

Ranks find each other through the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT -- the variables
elastic launchers set).  MASTER_PORT itself is NOT listened on: under such a launcher that port already belongs to the
launcher's own key-value store.  Rank 0 listens on an ephemeral loopback port
and publishes it (with a nonce) in a file named after MASTER_ADDR / MASTER_PORT in the temporary directory; the other ranks
poll the file, connect and present the nonce.  A stale file of an earlier run leads to a refused connection or a refused
nonce, and the peer reads the file again.  Star topology: every operation is a gather at rank 0 followed by a broadcast.

Wire format: 4-byte length, JSON.  bytes travel as {"__b": hex}, numpy arrays as {"__nd": [dtype, shape, hex]}: no pickle
(the sockets are loopback-only, but nothing executable is ever taken from one).
"""
import json
import os
import secrets
import socket
import struct
import tempfile
import time

import numpy as np

_MAGIC = "lnasr-rdzv-1"


def _enc(obj):
    if isinstance(obj, (bytes, bytearray)):
        return {"__b": bytes(obj).hex()}
    if isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        return {"__nd": [a.dtype.str, list(a.shape), a.tobytes().hex()]}
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, (list, tuple)):
        return [_enc(x) for x in obj]
    if isinstance(obj, dict):
        return {str(k): _enc(v) for k, v in obj.items()}
    return obj


def _dec(obj):
    if isinstance(obj, list):
        return [_dec(x) for x in obj]
    if isinstance(obj, dict):
        if set(obj) == {"__b"}:
            return bytes.fromhex(obj["__b"])
        if set(obj) == {"__nd"}:
            dt, shape, hx = obj["__nd"]
            return np.frombuffer(bytes.fromhex(hx), dtype=np.dtype(dt)).reshape(shape).copy()
        return {k: _dec(v) for k, v in obj.items()}
    return obj


def _send(sock, obj):
    data = json.dumps(_enc(obj)).encode()
    sock.sendall(struct.pack("<I", len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += part
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<I", _recv_exact(sock, 4))
    return _dec(json.loads(_recv_exact(sock, n).decode()))


def rendezvous_file(addr=None, port=None):
    """The file rank 0 publishes its listening port in (LNASR_RDZV_FILE overrides)."""
    if os.environ.get("LNASR_RDZV_FILE"):
        return os.environ["LNASR_RDZV_FILE"]
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = port or os.environ.get("MASTER_PORT", "0")
    return os.path.join(tempfile.gettempdir(), "lnasr_rdzv_%s_%s_%d" % (addr.replace(":", "_"), port, os.getuid()))


class TcpGroup:
    """The ranks of one launch.  world == 1: every operation is the identity and no socket is opened."""

    def __init__(self, rank, world, addr=None, port=None, timeout=None):
        self.rank, self.world = int(rank), int(world)
        self.timeout = float(timeout if timeout is not None else os.environ.get("LNASR_RDZV_TIMEOUT", "300"))
        self._peers, self._root, self._listener, self._file = {}, None, None, None
        if self.world <= 1:
            return
        path = rendezvous_file(addr, port)
        deadline = time.time() + self.timeout
        if self.rank == 0:
            try:
                os.unlink(path)                                   # an earlier run's file
            except FileNotFoundError:
                pass
            lst = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            lst.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            lst.bind(("127.0.0.1", 0))
            lst.listen(self.world)
            nonce = secrets.token_hex(8)
            tmp = "%s.%d.tmp" % (path, os.getpid())
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)     # the nonce is for this user's ranks only
            with os.fdopen(fd, "w") as fp:
                fp.write("%d %s\n" % (lst.getsockname()[1], nonce))
            os.replace(tmp, path)                                 # atomic: a reader sees nothing or the whole line
            self._listener, self._file = lst, path
            while len(self._peers) < self.world - 1:
                lst.settimeout(max(0.1, deadline - time.time()))
                try:
                    conn, _ = lst.accept()
                except socket.timeout:
                    raise TimeoutError("rendezvous: %d of %d ranks arrived within %.0f s"
                                       % (len(self._peers) + 1, self.world, self.timeout)) from None
                conn.settimeout(self.timeout)
                try:
                    hello = _recv(conn)
                    ok = (isinstance(hello, dict) and hello.get("magic") == _MAGIC and hello.get("nonce") == nonce and
                          isinstance(hello.get("rank"), int) and 0 < hello["rank"] < self.world and
                          hello["rank"] not in self._peers)
                except Exception:                                 # noqa: BLE001 -- a stranger on the port
                    ok = False
                if not ok:
                    conn.close()
                    continue
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                _send(conn, {"ok": True})
                self._peers[hello["rank"]] = conn
        else:
            while True:
                if time.time() > deadline:
                    raise TimeoutError("rendezvous: rank %d found no rank 0 within %.0f s (%s)" % (self.rank, self.timeout, path))
                try:
                    with open(path) as fp:
                        p, nonce = fp.read().split()
                    s = socket.create_connection(("127.0.0.1", int(p)), timeout=5.0)
                    s.settimeout(self.timeout)
                    _send(s, {"magic": _MAGIC, "nonce": nonce, "rank": self.rank})
                    if _recv(s).get("ok"):
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._root = s
                        break
                    s.close()
                except (OSError, ValueError, ConnectionError):
                    pass                                          # no file yet, a stale one, or rank 0 not listening yet
                time.sleep(0.05)

    # -- collectives (small host values only) ------------------------------------------------------
    def allgather(self, obj):
        """[value of rank 0, ..., value of rank world-1] on every rank"""
        if self.world <= 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [None] * (self.world - 1)
            for r, s in self._peers.items():
                out[r] = _recv(s)
            for s in self._peers.values():
                _send(s, out)
            return out
        _send(self._root, obj)
        return _recv(self._root)

    def gather(self, obj):
        return self.allgather(obj)

    def bcast(self, obj, src=0):
        if self.world <= 1:
            return obj
        return self.allgather(obj if self.rank == src else None)[src]

    def barrier(self):
        self.allgather(None)

    def max(self, x):
        return float(max(self.allgather(float(x))))

    def close(self):
        for s in list(self._peers.values()) + [self._root, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._root, self._listener = {}, None, None
        if self._file:
            try:
                os.unlink(self._file)
            except OSError:
                pass
            self._file = None


def from_env(timeout=None):
    """The group the launcher's environment describes (RANK / WORLD_SIZE; one rank when unset)."""
    return TcpGroup(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), timeout=timeout)
