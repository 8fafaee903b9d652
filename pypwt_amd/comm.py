"""Communicator -- RCCL point-to-point exchange through the C ABI (pdwt_comm_*, include/pypwt_amd.h), no torch.

One process per GPU.  `Communicator.from_env()` reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (the
variables torch.distributed.run and bench.py's self-launch export) and passes rank 0's RCCL unique id to the other ranks
over one TCP connection each; `Communicator.single()` is a ring of one (the rank is its own neighbour: how the transport
runs on a one-GPU box).  Used by pypwt_amd.tiled.TiledWavelets(comm=...) for the halo rows of a tiled image.

`HostRing` is the same ring over TCP with the messages staged on the host: RCCL wants one GPU per rank, so several ranks SHARING a
GPU -- the tests of the tiled path on a one-GPU box -- talk through it (TiledWavelets(ring=...)); it is not a fast path.
"""
import ctypes as C
import os
import socket
import time

from . import _lib
from ._lib import handle_t

ID_BYTES = 128


def _check(lib, rc, what):
    if rc < 0:
        msg = lib.pdwt_comm_last_error()
        raise RuntimeError("%s: %s" % (what, msg.decode("utf-8", "replace") if msg else "error %d" % rc))
    return rc


HELLO = b"PDWTID1 "  # + b"<rank> <nonce>\n"


def _nonce():
    """Ranks of ONE job agree on it without talking: the launcher exports the same TORCHELASTIC_RUN_ID / PDWT_COMM_NONCE /
    MASTER_PORT to all of them.  A stray connection (a port scanner, a rank of a previous run) does not know it."""
    return (os.environ.get("PDWT_COMM_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID") or "port%s" % os.environ.get("MASTER_PORT", "29500")).encode()


def _recv_line(conn, limit=256):
    buf = b""
    while not buf.endswith(b"\n") and len(buf) < limit:
        chunk = conn.recv(1)
        if not chunk:
            break
        buf += chunk
    return buf


def _share_id(rank, size, unique_id, addr, port, timeout=120.0):
    """rank 0 -> every other rank: the 128 bytes of the RCCL unique id, one TCP connection per rank.  A client says who it is
    (`PDWTID1 <rank> <nonce>`); rank 0 serves each of the ranks 1 .. size-1 once and drops every connection that does not
    say so (it does not count against the ranks it is waiting for).  IPv4 and IPv6 (socket.create_server / getaddrinfo).
    The id can be carried by any other channel instead (a file, MPI): Communicator(rank, size, unique_id)."""
    if size == 1:
        return unique_id
    nonce = _nonce()
    if rank == 0:
        family = socket.AF_INET6 if ":" in addr else socket.AF_INET
        deadline = time.time() + timeout
        with socket.create_server((addr, port), family=family, backlog=max(8, size), reuse_port=False) as srv:
            waiting = set(range(1, size))
            while waiting:
                srv.settimeout(max(0.1, deadline - time.time()))
                if time.time() > deadline:
                    raise TimeoutError("Communicator: ranks %s never asked for the unique id" % sorted(waiting))
                conn, _ = srv.accept()
                with conn:
                    conn.settimeout(5.0)
                    try:
                        hello = _recv_line(conn)
                        parts = hello[len(HELLO):].split()
                        if not hello.startswith(HELLO) or len(parts) != 2 or parts[1] != nonce or int(parts[0]) not in waiting:
                            continue  # not one of ours (or a rank served already): dropped, nobody's turn is used up
                        conn.sendall(unique_id)
                        waiting.discard(int(parts[0]))
                    except (OSError, ValueError):
                        continue
        return unique_id
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as s:
                s.sendall(HELLO + b"%d " % rank + nonce + b"\n")
                buf = b""
                while len(buf) < ID_BYTES:
                    chunk = s.recv(ID_BYTES - len(buf))
                    if not chunk:
                        raise ConnectionError("rank 0 closed the connection early")
                    buf += chunk
                return buf
        except (ConnectionRefusedError, socket.timeout, ConnectionError, OSError):
            if time.time() > deadline:
                raise
            time.sleep(0.05)


class Communicator(object):
    def __init__(self, rank, size, unique_id, device=-1):
        self._lib = _lib.load()
        self._h = None
        if len(unique_id) != ID_BYTES:
            raise ValueError("Communicator: the unique id is %d bytes" % ID_BYTES)
        h = handle_t()
        idbuf = (C.c_char * ID_BYTES).from_buffer_copy(unique_id)
        _check(self._lib, self._lib.pdwt_comm_create(C.cast(idbuf, C.c_void_p), int(size), int(rank), int(device), C.byref(h)),
               "pdwt_comm_create")
        self._h = h
        self.rank, self.size = int(rank), int(size)

    @staticmethod
    def unique_id():
        lib = _lib.load()
        buf = (C.c_char * ID_BYTES)()
        _check(lib, lib.pdwt_comm_unique_id(C.cast(buf, C.c_void_p)), "pdwt_comm_unique_id")
        return bytes(buf)

    @classmethod
    def single(cls, device=-1):
        """a ring of one rank: it is its own neighbour (the periodic image closes on itself through RCCL)"""
        return cls(0, 1, cls.unique_id(), device)

    @classmethod
    def from_env(cls, device=None, port_offset=17):
        rank, size = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29500")) + port_offset  # next to, not on, the launcher's own port
        uid = cls.unique_id() if rank == 0 else None
        uid = _share_id(rank, size, uid, addr, port)
        return cls(rank, size, uid, device)

    # ---- transfers, all enqueued on `stream` (an integer HIP stream handle; 0 / None = the legacy default stream)
    def exchange(self, sends, recvs, stream=None):
        """sends: [(device pointer, count of values, peer)], recvs: the same -- ONE grouped RCCL call"""
        n = max(len(sends), len(recvs))
        sp, sc, sr = (C.c_void_p * n)(), (C.c_longlong * n)(), (C.c_int * n)()
        rp, rc_, rr = (C.c_void_p * n)(), (C.c_longlong * n)(), (C.c_int * n)()
        for i, (p, cnt, peer) in enumerate(sends):
            sp[i], sc[i], sr[i] = p, cnt, peer
        for i, (p, cnt, peer) in enumerate(recvs):
            rp[i], rc_[i], rr[i] = p, cnt, peer
        _check(self._lib, self._lib.pdwt_comm_exchange(self._h, n, sp, sc, sr, rp, rc_, rr, C.c_void_p(stream or 0)),
               "pdwt_comm_exchange")

    def prepare(self, sends, recvs):
        """the argument arrays of exchange(), built once for a message list that repeats (a level's halos)"""
        n = max(len(sends), len(recvs))
        sp, sc, sr = (C.c_void_p * n)(), (C.c_longlong * n)(), (C.c_int * n)()
        rp, rc_, rr = (C.c_void_p * n)(), (C.c_longlong * n)(), (C.c_int * n)()
        for i, (p, cnt, peer) in enumerate(sends):
            sp[i], sc[i], sr[i] = p, cnt, peer
        for i, (p, cnt, peer) in enumerate(recvs):
            rp[i], rc_[i], rr[i] = p, cnt, peer
        return (n, sp, sc, sr, rp, rc_, rr)

    def exchange_prepared(self, prep, stream=None):
        _check(self._lib, self._lib.pdwt_comm_exchange(self._h, prep[0], prep[1], prep[2], prep[3], prep[4], prep[5], prep[6],
                                                       C.c_void_p(stream or 0)), "pdwt_comm_exchange")

    def all_gather(self, send_ptr, recv_ptr, count_per_rank, stream=None):
        _check(self._lib, self._lib.pdwt_comm_all_gather(self._h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), int(count_per_rank),
                                                         C.c_void_p(stream or 0)), "pdwt_comm_all_gather")

    def broadcast(self, ptr, count, root=0, stream=None):
        _check(self._lib, self._lib.pdwt_comm_broadcast(self._h, C.c_void_p(ptr), int(count), int(root), C.c_void_p(stream or 0)),
               "pdwt_comm_broadcast")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pdwt_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostRing(object):
    """A ring of `size` ranks over TCP.  Every rank listens on a port the system picks, tells rank 0 (which listens on `port`, the
    only agreed one: MASTER_PORT + port_offset, or PDWT_RING_PORT) and gets every rank's port back; then it connects to the next
    rank and accepts its previous one.  Messages are host byte strings (TiledWavelets stages its halo rows through pdwt_copy).
    For ranks of ONE HOST that share a GPU (tests) or have no RCCL: every rank binds its listener to `addr` and looks for its
    neighbour there, so `addr` must be an address of the machine all ranks run on (a rank on another node fails to bind with
    EADDRNOTAVAIL -- several nodes need RCCL, `Communicator`).  Hello line and job nonce as in the unique-id rendezvous."""

    def __init__(self, rank, size, addr="127.0.0.1", port=29540, timeout=120.0):
        import threading
        self._threading = threading
        self.rank, self.size = int(rank), int(size)
        self._next = self._prev = None
        if self.size == 1:
            return
        nonce = _nonce()
        family = socket.AF_INET6 if ":" in addr else socket.AF_INET
        mine = socket.create_server((addr, 0), family=family, backlog=4)
        mine.settimeout(timeout)
        my_port = mine.getsockname()[1]
        # ---- every rank's port, through rank 0
        if self.rank == 0:
            ports, conns = {0: my_port}, []
            deadline = time.time() + timeout
            with socket.create_server((addr, int(port)), family=family, backlog=max(8, self.size)) as srv:
                while len(ports) < self.size:
                    if time.time() > deadline:
                        raise TimeoutError("HostRing: ranks %s never reported" % sorted(set(range(self.size)) - set(ports)))
                    srv.settimeout(max(0.1, deadline - time.time()))
                    conn, _ = srv.accept()
                    conn.settimeout(5.0)
                    try:
                        hello = _recv_line(conn)
                        parts = hello[len(HELLO):].split() if hello.startswith(HELLO) else []
                        if len(parts) != 3 or parts[1] != nonce or int(parts[0]) in ports or not 0 < int(parts[0]) < self.size:
                            conn.close()
                            continue
                        ports[int(parts[0])] = int(parts[2])
                        conns.append(conn)
                    except (OSError, ValueError):
                        conn.close()
            table = (" ".join(str(ports[r]) for r in range(self.size)) + "\n").encode()
            for conn in conns:
                with conn:
                    conn.sendall(table)
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    with socket.create_connection((addr, int(port)), timeout=5.0) as s0:
                        s0.settimeout(timeout)
                        s0.sendall(HELLO + b"%d " % self.rank + nonce + b" %d\n" % my_port)
                        table = _recv_line(s0, limit=16 * self.size + 16)
                    if not table.endswith(b"\n"):
                        raise ConnectionError("rank 0 closed the connection early")
                    break
                except (ConnectionRefusedError, socket.timeout, ConnectionError, OSError):
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
        ports = [int(x) for x in table.split()]
        # ---- the ring: connect to the next rank, accept the previous one
        accepted = {}

        def accept():
            deadline = time.time() + timeout
            while time.time() < deadline:
                conn, _ = mine.accept()
                conn.settimeout(timeout)
                hello = _recv_line(conn)
                parts = hello[len(HELLO):].split() if hello.startswith(HELLO) else []
                if len(parts) == 2 and parts[1] == nonce and int(parts[0]) == (self.rank - 1) % self.size:
                    accepted["conn"] = conn
                    return
                conn.close()

        th = threading.Thread(target=accept, daemon=True)
        th.start()
        nxt = socket.create_connection((addr, ports[(self.rank + 1) % self.size]), timeout=timeout)
        nxt.settimeout(timeout)
        nxt.sendall(HELLO + b"%d " % self.rank + nonce + b"\n")
        th.join(timeout)
        mine.close()
        if "conn" not in accepted:
            raise TimeoutError("HostRing: rank %d never heard from its previous rank" % self.rank)
        self._next, self._prev = nxt, accepted["conn"]
        for c in (self._next, self._prev):
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    @classmethod
    def from_env(cls, port_offset=40):
        port = int(os.environ["PDWT_RING_PORT"]) if os.environ.get("PDWT_RING_PORT") else int(os.environ.get("MASTER_PORT", "29500")) + port_offset
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), os.environ.get("MASTER_ADDR", "127.0.0.1"), port)

    @staticmethod
    def _send(conn, payload):
        conn.sendall(len(payload).to_bytes(8, "little") + payload)

    @staticmethod
    def _recv(conn):
        def exactly(n):
            buf = bytearray()
            while len(buf) < n:
                chunk = conn.recv(min(1 << 20, n - len(buf)))
                if not chunk:
                    raise ConnectionError("HostRing: a neighbour closed the connection")
                buf += chunk
            return bytes(buf)
        return exactly(int.from_bytes(exactly(8), "little"))

    def sendrecv(self, to_prev, to_next):
        """-> (from_prev, from_next): what the previous rank sent to its next and the next rank to its previous"""
        if self.size == 1:
            return to_next, to_prev
        err = []

        def send():
            try:
                self._send(self._next, to_next)
                self._send(self._prev, to_prev)
            except Exception as e:  # noqa: BLE001
                err.append(e)
        th = self._threading.Thread(target=send, daemon=True)
        th.start()
        from_prev = self._recv(self._prev)
        from_next = self._recv(self._next)
        th.join()
        if err:
            raise err[0]
        return from_prev, from_next

    def all_gather(self, mine):
        """-> the ranks' byte strings in rank order (size - 1 steps around the ring)"""
        parts = {self.rank: mine}
        cur = mine
        for step in range(1, self.size):
            cur, _ = self.sendrecv(b"", cur)
            parts[(self.rank - step) % self.size] = cur
        return [parts[r] for r in range(self.size)]

    def broadcast(self, payload, root=0):
        """-> root's byte string on every rank (passed along the ring)"""
        if self.size == 1:
            return payload
        if self.rank != root:
            payload = self._recv(self._prev)
        if (self.rank + 1) % self.size != root:
            self._send(self._next, payload)
        return payload

    def barrier(self):
        self.all_gather(b"")

    def close(self):
        for c in (self._next, self._prev):
            if c is not None:
                try:
                    c.close()
                except OSError:
                    pass
        self._next = self._prev = None
