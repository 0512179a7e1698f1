"""HostRendezvous — the control plane of a one-process-per-GPU job, on plain TCP sockets.

The reference is single-process; the N > 1 path needs exactly three tiny host-side exchanges around the
one RCCL gather: ship the 128-byte RCCL unique id from rank 0 to everyone, agree that every rank brought its
communicator up, and tell each other how many bytes of text each rank formatted.  The product does this itself — no
PyTorch, no gloo (BASELINE.json north_star) — with a star of TCP connections to rank 0.

Launch contract: the usual RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT environment (``torch.distributed.run`` or any
other launcher).  MASTER_PORT itself is NOT used: under torch's elastic agent that port belongs to the agent's own
store.  Rank 0 binds the first free port of ``MASTER_PORT + 1 … + 32`` (or ``TIMED_RDZV_PORT``), the other ranks probe
those ports and recognise the right listener by a job token (address, port, run id, world size) in the handshake, so a
foreign service on one of the ports is skipped, not mistaken for the job.

Every payload is length-framed bytes; nothing is unpickled.
"""
from __future__ import annotations

import os
import socket
import struct
import time
from typing import List, Optional, Sequence

_MAGIC = b"THRDZV01"
_PORT_SPAN = 32


def _send(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    chunks, got = [], 0
    while got < n:
        b = sock.recv(min(n - got, 1 << 20))
        if not b:
            raise ConnectionError("rendezvous peer closed the connection")
        chunks.append(b)
        got += len(b)
    return b"".join(chunks)


def _recv(sock: socket.socket, limit: int = 1 << 30) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > limit:
        raise ConnectionError(f"rendezvous frame of {n} bytes exceeds the limit")
    return _recv_exact(sock, n)


class HostRendezvous:
    """Star of TCP connections to rank 0: ``broadcast`` (from rank 0), ``allgather``, ``barrier``, ``all_min``."""

    def __init__(self, rank: int, world: int, addr: Optional[str] = None, port: Optional[int] = None, timeout: float = 120.0):
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world {world}")
        self.rank, self.world = rank, world
        self._peers: List[Optional[socket.socket]] = [None] * world     # rank 0: one socket per other rank
        self._root: Optional[socket.socket] = None                      # other ranks: the socket to rank 0
        if world == 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(os.environ.get("MASTER_PORT", "29500"))
        fixed = port if port is not None else (int(os.environ["TIMED_RDZV_PORT"]) if os.environ.get("TIMED_RDZV_PORT") else None)
        ports = [fixed] if fixed is not None else [base + 1 + k for k in range(_PORT_SPAN) if base + 1 + k < 65536]
        # what a peer must present: the job's coordinates plus, when the launcher exports one, a shared secret
        # (TIMED_RDZV_SECRET) — the coordinates alone are guessable from outside the job
        token = "|".join([addr, str(base), os.environ.get("TORCHELASTIC_RUN_ID", ""), str(world),
                          os.environ.get("TIMED_RDZV_SECRET", "")]).encode()
        self._bind_addr = addr
        deadline = time.monotonic() + timeout
        if rank == 0:
            self._serve(ports, token, deadline)
        else:
            self._join(addr, ports, token, deadline)

    def _listen_address(self) -> str:
        mode = os.environ.get("TIMED_RDZV_BIND", "")
        if mode == "wildcard":
            return ""
        if mode == "strict":
            return self._bind_addr
        try:
            resolved = socket.gethostbyname(self._bind_addr)
        except OSError:
            return ""
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "0") or 0)
        multi_node = local_world > 0 and self.world > local_world
        is_literal = resolved == self._bind_addr                      # MASTER_ADDR was given as an IPv4 literal
        if resolved.startswith("127.") and multi_node and not is_literal:
            return ""                                                 # a name that is loopback HERE only: peers on other nodes need the wildcard
        return self._bind_addr

    # ---- connection set-up -----------------------------------------------------------------------------------------
    def _serve(self, ports: Sequence[int], token: bytes, deadline: float) -> None:
        srv = None
        for p in ports:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                # listen on MASTER_ADDR only (loopback for a one-node job), not on every interface; an address that is not
                # local to this host (a name that resolves elsewhere, a NAT'd address) falls back to the wildcard — and so does
                # a multi-node job whose MASTER_ADDR is a host NAME that rank 0's own /etc/hosts maps to loopback (Debian's
                # 127.0.1.1 line): the bind would succeed on loopback and the other nodes, resolving the real address, would be
                # refused until the timeout.  TIMED_RDZV_BIND=wildcard|strict overrides the choice.
                try:
                    s.bind((self._listen_address(), p))
                except (OSError, socket.gaierror) as e:
                    if getattr(e, "errno", None) == 98:          # EADDRINUSE: this port is taken, try the next one
                        raise
                    s.bind(("", p))
                s.listen(self.world + 8)
                srv = s
                break
            except OSError:
                s.close()
        if srv is None:
            raise RuntimeError(f"rendezvous: rank 0 found no free port among {list(ports)[:4]}…")
        try:
            missing = self.world - 1
            while missing:
                left = deadline - time.monotonic()
                if left <= 0:
                    raise TimeoutError(f"rendezvous: {missing} of {self.world - 1} ranks never connected to rank 0")
                srv.settimeout(min(left, 5.0))
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                try:
                    c.settimeout(5.0)
                    hello = _recv(c, 4096)
                    ok = hello[:8] == _MAGIC and hello[12:] == token
                    r = struct.unpack("<I", hello[8:12])[0] if len(hello) >= 12 else 0
                    if not ok or not (0 < r < self.world) or self._peers[r] is not None:
                        c.close()
                        continue
                    _send(c, b"OK")
                    c.settimeout(None)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._peers[r] = c
                    missing -= 1
                except (OSError, ConnectionError, struct.error):
                    c.close()
        finally:
            srv.close()

    def _join(self, addr: str, ports: Sequence[int], token: bytes, deadline: float) -> None:
        hello = _MAGIC + struct.pack("<I", self.rank) + token
        while True:
            for p in ports:
                try:
                    c = socket.create_connection((addr, p), timeout=2.0)
                except OSError:
                    continue
                try:
                    c.settimeout(5.0)
                    _send(c, hello)
                    if _recv(c, 16) == b"OK":
                        c.settimeout(None)
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._root = c
                        return
                except (OSError, ConnectionError, struct.error):
                    pass
                c.close()
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: rank {self.rank} could not reach rank 0 at {addr} (ports {ports[0]}…{ports[-1]})")
            time.sleep(0.05)

    # ---- collectives over the star ---------------------------------------------------------------------------------
    def allgather(self, payload: bytes) -> List[bytes]:
        """every rank's payload, in rank order, on every rank"""
        if self.world == 1:
            return [bytes(payload)]
        if self.rank == 0:
            parts = [bytes(payload)] + [_recv(self._peers[r]) for r in range(1, self.world)]
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
            for r in range(1, self.world):
                _send(self._peers[r], blob)
            return parts
        _send(self._root, bytes(payload))
        blob = _recv(self._root)
        parts, pos = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, pos)
            parts.append(blob[pos + 8: pos + 8 + n])
            pos += 8 + n
        return parts

    def broadcast(self, payload: Optional[bytes]) -> bytes:
        """rank 0's payload on every rank"""
        return self.allgather(payload if self.rank == 0 and payload is not None else b"")[0]

    def allgather_ints(self, values: Sequence[int]) -> List[List[int]]:
        parts = self.allgather(struct.pack(f"<{len(values)}q", *[int(v) for v in values]))
        return [list(struct.unpack(f"<{len(p) // 8}q", p)) for p in parts]

    def all_min(self, value: int) -> int:
        return min(v[0] for v in self.allgather_ints([value]))

    def all_max_float(self, value: float) -> float:
        return max(struct.unpack("<d", p)[0] for p in self.allgather(struct.pack("<d", float(value))))

    def barrier(self) -> None:
        self.allgather(b"")

    def close(self) -> None:
        for s in self._peers + [self._root]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers = [None] * self.world
        self._root = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
