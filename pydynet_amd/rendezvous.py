"""Host-side rendezvous for the RCCL communicator: rank 0's 128-byte unique id must reach every
other rank before `pdn_comm_init`.  The reference has nothing of the kind (no multi-process code,
SURVEY 2a); launchers (`python -m torch.distributed.run`, mpirun-style scripts) only provide
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment.  MASTER_PORT itself belongs to
the launcher's own store, so rank 0 serves the payload on the first free port ABOVE it and the other
ranks probe that short range; a handshake (magic, world size, sequence number, and a job token hashed from
MASTER_ADDR:MASTER_PORT plus PDN_JOB_ID / TORCHELASTIC_RUN_ID when set) rejects anything else that may be
listening there -- including another job of the same world size on an adjacent MASTER_PORT -- and rank 0
binds MASTER_ADDR itself when that address is local instead of every interface.  Plain sockets, no third-party dependency, one node or many.
"""
from __future__ import annotations

import hashlib
import os
import socket
import struct
import time

_MAGIC = b"PDNRDZV1"
_RANGE = 48                     # ports probed above MASTER_PORT
_seq = [0]                      # exchanges performed by this process (keeps successive groups apart)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def _endpoint():
    return os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500"))


def _job_token() -> bytes:
    addr, port = _endpoint()
    ident = f"{addr}:{port}:{os.environ.get('PDN_JOB_ID', '')}:{os.environ.get('TORCHELASTIC_RUN_ID', '')}"
    return hashlib.sha256(ident.encode()).digest()[:8]


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return buf


def _bind_host(addr: str, world: int) -> str:
    """The address rank 0 listens on: MASTER_ADDR, unless it resolves to loopback HERE while the job spans nodes.
    (A container's /etc/hosts often maps its own hostname to 127.0.1.1: rank 0 would then listen on loopback only
    while the other nodes resolve the same name to the real address and get `connection refused`.)  In that case --
    and when the name does not resolve, or PDN_RDZV_BIND_ALL=1 -- all interfaces; the job-token handshake keeps
    strangers out."""
    if os.environ.get("PDN_RDZV_BIND_ALL") == "1":
        return ""
    try:
        ips = {info[4][0] for info in socket.getaddrinfo(addr, None, socket.AF_INET)}
    except (socket.gaierror, UnicodeError):
        return ""
    try:
        local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        nnodes = int(os.environ.get("GROUP_WORLD_SIZE", os.environ.get("NNODES", "1")))
    except ValueError:
        local, nnodes = world, 1
    spans_nodes = local < world or nnodes > 1
    if spans_nodes and ips and all(ip.startswith("127.") for ip in ips):
        return ""
    return addr


def broadcast_bytes(payload: bytes | None, rank: int, world: int, timeout: float = 600.0) -> bytes:
    """Rank 0 passes `payload`; every rank returns rank 0's payload.  Blocks until all `world - 1`
    peers have fetched it (rank 0) or until it is received (others); raises TimeoutError."""
    seq = _seq[0]
    _seq[0] += 1
    if world == 1:
        return payload
    addr, base = _endpoint()
    hello = _MAGIC + _job_token() + struct.pack("<ii", world, seq)
    deadline = time.monotonic() + timeout
    if rank == 0:
        assert payload is not None
        srv = None
        for off in range(1, _RANGE + 1):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                try:
                    s.bind((_bind_host(addr, world), base + off))   # MASTER_ADDR when it is one of this host's addresses
                except OSError as e:
                    import errno
                    if e.errno != errno.EADDRNOTAVAIL and not isinstance(e, socket.gaierror):
                        raise
                    s.bind(("", base + off))
                s.listen(world + 8)
                srv = s
                break
            except OSError:
                s.close()
        if srv is None:
            raise RuntimeError(f"rendezvous: no free port in ({base}, {base + _RANGE}]")
        served = set()
        srv.settimeout(1.0)
        try:
            while len(served) < world - 1:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: only {len(served)} of {world - 1} peers arrived")
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                with conn:
                    conn.settimeout(5.0)
                    try:
                        msg = _recv_exact(conn, len(hello) + 4)
                        if msg[:len(hello)] != hello:
                            conn.sendall(b"NO")
                            continue
                        peer = struct.unpack("<i", msg[len(hello):])[0]
                        if not 0 < peer < world:
                            conn.sendall(b"NO")
                            continue
                        conn.sendall(b"OK" + struct.pack("<i", len(payload)) + payload)
                        served.add(peer)
                    except (OSError, ConnectionError):
                        continue
        finally:
            srv.close()
        return payload
    delay = 0.02
    while True:
        for off in range(1, _RANGE + 1):
            try:
                with socket.create_connection((addr, base + off), timeout=2.0) as conn:
                    conn.settimeout(5.0)
                    conn.sendall(hello + struct.pack("<i", rank))
                    if _recv_exact(conn, 2) != b"OK":
                        continue
                    n = struct.unpack("<i", _recv_exact(conn, 4))[0]
                    return _recv_exact(conn, n)
            except (OSError, ConnectionError):
                continue
        if time.monotonic() > deadline:
            raise TimeoutError(f"rendezvous: rank {rank} could not reach rank 0 at {addr}:{base + 1}..{base + _RANGE}")
        time.sleep(delay)
        delay = min(delay * 1.5, 0.5)
