"""Rendezvous of the ranks of one data-parallel job: the only thing the ranks ever exchange on the host is the
128-byte ncclUniqueId that rank 0 draws (dfhip_comm_unique_id) — everything after that, barriers and the
max-over-ranks of a timing included, is a collective on the RCCL communicator.

A launcher (e.g. `python -m torch.distributed.run`, used as a process spawner only) sets RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT.  Rank 0 serves the id from a TCP socket on MASTER_ADDR at MASTER_PORT + DF_STORE_PORT_OFFSET
(default 23: the launcher's own store may own MASTER_PORT itself); every other rank connects (retrying until the
server is up), says its rank and the job's token, gets the bytes.  Standard library only.

The token (16 bytes: a hash of MASTER_ADDR, MASTER_PORT and the launcher's run id -- TORCHELASTIC_RUN_ID or
launch.py's DF_RUN_ID) keeps two jobs that meet on one host and port apart: a rank of another job is turned away
instead of taking a slot and receiving the wrong id.  If MASTER_ADDR cannot be bound on this host (a name that
resolves to another interface), rank 0 listens on every interface instead.
"""
import hashlib
import os
import socket
import struct
import time

_MAGIC = b"DFRV2"
_REFUSED = -2


def job_token(env=None):
    env = os.environ if env is None else env
    text = "|".join((env.get("MASTER_ADDR", "127.0.0.1"), env.get("MASTER_PORT", "29500"),
                     env.get("TORCHELASTIC_RUN_ID") or env.get("DF_RUN_ID") or "", env.get("WORLD_SIZE", "1")))
    return hashlib.sha256(text.encode()).digest()[:16]


def store_address(env=None):
    env = os.environ if env is None else env
    host = env.get("MASTER_ADDR", "127.0.0.1")
    port = int(env.get("DF_STORE_PORT") or int(env.get("MASTER_PORT", "29500")) + int(env.get("DF_STORE_PORT_OFFSET", "23")))
    return host, port


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        part = conn.recv(n - len(buf))
        if not part:
            raise ConnectionError("rendezvous: peer closed the connection")
        buf += part
    return buf


def broadcast_from_rank0(payload, rank, world, timeout=120.0, address=None, token=None):
    """Every rank returns rank 0's `payload` (bytes; only rank 0's argument is used).  Rank 0 answers exactly
    world - 1 peers, each of which must present the job's token and a distinct rank in 1 .. world - 1."""
    if world <= 1:
        return bytes(payload)
    host, port = address or store_address()
    token = job_token() if token is None else bytes(token)
    if len(token) != 16:
        raise ValueError("rendezvous: the job token is 16 bytes")
    deadline = time.monotonic() + timeout
    if rank == 0:
        payload = bytes(payload)
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            srv.bind((host, port))
        except OSError as first:
            try:
                srv.bind(("", port))                   # MASTER_ADDR is not an address of this host: every interface
            except OSError:
                srv.close()
                raise first
        srv.listen(world)
        seen = set()
        try:
            while len(seen) < world - 1:
                srv.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise TimeoutError(f"rendezvous: only {len(seen) + 1} of {world} ranks arrived at {host}:{port}")
                with conn:
                    conn.settimeout(10.0)
                    try:
                        hello = _recv_exact(conn, len(_MAGIC) + 4 + 16)
                    except (ConnectionError, socket.timeout):
                        continue                                   # a port scanner, not a rank
                    if hello[:len(_MAGIC)] != _MAGIC:
                        continue
                    peer = struct.unpack("<i", hello[len(_MAGIC):len(_MAGIC) + 4])[0]
                    if hello[len(_MAGIC) + 4:] != token or not (0 < peer < world) or peer in seen:
                        try:                                       # a rank of another job (or a duplicate): say so
                            conn.sendall(struct.pack("<i", _REFUSED))
                        except OSError:
                            pass
                        continue
                    conn.sendall(struct.pack("<i", len(payload)) + payload)
                    seen.add(peer)
        finally:
            srv.close()
        return payload
    while True:
        try:
            with socket.create_connection((host, port), timeout=5.0) as conn:
                conn.sendall(_MAGIC + struct.pack("<i", rank) + token)
                n = struct.unpack("<i", _recv_exact(conn, 4))[0]
                if n == _REFUSED:
                    raise PermissionError(f"rendezvous: rank 0 at {host}:{port} belongs to another job (or rank "
                                          f"{rank} is already present): token / rank refused")
                return _recv_exact(conn, n)
        except PermissionError:
            raise
        except (ConnectionRefusedError, ConnectionResetError, ConnectionError, socket.timeout, OSError):
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: rank {rank} could not reach rank 0 at {host}:{port}")
            time.sleep(0.05)
