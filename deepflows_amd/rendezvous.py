"""Rendezvous of the ranks of one data-parallel job: the only thing the ranks ever exchange on the host is the
128-byte ncclUniqueId that rank 0 draws (dfhip_comm_unique_id) — everything after that, barriers and the
max-over-ranks of a timing included, is a collective on the RCCL communicator.

A launcher (e.g. `python -m torch.distributed.run`, used as a process spawner only) sets RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT.  Rank 0 serves the id from a TCP socket on MASTER_ADDR at MASTER_PORT + DF_STORE_PORT_OFFSET
(default 23: the launcher's own store may own MASTER_PORT itself); every other rank connects (retrying until the
server is up), says its rank, gets the bytes.  Standard library only.
"""
import os
import socket
import struct
import time

_MAGIC = b"DFRV1"


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


def broadcast_from_rank0(payload, rank, world, timeout=120.0, address=None):
    """Every rank returns rank 0's `payload` (bytes; only rank 0's argument is used).  Rank 0 answers exactly
    world - 1 peers, each of which must present a distinct rank in 1 .. world - 1."""
    if world <= 1:
        return bytes(payload)
    host, port = address or store_address()
    deadline = time.monotonic() + timeout
    if rank == 0:
        payload = bytes(payload)
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((host, port))
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
                        hello = _recv_exact(conn, len(_MAGIC) + 4)
                    except (ConnectionError, socket.timeout):
                        continue                                   # a port scanner, not a rank
                    peer = struct.unpack("<i", hello[len(_MAGIC):])[0]
                    if hello[:len(_MAGIC)] != _MAGIC or not (0 < peer < world) or peer in seen:
                        continue
                    conn.sendall(struct.pack("<i", len(payload)) + payload)
                    seen.add(peer)
        finally:
            srv.close()
        return payload
    while True:
        try:
            with socket.create_connection((host, port), timeout=5.0) as conn:
                conn.sendall(_MAGIC + struct.pack("<i", rank))
                n = struct.unpack("<i", _recv_exact(conn, 4))[0]
                return _recv_exact(conn, n)
        except (ConnectionRefusedError, ConnectionResetError, ConnectionError, socket.timeout, OSError):
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: rank {rank} could not reach rank 0 at {host}:{port}")
            time.sleep(0.05)
