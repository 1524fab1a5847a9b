"""CPU: the torch-free rendezvous of the RCCL path (deepflows_amd/rendezvous.py + parallel.RcclCommunicator): two
spawned ranks, a fake device module that records what dfhip_comm_init would receive.  Both ranks must end up with
rank 0's 128-byte id, whatever order they start in."""
import json
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, os.environ["DF_ROOT"])
from deepflows_amd import parallel

class FakeMod:
    fused = True
    def __init__(self): self.init = None; self.maxes = []
    def comm_unique_id(self): return bytes((7 * i + 3) % 251 for i in range(128))
    def comm_init(self, rank, world, uid): self.init = (rank, world, bytes(uid).hex())
    def comm_allreduce_max(self, v): self.maxes.append(float(v)); return float(v)
    def comm_destroy(self): pass

class FakeDev:
    mod = FakeMod()

rank = int(os.environ["RANK"])
time.sleep(float(os.environ.get("DF_DELAY_RANK%d" % rank, "0")))      # rank 0 may come up AFTER its peer
comm = parallel.init_communicator(FakeDev)
assert isinstance(comm, parallel.RcclCommunicator) and comm.exchanges
comm.barrier()
json.dump({"rank": comm.rank, "world": comm.world, "init": FakeDev.mod.init, "maxes": FakeDev.mod.maxes},
          open(os.path.join(os.environ["DF_OUT"], "r%d.json" % rank), "w"))
'''


def _spawn(tmp_path, delays):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    port = 31000 + (os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, DF_ROOT=ROOT, DF_OUT=str(tmp_path), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **{f"DF_DELAY_RANK{r}": str(d) for r, d in delays.items()})
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err.decode()[-2000:]
    return [json.load(open(tmp_path / f"r{k}.json")) for k in range(2)]


def test_two_ranks_receive_rank0s_unique_id(tmp_path):
    want = bytes((7 * i + 3) % 251 for i in range(128)).hex()
    for delays in ({}, {0: 0.8}):            # rank 1 connecting before rank 0 listens must retry, not fail
        outs = _spawn(tmp_path, delays)
        for k, o in enumerate(outs):
            assert o["rank"] == k and o["world"] == 2
            assert o["init"] == [k, 2, want]
            assert o["maxes"] == [0.0]        # the barrier is a collective on the communicator


def test_store_address_follows_the_launcher_environment():
    sys.path.insert(0, ROOT)
    from deepflows_amd import rendezvous
    assert rendezvous.store_address({"MASTER_ADDR": "10.0.0.5", "MASTER_PORT": "29500"}) == ("10.0.0.5", 29523)
    assert rendezvous.store_address({"MASTER_PORT": "1000", "DF_STORE_PORT_OFFSET": "1"}) == ("127.0.0.1", 1001)
    assert rendezvous.store_address({"DF_STORE_PORT": "4242"}) == ("127.0.0.1", 4242)
    assert rendezvous.broadcast_from_rank0(b"abc", 0, 1) == b"abc"


def test_the_package_does_not_import_torch():
    code = ("import sys; sys.path.insert(0, %r); import deepflows_amd; from deepflows_amd import parallel, rendezvous, "
            "graph_step, workloads, utils; assert 'torch' not in sys.modules, 'torch was imported'" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_a_rank_of_another_job_is_turned_away(tmp_path):
    """a peer with a valid rank number but another job's token must not take a slot or get the id."""
    import threading
    sys.path.insert(0, ROOT)
    from deepflows_amd import launch, rendezvous
    port = launch.free_port_pair()
    addr = ("127.0.0.1", port)
    mine, other = rendezvous.job_token({"MASTER_PORT": "1", "DF_RUN_ID": "a"}), rendezvous.job_token({"MASTER_PORT": "1", "DF_RUN_ID": "b"})
    assert mine != other and len(mine) == 16
    got = {}

    def serve():
        got[0] = rendezvous.broadcast_from_rank0(b"x" * 128, 0, 2, timeout=30, address=addr, token=mine)
    t = threading.Thread(target=serve)
    t.start()
    import pytest
    with pytest.raises(PermissionError):
        rendezvous.broadcast_from_rank0(b"", 1, 2, timeout=10, address=addr, token=other)
    assert t.is_alive()                                  # the stranger consumed no slot: rank 0 is still waiting
    assert rendezvous.broadcast_from_rank0(b"", 1, 2, timeout=10, address=addr, token=mine) == b"x" * 128
    t.join(timeout=10)
    assert got[0] == b"x" * 128


def test_rank0_falls_back_to_every_interface_when_master_addr_is_not_local():
    import threading
    sys.path.insert(0, ROOT)
    from deepflows_amd import launch, rendezvous
    port = launch.free_port_pair()
    tok = rendezvous.job_token({"DF_RUN_ID": "bind"})
    got = {}

    def serve():                                         # 192.0.2.1 (TEST-NET-1) is no address of this host
        got[0] = rendezvous.broadcast_from_rank0(b"y" * 128, 0, 2, timeout=30, address=("192.0.2.1", port), token=tok)
    t = threading.Thread(target=serve)
    t.start()
    assert rendezvous.broadcast_from_rank0(b"", 1, 2, timeout=20, address=("127.0.0.1", port), token=tok) == b"y" * 128
    t.join(timeout=10)
    assert got[0] == b"y" * 128
