"""CPU: `python bench.py --gpus N` launches its N ranks by itself (deepflows_amd/launch.py).  The
spawner is exercised with a stub worker (environment, exit-code propagation, JSON forwarding), and bench.py itself
is run with --gpus 2 on this GPU-less box: it must fail INSIDE the ranks, at Device('hip'), not while parsing."""
import io
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)

STUB = r'''
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = sys.argv[1]
print("hello from", rank, flush=True)
if mode == "fail" and rank == 1:
    sys.exit(7)
if mode == "fail":
    time.sleep(60)                                   # a survivor stuck in a collective
if mode == "early-json" and rank == 0:
    print(json.dumps({"metric": "not the last one"}), flush=True)
if rank == 0:
    print(json.dumps({"metric": "stub", "world": world, "local_rank": os.environ["LOCAL_RANK"],
                      "addr": os.environ["MASTER_ADDR"], "port": int(os.environ["MASTER_PORT"]),
                      "run_id": os.environ["DF_RUN_ID"], "ipc": os.environ["HSA_ENABLE_IPC_MODE_LEGACY"]}), flush=True)
    print("trailing library banner", flush=True)
'''


def _run(tmp_path, mode, world=3, **kw):
    from deepflows_amd import launch
    script = tmp_path / "stub.py"
    script.write_text(STUB)
    out, err = io.StringIO(), io.StringIO()
    rc, last = launch.spawn_ranks(world, [sys.executable, str(script), mode], out=out, err=err, **kw)
    return rc, last, out.getvalue(), err.getvalue()


def test_ranks_get_the_launcher_environment_and_rank0s_json_comes_last(tmp_path):
    rc, last, out, err = _run(tmp_path, "ok")
    assert rc == 0
    rec = json.loads(last)
    assert rec["world"] == 3 and rec["local_rank"] == "0" and rec["addr"] == "127.0.0.1" and rec["ipc"] == "0"
    assert rec["port"] > 0 and rec["run_id"]
    lines = out.strip().splitlines()
    assert lines[-1] == last                              # held back until every rank has exited
    assert "hello from 0" in out and "trailing library banner" in out
    assert "[rank 1] hello from 1" in err and "[rank 2] hello from 2" in err and "hello from 1" not in out


def test_only_the_last_json_line_is_held_back(tmp_path):
    rc, last, out, _ = _run(tmp_path, "early-json", world=2)
    assert rc == 0 and json.loads(last)["metric"] == "stub"
    lines = out.strip().splitlines()
    assert json.loads(lines[-1])["metric"] == "stub"
    assert any(l.startswith("{") and json.loads(l)["metric"] == "not the last one" for l in lines[:-1])


def test_a_failed_rank_fails_the_job_and_the_survivors_are_terminated(tmp_path):
    import time
    t0 = time.monotonic()
    rc, last, out, err = _run(tmp_path, "fail", world=3, grace=1.0)
    assert rc == 7 and last is None
    assert time.monotonic() - t0 < 30                     # did not wait for the 60 s sleepers
    assert "rank 1 exited with 7" in err


def test_free_port_pair_is_bindable():
    import socket
    from deepflows_amd import launch
    p = launch.free_port_pair()
    for q in (p, p + launch.STORE_PORT_OFFSET):
        with socket.socket() as s:
            s.bind(("127.0.0.1", q))


def test_bench_gpus_2_fails_inside_the_ranks_not_at_argument_parsing():
    from deepflows_amd import hip_device
    if hip_device.available():
        pytest.skip("a GPU is visible here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode != 0
    assert "needs `python -m torch.distributed.run" not in r.stderr
    assert "[launch] rank" in r.stderr                    # the spawner reported a rank's exit ...
    assert "HIP" in r.stderr or "hip" in r.stderr         # ... whose own error is the missing device


def test_a_terminated_spawner_leaves_no_rank_behind(tmp_path):
    """SIGTERM to the spawner (a job timeout): the ranks it started -- which would otherwise wait in a collective forever and
    hold their GPUs -- are terminated by PID before it exits (spawn_ranks never cleaned up on an abnormal exit)."""
    import signal
    import time
    worker = tmp_path / "sleeper.py"
    worker.write_text("import os, time\n"
                      "open(os.path.join(%r, 'pid.' + os.environ['RANK']), 'w').write(str(os.getpid()))\n"
                      "time.sleep(600)\n" % str(tmp_path))
    spawner = tmp_path / "spawner.py"
    spawner.write_text("import sys\nsys.path.insert(0, %r)\nfrom deepflows_amd import launch\n"
                       "rc, _ = launch.spawn_ranks(2, [sys.executable, %r])\nsys.exit(rc)\n" % (ROOT, str(worker)))
    p = subprocess.Popen([sys.executable, str(spawner)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        t_end = time.monotonic() + 30
        while not all((tmp_path / f"pid.{r}").exists() and (tmp_path / f"pid.{r}").read_text() for r in range(2)):
            assert time.monotonic() < t_end, "the ranks never started"
            time.sleep(0.05)
        pids = [int((tmp_path / f"pid.{r}").read_text()) for r in range(2)]
        p.send_signal(signal.SIGTERM)
        rc = p.wait(timeout=30)
        assert rc == 128 + signal.SIGTERM, (rc, p.stderr.read().decode()[-500:])
        t_end = time.monotonic() + 10
        for pid in pids:
            while True:
                try:
                    os.kill(pid, 0)
                except ProcessLookupError:
                    break
                assert time.monotonic() < t_end, f"rank process {pid} outlived its spawner"
                time.sleep(0.05)
    finally:
        if p.poll() is None:
            p.kill()
