"""Spawner of the ranks of one single-node data-parallel job: `python bench.py --gpus N` with no launcher around it
becomes N child processes of the same command line, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT in their environment, as `python -m torch.distributed.run` would set them).

Standard library only, and deliberately so: the parent never loads libdfhip.so, never makes a HIP call and never
exec's -- it only starts children, forwards rank 0's standard output and reports how the ranks ended.  (The
reference has no distributed code at all; the contract is this repository's own)

    rc, last_json = spawn_ranks(4, [sys.executable, "bench.py", "--gpus", "4"])

Rank 0's stdout is streamed through line by line, except that the LAST line that parses as a JSON object is held back
and printed after every rank has exited, so that it is the last line of the parent's stdout whatever a library's
exit handler writes.  The other ranks' stdout goes to the parent's stderr with a `[rank k]` prefix.  Any rank that
exits non-zero makes the job fail with that code; the surviving ranks (which would wait in a collective for ever) get
`grace` seconds and are then terminated -- by PID, never by pattern.
"""
import json
import os
import socket
import signal
import subprocess
import sys
import threading
import time

STORE_PORT_OFFSET = 23          # rendezvous.py serves the ncclUniqueId on MASTER_PORT + this


def free_port_pair(offset=STORE_PORT_OFFSET, host="127.0.0.1", tries=64):
    """A port p such that p and p + offset are both free right now (MASTER_PORT and the rendezvous port)."""
    for _ in range(tries):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as a:
            a.bind((host, 0))
            p = a.getsockname()[1]
            if p + offset > 65535:
                continue
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as b:
                try:
                    b.bind((host, p + offset))
                except OSError:
                    continue
            return p
    raise RuntimeError("no free (port, port + %d) pair on %s" % (offset, host))


def rank_environment(rank, world, port, base=None, run_id=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                "DF_RUN_ID": run_id or f"{os.getpid()}-{port}",
                # the host driver only supports dmabuf IPC: RCCL's peer mappings fail without it
                "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    return env


def _is_json_object(line):
    line = line.strip()
    if not (line.startswith("{") and line.endswith("}")):
        return False
    try:
        return isinstance(json.loads(line), dict)
    except ValueError:
        return False


class _Terminated(Exception):
    """SIGTERM reached the spawner"""


def spawn_ranks(world, command, timeout=None, grace=20.0, out=None, err=None, env=None):
    """Run `command` (argv list) as `world` ranks.  Returns (exit code, last JSON-object line of rank 0 or None).
    The JSON line is written to `out` last; exit code = 0 only if every rank exited 0."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port_pair()
    procs, threads = [], []
    held = {"json": None}
    lock = threading.Lock()

    def pump_rank0(stream):
        for line in stream:
            with lock:
                if _is_json_object(line):
                    if held["json"] is not None:
                        out.write(held["json"])          # an earlier JSON-looking line was not the last one
                    held["json"] = line if line.endswith("\n") else line + "\n"
                else:
                    out.write(line)
                out.flush()

    def pump_other(rank, stream):
        for line in stream:
            with lock:
                err.write(f"[rank {rank}] {line}")
                err.flush()

    def reap():
        """terminate, then kill, every rank that is still alive (by PID; a child is a fresh process: nothing is re-exec'ed)"""
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            try:
                p.terminate()
            except OSError:
                pass
        t_end = time.monotonic() + 5.0
        for p in alive:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                try:
                    p.kill()
                except OSError:
                    pass
        for p in alive:
            try:
                p.wait(timeout=5.0)
            except subprocess.TimeoutExpired:
                pass

    # SIGTERM to the spawner (a job timeout) must not leave ranks behind that wait in a collective forever and hold their
    # GPUs.  The handler only sets a flag: the poll loop below sees it within 50 ms and unwinds through `finally` (an
    # exception raised from the handler could land inside Popen -- a child that is not in `procs` yet -- or inside reap()).
    term = {"signum": None}

    def on_term(signum, frame):
        term["signum"] = signum

    _unset = object()
    old_term = _unset
    if threading.current_thread() is threading.main_thread():
        try:
            old_term = signal.signal(signal.SIGTERM, on_term)
        except (ValueError, OSError):
            old_term = _unset
    code = 0
    try:
        for r in range(world):
            if term["signum"] is not None:
                raise _Terminated(term["signum"])
            p = subprocess.Popen(command, env=rank_environment(r, world, port, base=env), stdout=subprocess.PIPE,
                                 stderr=None, text=True, bufsize=1)
            procs.append(p)
            t = threading.Thread(target=pump_rank0 if r == 0 else pump_other,
                                 args=(p.stdout,) if r == 0 else (r, p.stdout), daemon=True)
            t.start()
            threads.append(t)

        deadline = None if timeout is None else time.monotonic() + timeout
        failed_at = None
        while True:
            if term["signum"] is not None:
                raise _Terminated(term["signum"])
            states = [p.poll() for p in procs]
            bad = [(r, s) for r, s in enumerate(states) if s not in (None, 0)]
            if bad and failed_at is None:
                failed_at = time.monotonic()
                code = bad[0][1] if bad[0][1] > 0 else 128 - bad[0][1]      # a signal -n reads as 128 + n
                err.write("[launch] rank %d exited with %d; the other ranks get %.0f s\n" % (bad[0][0], bad[0][1], grace))
                err.flush()
            if all(s is not None for s in states):
                break
            now = time.monotonic()
            if (failed_at is not None and now - failed_at > grace) or (deadline is not None and now > deadline):
                if failed_at is None:
                    code = 124
                    err.write("[launch] timeout: terminating the ranks\n")
                reap()
                break
            time.sleep(0.05)
    except _Terminated as e:
        code = 128 + int(e.args[0])
        err.write("[launch] signal %d: terminating the ranks\n" % int(e.args[0]))
    finally:
        # (a failed Popen of a later rank, KeyboardInterrupt, SIGTERM, any exception: no rank outlives the spawner)
        reap()                   # (a second SIGTERM while this runs only sets the flag again: the reap completes)
        if old_term is not _unset:
            try:
                # (None = the previous handler was installed from C: SIG_DFL is the nearest thing Python can put back)
                signal.signal(signal.SIGTERM, old_term if old_term is not None else signal.SIG_DFL)
            except (ValueError, OSError):
                pass
    for p in procs:
        p.wait()
    for t in threads:
        t.join(timeout=5.0)
    if code == 0:
        bad = [p.returncode for p in procs if p.returncode != 0]
        if bad:
            code = bad[0] if bad[0] > 0 else 128 - bad[0]
    with lock:
        if held["json"] is not None:
            out.write(held["json"])
        out.flush()
    return code, (held["json"].strip() if held["json"] else None)
