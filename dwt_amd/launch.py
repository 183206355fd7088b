"""One process per GPU, started by a parent that never touches the GPU.

`python bench.py --gpus N` (N > 1, no RANK in the environment) lands here: the parent starts N
children of the same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set
(the variables `torch.distributed.run` would set), waits for them and returns the first non-zero
exit status.  It imports neither torch nor the HIP library, so no process that has initialised the
GPU is ever replaced or forked.  Under `torch.distributed.run` (RANK already set) nothing is spawned.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def needs_spawn(gpus, env=None):
    """True when this process is the GPU-free parent of an N>1 run."""
    env = os.environ if env is None else env
    return gpus > 1 and "RANK" not in env and "WORLD_SIZE" not in env


def check_world(gpus, env=None):
    """Under a launcher WORLD_SIZE must equal --gpus: a mismatch is an error, not a warning."""
    env = os.environ if env is None else env
    world = int(env.get("WORLD_SIZE", "1"))
    if world != gpus:
        raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}: start {gpus} ranks (or run without a launcher, "
                         "bench.py then starts them itself)")
    return world


def child_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({
        "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
        "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
    })
    return env


def spawn_ranks(script, argv, world, poll_s=0.2, python=None):
    """Run `python script *argv` once per rank; returns the exit status for the parent (0 iff all ranks
    returned 0).  If one rank fails the others are terminated (their own process groups only)."""
    assert "torch" not in sys.modules or not _cuda_initialised(), "the parent of the ranks must not have initialised the GPU"
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([python or sys.executable, script, *argv], env=child_env(r, world, port),
                                      start_new_session=True))
    status = 0
    live = set(range(world))
    try:
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 128 - rc
                    print(f"rank {r} exited with {rc}; stopping the other ranks", file=sys.stderr)
                    for q in sorted(live):
                        _stop(procs[q])
            if live:
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                _stop(p)
    return status


def _stop(p, grace_s=10.0):
    try:
        os.killpg(p.pid, signal.SIGTERM)
    except (ProcessLookupError, PermissionError):
        return
    t0 = time.time()
    while p.poll() is None and time.time() - t0 < grace_s:
        time.sleep(0.1)
    if p.poll() is None:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass


def _cuda_initialised():
    torch = sys.modules.get("torch")
    try:
        return bool(torch and torch.cuda.is_initialized())
    except Exception:
        return False
