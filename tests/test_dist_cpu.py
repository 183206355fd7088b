"""CPU tests of the N>1 host logic (no GPU, no codec calls): frame sharding, the gather of
variable-length streams to rank 0 over gloo (world_size 2) — one-shot and pipelined one step
behind the producer as bench.py runs it — and the launcher that starts one process per GPU."""
import os
import subprocess
import sys
import textwrap

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dwt_amd import launch  # noqa: E402


def _rows(rank, step, n, stride):
    g = torch.Generator().manual_seed(100 + rank + 17 * step)
    lens = torch.randint(1, stride - 8, (n,), generator=g, dtype=torch.int64)
    streams = torch.zeros((n, stride), dtype=torch.uint8)
    for i in range(n):
        streams[i, : int(lens[i])] = (torch.arange(int(lens[i])) * (rank + 3) + i + step).to(torch.uint8)
    return streams, lens


def _check(bufs, all_lens, world, step, n, stride):
    ok = True
    for r in range(world):
        s2, l2 = _rows(r, step, n, stride)
        ok &= bool((all_lens[r * n:(r + 1) * n] == l2).all())
        for i in range(n):
            ok &= bool((bufs[r][i, : int(l2[i])] == s2[i, : int(l2[i])]).all())
    return ok


def _worker(rank, world, port, q):
    from dwt_amd.dist import StreamGather, gather_streams, shard_frames

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard_frames(7, rank, world)
    n, stride = 4, 64
    streams, lens = _rows(rank, 0, n, stride)
    bufs, all_lens = gather_streams(streams, lens, dst=0)
    ok = _check(bufs, all_lens, world, 0, n, stride) if rank == 0 else bufs is None

    # pipelined: step k posts its lengths, the streams of step k-1 travel meanwhile, two slots
    g = StreamGather(n, "cpu", dst=0, slots=2)
    steps = 5
    slot_streams = [None, None]
    for k in range(steps):
        g.wait(k - 2)
        slot_streams[k % 2], lk = _rows(rank, k, n, stride)
        g.post(k, slot_streams[k % 2], lk)
        if k >= 1:
            g.collect(k - 1)
            g.collect(k - 1)   # a second call is a no-op
            bufs, al = g.result(k - 1)
            if rank == 0:
                ok &= _check(bufs, al, world, k - 1, n, stride)
    g.collect(steps - 1)
    bufs, al = g.result(steps - 1)
    if rank == 0:
        ok &= _check(bufs, al, world, steps - 1, n, stride)
        ok &= g.bytes_gathered > 0
    q.put((rank, first, count, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = launch.free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:3] == (0, 4) and res[1][1:3] == (4, 3)
    assert all(r[3] for r in res)


def test_shard_frames_covers_everything():
    from dwt_amd.dist import shard_frames

    for total in (1, 7, 8, 1024, 8192):
        for world in (1, 2, 3, 8):
            spans = [shard_frames(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (a, ca), (b, _) in zip(spans, spans[1:]):
                assert a + ca == b


# ---- the launcher behind `python bench.py --gpus N` -------------------------------

def test_needs_spawn_and_world_check():
    assert launch.needs_spawn(2, env={})
    assert not launch.needs_spawn(1, env={})
    assert not launch.needs_spawn(2, env={"RANK": "0", "WORLD_SIZE": "2"})
    assert launch.check_world(2, env={"WORLD_SIZE": "2"}) == 2
    assert launch.check_world(1, env={}) == 1
    for gpus, env in ((8, {"WORLD_SIZE": "1"}), (8, {}), (1, {"WORLD_SIZE": "2"})):
        try:
            launch.check_world(gpus, env=env)
        except SystemExit as e:
            assert "WORLD_SIZE" in str(e)
        else:
            raise AssertionError("a world size that differs from --gpus must be an error")


def _script(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_spawn_ranks_sets_the_rendezvous_environment(tmp_path):
    out = tmp_path / "out"
    out.mkdir()
    script = _script(tmp_path, f"""
        import os, sys
        keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"]
        open(os.path.join({str(out)!r}, os.environ["RANK"]), "w").write(" ".join(os.environ[k] for k in keys) + " " + " ".join(sys.argv[1:]))
    """)
    assert launch.spawn_ranks(script, ["--gpus", "3"], 3) == 0
    seen = sorted(open(out / f).read().split() for f in os.listdir(out))
    assert [s[0] for s in seen] == ["0", "1", "2"] and [s[1] for s in seen] == ["0", "1", "2"]
    assert all(s[2] == "3" and s[3] == "127.0.0.1" and s[5:] == ["--gpus", "3"] for s in seen)
    assert len({s[4] for s in seen}) == 1


def test_spawn_ranks_propagates_a_failing_rank(tmp_path):
    script = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)      # the surviving rank is stopped by the parent, not waited for
    """)
    import time
    t0 = time.time()
    assert launch.spawn_ranks(script, [], 2) == 7
    assert time.time() - t0 < 30


def test_bench_parent_stays_off_the_gpu_and_reports_failure():
    """Here (no GPU) the ranks of `bench.py --gpus 2` cannot run: the parent must say so with a
    non-zero status, and must get there without importing torch (it may never initialise the GPU)."""
    code = textwrap.dedent(f"""
        import runpy, sys
        sys.argv = ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-frames", "0"]
        try:
            runpy.run_path({os.path.join(ROOT, "bench.py")!r}, run_name="__main__")
        except SystemExit as e:
            print("PARENT_TORCH", "torch" in sys.modules, "STATUS", e.code)
    """)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("PARENT_TORCH")]
    assert line, r.stdout + r.stderr
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert line[0].split()[1] == "False"
    else:
        assert line[0].split() == ["PARENT_TORCH", "False", "STATUS"] + [line[0].split()[3]] and line[0].split()[3] != "0"


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
