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


def _rows(rank, step, n, stride, kind="plain"):
    g = torch.Generator().manual_seed(100 + rank + 17 * step)
    if kind == "plain":
        lens = torch.randint(1, stride - 8, (n,), generator=g, dtype=torch.int64)
    else:   # uneven: a third empty streams, a few that fill the stride, the rest short, and one rank mostly empty
        lens = torch.randint(0, 40, (n,), generator=g, dtype=torch.int64)
        pick = torch.randint(0, 6, (n,), generator=g)
        lens[pick <= 1] = 0
        lens[pick == 5] = stride - int(torch.randint(0, 9, (1,), generator=g))
        if (rank + step) % 3 == 0:
            lens[: n - 2] = 0
    streams = torch.zeros((n, stride), dtype=torch.uint8)
    for i in range(n):
        streams[i, : int(lens[i])] = (torch.arange(int(lens[i])) * (rank + 3) + i + step).to(torch.uint8)
    return streams, lens


def _check(got, world, step, n, stride, kind="plain"):
    ok = True
    for r in range(world):
        s2, l2 = _rows(r, step, n, stride, kind)
        ok &= bool((got.lens[r * n:(r + 1) * n] == l2).all())
        for i in range(n):
            ok &= bool(torch.equal(got.stream(r, i), s2[i, : int(l2[i])]))
    return ok


def _worker(rank, world, port, q):
    from dwt_amd.dist import StreamGather, gather_streams, shard_frames

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard_frames(7, rank, world)
    n, stride = 4, 64
    ok = True
    for mode in ("packed", "rows"):
        streams, lens = _rows(rank, 0, n, stride)
        got = gather_streams(streams, lens, dst=0, mode=mode)
        ok &= _check(got, world, 0, n, stride) if rank == 0 else got.bufs is None

        # pipelined: step k posts its lengths, the streams of step k-1 travel meanwhile, two slots
        g = StreamGather(n, "cpu", dst=0, slots=2, mode=mode)
        steps = 5
        slot_streams = [None, None]
        for k in range(steps):
            g.wait(k - 2)
            slot_streams[k % 2], lk = _rows(rank, k, n, stride)
            g.post(k, slot_streams[k % 2], lk)
            if k >= 1:
                g.collect(k - 1)
                g.collect(k - 1)   # a second call is a no-op
                got = g.result(k - 1)
                if rank == 0:
                    ok &= _check(got, world, k - 1, n, stride)
        g.collect(steps - 1)
        got = g.result(steps - 1)
        if rank == 0:
            ok &= _check(got, world, steps - 1, n, stride)
            ok &= g.bytes_gathered > 0
            # one message per peer and step when packed; one per frame and peer otherwise
            ok &= g.messages_posted == (steps * (world - 1) if mode == "packed" else steps * (world - 1) * n)
    q.put((rank, first, count, ok))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(target, world=2, timeout=180):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = launch.free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=timeout) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_shard_and_gather_world2():
    res = _run_world(_worker)
    assert res[0][1:3] == (0, 4) and res[1][1:3] == (4, 3)
    assert all(r[3] for r in res)


def _worker_uneven(rank, world, port, q):
    """Empty streams, streams that fill the stride, a rank that sends (almost) nothing, 160 frames per step, both slots
    used over and over with lengths that shrink and grow: what the receive slots held before must never show."""
    from dwt_amd.dist import StreamGather

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, stride = 160, 96
    ok = True
    for mode in ("packed", "rows"):
        g = StreamGather(n, "cpu", dst=0, slots=2, mode=mode)
        keep = [None, None]
        steps = 7
        for k in range(steps):
            g.wait(k - 2)
            keep[k % 2], lk = _rows(rank, k, n, stride, "uneven")
            g.post(k, keep[k % 2], lk)
            if k >= 1:
                g.collect(k - 1)
                got = g.result(k - 1)
                if rank == 0:
                    ok &= _check(got, world, k - 1, n, stride, "uneven")
        g.collect(steps - 1)
        got = g.result(steps - 1)
        if rank == 0:
            ok &= _check(got, world, steps - 1, n, stride, "uneven")
            if mode == "packed":   # no byte travels that is not stream (rounded to 8)
                for r in range(world):
                    ok &= got.offsets[r][n] == sum((int(v) + 7) // 8 * 8 for v in got.lens[r * n:(r + 1) * n])
    # a step in which nobody has anything to send
    g = StreamGather(3, "cpu", dst=0, slots=1)
    g.post(0, torch.zeros((3, 16), dtype=torch.uint8), torch.zeros(3, dtype=torch.int64))
    g.collect(0)
    got = g.result(0)
    ok &= int(got.lens.sum()) == 0 and (rank != 0 or got.stream(1, 2).numel() == 0)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_with_empty_uneven_and_many_streams_world2():
    assert all(r[1] for r in _run_world(_worker_uneven))


def test_packed_offsets_and_torch_pack_agree():
    from dwt_amd.dist import packed_offsets, torch_pack

    lens = [0, 1, 8, 9, 64, 70, 0, 3]
    off = packed_offsets(lens, 64)
    assert off == [0, 0, 8, 16, 32, 96, 160, 160, 168]     # 70 is clamped to the stride of 64
    streams = torch.arange(8 * 64, dtype=torch.int64).remainder(251).to(torch.uint8).view(8, 64)
    out = torch_pack(streams, lens, torch.full((off[-1] + 8,), 255, dtype=torch.uint8))
    for i, v in enumerate(lens):
        v = min(v, 64)
        assert torch.equal(out[off[i]:off[i] + v], streams[i, :v])
    assert int(out[off[-1]]) == 255


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
