"""CPU (gloo, world_size 2) test of the N>1 host logic: frame sharding and the gather of
variable-length streams to rank 0.  No GPU, no codec calls."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    from dwt_amd.dist import gather_streams, shard_frames

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 7
    first, count = shard_frames(total, rank, world)
    n, stride = 4, 64
    g = torch.Generator().manual_seed(100 + rank)
    lens = torch.randint(1, stride - 8, (n,), generator=g, dtype=torch.int64)
    streams = torch.zeros((n, stride), dtype=torch.uint8)
    for i in range(n):
        streams[i, : int(lens[i])] = (torch.arange(int(lens[i])) * (rank + 3) + i).to(torch.uint8)
    bufs, all_lens = gather_streams(streams, lens, dst=0)
    ok = True
    if rank == 0:
        for r in range(world):
            g2 = torch.Generator().manual_seed(100 + r)
            l2 = torch.randint(1, stride - 8, (n,), generator=g2, dtype=torch.int64)
            ok &= bool((all_lens[r * n:(r + 1) * n] == l2).all())
            for i in range(n):
                want = (torch.arange(int(l2[i])) * (r + 3) + i).to(torch.uint8)
                ok &= bool((bufs[r][i, : int(l2[i])] == want).all())
    else:
        ok = bufs is None
    q.put((rank, first, count, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
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
