"""`python bench.py --gpus 2` on the GPU box: the parent starts two fresh rank processes (here both on
cuda:0 over gloo — RCCL refuses two ranks on one device; the collective calls are the same), every rank
encodes and decodes its own frames on the GPU, the streams of every step are gathered to rank 0.  The
gathered streams of BOTH ranks must be the oracle's bytes and the line must say n_gpus 2, lossless."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("geom,frames,mode", [((256, 192, 3), 3, "packed"), ((331, 277, 1), 2, "rows"), ((331, 277, 1), 5, "packed")])
def test_two_ranks_encode_gather_decode(tmp_path, geom, frames, mode):
    W, H, C = geom
    dump = str(tmp_path / "gathered.npz")
    cmd = [sys.executable, os.path.join(orc.ROOT, "bench.py"), "--gpus", "2", "--one-device", "--backend", "gloo",
           "--steps", "3", "--warmup", "1", "--frames", str(frames), "--geometry", str(W), str(H), str(C),
           "--cpu-frames", "0", "--lift-reps", "1", "--dump-gathered", dump, "--gather", mode]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["frames_per_gpu"] == frames
    assert line["bit_exact"]["roundtrip_lossless"] is True          # MIN over both ranks' decoded pixels == inputs
    assert line["gathered"]["world"] == 2 and line["gathered"]["own_rows_match"] is True and line["gathered"]["mode"] == mode
    assert line["gathered"]["p2p_operations_per_step_on_rank0"] == (1 if mode == "packed" else frames)
    z = np.load(dump)
    lens = z["lens"]
    assert lens.shape == (2 * frames,)
    for rank in range(2):
        rows = z[f"rank{rank}"]
        for i in range(frames):
            want, _ = orc.encode(orc.synth(W, H, C, rank * frames + i, 0))   # bench.py: seed0 = rank * frames
            n = int(lens[rank * frames + i])
            assert n == len(want) and rows[i, :n].tobytes() == want, f"rank {rank} frame {i}"
    assert line["gathered"]["bytes_last_step"] == int(lens.sum())


def test_one_rank_under_a_launcher_uses_rccl(tmp_path):
    """WORLD_SIZE=1 through torch.distributed.run-style variables: world 1 takes the plain path, and a
    world size that disagrees with --gpus is refused."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, os.path.join(orc.ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--frames", "2",
           "--geometry", "256", "256", "1", "--cpu-frames", "0", "--lift-reps", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["bit_exact"]["roundtrip_lossless"] is True
