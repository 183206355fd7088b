#!/usr/bin/env python3
"""bench.py — Mpixels/s of the lossless encode+decode round trip on MI355X.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic frames that
are already resident in HBM: pixels -> (YCoCg) -> forward CDF 5/3 -> Hilbert
linearisation -> bit-plane/RLE/VLI packer -> .dwt streams, then streams ->
token walk -> plane scatter -> reconstruction -> inverse CDF 5/3 -> pixels.
With N > 1 (one process per GPU, launched by torch.distributed.run) every rank
runs the same per-GPU batch (weak scaling; frames are independent)
and the encoded streams are gathered to rank 0 over RCCL inside the timed step.

Prints ONE JSON line on rank 0.  The workload at N=1 is BASELINE.json configs[1]
(4096x4096 8-bit gray, lossless).  `roofline` is measured live with HIP events
around the forward+inverse lifting kernels on the same frames; `cpu_baseline`
times the real reference binaries (oracle/_ref, built from /root/reference in
the build container) on a bounded sample of the same frames.
"""
import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (W, H, C, default frames per GPU per step)
    "gray4096": (4096, 4096, 1, 64),
    "rgb1080p": (1920, 1080, 3, 32),
    "rgb4096": (4096, 4096, 3, 4),
}
CONFIG_OF = {
    "gray4096": "BASELINE.json configs[1] geometry",
    "rgb1080p": "BASELINE.json configs[2] geometry",
    "rgb4096": "BASELINE.json configs[4] geometry",
}
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LIFT_BYTES_PER_SAMPLE = 16      # int32 read + write, forward and inverse


def cpu_baseline(W, H, C, frames, first_frames_pix):
    """Time the reference binaries (kind 'reference') or, without them, the oracle port, single thread."""
    ref_enc = os.path.join(ROOT, "oracle", "_ref", "encode")
    ref_dec = os.path.join(ROOT, "oracle", "_ref", "decode")
    port = os.path.join(ROOT, "oracle", "orc_cli")
    kind = "reference" if os.path.exists(ref_enc) and os.path.exists(ref_dec) else "port"
    te = td = 0.0
    ok = True
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(frames):
            src, dwt, dec = (os.path.join(tmp, n) for n in ("i.pnm", "o.dwt", "o.pnm"))
            with open(src, "wb") as f:
                f.write(b"P%d %d %d 255\n" % (5 if C == 1 else 6, W, H))
                f.write(first_frames_pix[i].tobytes())
            enc_cmd = [ref_enc, src, dwt] if kind == "reference" else [port, "encode", src, dwt]
            dec_cmd = [ref_dec, dwt, dec] if kind == "reference" else [port, "decode", dwt, dec]
            t0 = time.perf_counter()
            subprocess.run(enc_cmd, check=True, capture_output=True)
            t1 = time.perf_counter()
            subprocess.run(dec_cmd, check=True, capture_output=True)
            t2 = time.perf_counter()
            te += t1 - t0
            td += t2 - t1
            back = open(dec, "rb").read()
            ok = ok and back[back.index(b"\n") + 1:] == first_frames_pix[i].tobytes()
    return {
        "value": round(frames * W * H / (te + td) / 1e6, 4),
        "unit": "Mpixels/s",
        "cores": 1,
        "host_cores": os.cpu_count(),
        "kind": kind,
        "sample": f"{frames} of the benchmark's {W}x{H}x{C} frames, encode+decode CLI round trip incl. PNM file I/O, "
                  f"1 thread ({te:.2f}s encode + {td:.2f}s decode), lossless={ok}",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="gray4096", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (default per workload)")
    ap.add_argument("--cpu-frames", type=int, default=5, help="frames timed on the CPU reference (0 = skip)")
    ap.add_argument("--lift-reps", type=int, default=20)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: all ranks share cuda:0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import dwt_amd
    from dwt_amd.dist import gather_streams

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if args.one_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local)
    ctx = dwt_amd.Context(local)

    W, H, C, B = WORKLOADS[args.workload]
    if args.frames > 0:
        B = args.frames
    pix = ctx.synth_pixels(B, H, W, C, seed0=rank * B, kind=0)      # resident in HBM before the timed region
    stride = ctx.lib.dwtx_encode_bound(W, H, C)
    out = torch.empty((B, stride), dtype=torch.uint8, device=dev)
    info = torch.empty((B, ctypes.sizeof(dwt_amd.StreamInfo)), dtype=torch.uint8, device=dev)
    dec = torch.empty((B, W * H * C), dtype=torch.uint8, device=dev)
    gathered = {}

    def step():
        streams, inf = ctx.encode_device(pix, out=out, info=info)
        lens = ctx.stream_lengths(inf)
        work = None
        if world > 1:
            # the one exchange step of the path: lengths, then the streams, to rank 0 over RCCL/xGMI;
            # the transfer overlaps this rank's own decode
            gathered["streams"], gathered["lens"], work, gathered["send"] = gather_streams(streams, lens, dst=0,
                                                                                           async_op=True)
        d, dinfos = ctx.decode_device(streams, lens, W, H, C, out=dec)
        if work is not None:
            work.wait()
        return streams, lens, d, dinfos

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        streams, lens, d, dinfos = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- verification (outside the timed region) ---------------------------------
    lossless = bool(torch.equal(d.view(B, H, W, C), pix)) and all(i.status == 0 and not i.truncated for i in dinfos)
    lens_host = lens.cpu().tolist()
    golden_ok = None
    if rank == 0:
        gpath = os.path.join(ROOT, "tests", "golden", "golden.json")
        gname = {"gray4096": "g4096x4096", "rgb1080p": "c1920x1080", "rgb4096": "c4096x4096"}.get(args.workload)
        if gname and os.path.exists(gpath):
            rec = json.load(open(gpath))[gname]
            s0 = streams[0, : lens_host[0]].cpu().numpy().tobytes()
            golden_ok = len(s0) == rec["dwt_len"] and hashlib.sha256(s0).hexdigest() == rec["dwt_sha256"]
    if world > 1:
        flag = torch.tensor([1 if lossless else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        lossless = bool(flag.item())

    # ---- roofline of the lifting kernels, HIP events on the kernels' stream --------
    planes = ctx.planes_from_pixels(pix)
    pyr = torch.empty_like(planes)
    back = torch.empty_like(planes)
    ctx.transformation_fwd(planes, pyr)
    ctx.transformation_inv(pyr, back)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.lift_reps):
        ctx.transformation_fwd(planes, pyr)
        ctx.transformation_inv(pyr, back)
    e1.record()
    torch.cuda.synchronize()
    lift_ms = e0.elapsed_time(e1) / args.lift_reps
    lift_ok = bool(torch.equal(back, planes))
    samples = B * W * H * C
    achieved = LIFT_BYTES_PER_SAMPLE * samples / (lift_ms * 1e-3) / 1e9

    # ---- the entropy stage on its own (4 B coefficient + stream bytes per sample and direction) --
    lin = ctx.linearization(pyr)
    del back
    cstreams = torch.empty((B, stride), dtype=torch.uint8, device=dev)
    cinfo = torch.empty((B, ctypes.sizeof(dwt_amd.StreamInfo)), dtype=torch.uint8, device=dev)
    hinfo = (dwt_amd.DecodeInfo * B)()

    def coder_enc():
        rc = ctx.lib.dwtx_encode_planes(ctx.h, lin.data_ptr(), W, H, C, B, 0, cstreams.data_ptr(), stride, cinfo.data_ptr())
        assert rc == 0, rc

    coder_enc()
    clens = ctx.stream_lengths(cinfo)
    lin_out = torch.empty_like(lin)

    def coder_dec():
        rc = ctx.lib.dwtx_decode_planes(ctx.h, lin_out.data_ptr(), cstreams.data_ptr(), stride, clens.data_ptr(), W, H, C, B, -1,
                                        ctypes.cast(hinfo, ctypes.c_void_p))
        assert rc == 0, rc

    coder_dec()
    torch.cuda.synchronize()
    c0, c1, c2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    c0.record()
    coder_enc()
    c1.record()
    coder_dec()
    c2.record()
    torch.cuda.synchronize()
    coder_ok = bool(torch.equal(lin_out, lin))
    coder_bytes = 4 * samples + int(clens.sum().item())
    coder = {"what": "dwtx_encode_planes / dwtx_decode_planes alone on the same frames (linearised coefficients <-> streams)",
             "algorithmic_bytes_per_step": coder_bytes, "coefficients_roundtrip": coder_ok}
    for name, ms in (("encode", c0.elapsed_time(c1)), ("decode", c1.elapsed_time(c2))):
        coder[name] = {"ms_per_step": round(ms, 3), "achieved_GBs": round(coder_bytes / (ms * 1e-3) / 1e9, 1),
                       "frac_of_hbm_peak": round(coder_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del lin, lin_out, cstreams

    # ---- stage breakdown (one extra untimed pass with events) ----------------------
    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b), r

    enc_ms, _ = timed(lambda: ctx.encode_device(pix, out=out, info=info))
    dec_ms, _ = timed(lambda: ctx.decode_device(streams, lens, W, H, C, out=dec))

    # HBM bytes of the same kernels from rocprofv3 PMC counters (FETCH_SIZE doubled, WRITE_SIZE), collected
    # in a separate profiling run (profiles/): the counters cannot be read from inside this process
    traffic = traffic_detail = None
    tpath = os.path.join(ROOT, "profiles", "r01_lift_traffic_pmc.json")
    if os.path.exists(tpath):
        per_sample = json.load(open(tpath))["traffic_bytes_per_sample"]
        traffic = int(per_sample * samples)   # HBM bytes of one forward+inverse pass over the step's planes, like `achieved`
        traffic_detail = {"bytes_per_sample": round(per_sample, 2), "algorithmic_bytes_per_sample": LIFT_BYTES_PER_SAMPLE,
                          "source": "profiles/r01_lift_traffic_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes "
                                    "over the same kernels, FETCH_SIZE doubled for gfx950; tools/pmc_lift.sh)"}
    if rank == 0:
        total_px = world * B * W * H * args.steps
        result = {
            "metric": "Mpixels/s lossless encode+decode round-trip",
            "value": round(total_px / elapsed / 1e6, 3),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"{W}x{H}x{C} 8-bit synthetic smooth+noise frames, lossless, {B} frames per GPU per step "
                            f"({CONFIG_OF[args.workload]})",
                "frames_per_gpu": B,
                "parallelism": f"frames sharded over {world} GPU(s), RCCL gather of streams" if world > 1 else "1 GPU",
            },
            "bit_exact": {"roundtrip_lossless": lossless, "stream0_matches_reference_golden": golden_ok,
                          "lifting_roundtrip": lift_ok},
            "bytes_per_frame": int(sum(lens_host) / len(lens_host)),
            "stage_ms_per_step": {"encode": round(enc_ms, 3), "decode": round(dec_ms, 3)},
            "coder": coder,
            "roofline": {
                "kernel": "k_fwd_level + k_inv_level (all levels, forward+inverse CDF 5/3)",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_detail": traffic_detail,
                "algorithmic_bytes": LIFT_BYTES_PER_SAMPLE * samples,
                "bytes_per_sample": LIFT_BYTES_PER_SAMPLE,
                "us_per_frame": round(lift_ms * 1e3 / B, 2),
            },
        }
        if world == 1 and args.cpu_frames > 0:
            sample = pix[: min(args.cpu_frames, B)].cpu().numpy()
            result["cpu_baseline"] = cpu_baseline(W, H, C, sample.shape[0], sample)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
