#!/usr/bin/env python3
"""bench.py — Mpixels/s of the lossless encode+decode round trip on MI355X.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic frames that
are already resident in HBM: pixels -> (YCoCg) -> forward CDF 5/3 -> Hilbert
linearisation -> bit-plane/RLE/VLI packer -> .dwt streams, then streams ->
token walk -> plane scatter -> reconstruction -> inverse CDF 5/3 -> pixels.

N > 1: one process per GPU.  Started without a launcher (`python bench.py --gpus N`)
the parent — which never touches the GPU — starts the N ranks itself (dwt_amd/launch.py);
under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks
come from the launcher.  Every rank runs the same per-GPU batch on its own frames (weak
scaling; frames are independent); the encoded streams of every step are
gathered to rank 0 over RCCL inside the timed region, one step behind the encoder.

Prints ONE JSON line on rank 0.  The workload at N=1 is BASELINE.json configs[1]
(4096x4096 8-bit gray, lossless).  `roofline` is measured live with HIP events
around the forward+inverse lifting kernels on the same frames; `cpu_baseline`
times the real reference binaries (oracle/_ref, built from /root/reference in
the build container) on a bounded sample of the same frames and the GPU's streams
of those frames are compared with the reference's bytes.
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
    "gray4096": (4096, 4096, 1, 128),   # 64 frames until round 3; a part of the codec's four-part pipeline fills the chip better with 32 frames than with 16
    "rgb1080p": (1920, 1080, 3, 1024),  # configs[2]: 1024 frames per step, 6.4 G samples in one call each way
    "rgb4096": (4096, 4096, 3, 64),
}
CONFIG_OF = {
    "gray4096": "BASELINE.json configs[1] geometry",
    "rgb1080p": "BASELINE.json configs[2]",
    "rgb4096": "BASELINE.json configs[4] geometry",
}
GOLDEN_OF = {"gray4096": "g4096x4096", "rgb1080p": "c1920x1080", "rgb4096": "c4096x4096"}
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# wave64 vector instructions per second for the whole chip, MEASURED (tools/mb/mb_valu.hip -> profiles/r04_valu_peak.json, 8 waves
# per SIMD on all 256 CUs): gfx950 issues two classes of integer instructions.  Shifts left, bit-field extract / insert,
# v_alignbit, v_perm, every three-operand and every packed 16-bit instruction, DPP moves, v_cndmask, v_readlane, multiplies,
# min / max, carries: 4.1 cycles per wave64 instruction per SIMD = 597 G/s ("full" class: what the entropy stage is made of).
# v_add_u32 / v_sub_u32, and / or / xor / not, v_mov, shifts RIGHT: 2.4 cycles = 938 G/s ("fast" class).
VALU_PEAK_FILE = "r04_valu_peak.json"
VALU_FULL_CLASS_PEAK = 596.7e9
VALU_FAST_CLASS_PEAK = 938.1e9
KERNEL_RECORD_FRAMES = 64       # frames (planes x channels apart) per launch in the roofline / roofline_codec / coder records
LIFT_BYTES_PER_SAMPLE = 16      # int32 read + write, forward and inverse
LIFT_READ_BYTES_PER_SAMPLE = 8  # the read-only variant


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def sources_sha16(files=("dwt_amd/csrc/pack.hip", "dwt_amd/csrc/unpack.hip")):
    """what the kept instruction counts are stamped with (the GPU box has no .git): a hash of the entropy stage's sources"""
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def valu_peaks():
    """(full-class, fast-class) peak from the kept measurement, the constants above when the file is not there"""
    path = os.path.join(ROOT, "profiles", VALU_PEAK_FILE)
    if os.path.exists(path):
        j = json.load(open(path))
        return j["full_class_G_wave_insts_per_s"] * 1e9, j["fast_class_G_wave_insts_per_s"] * 1e9, f"profiles/{VALU_PEAK_FILE}"
    return VALU_FULL_CLASS_PEAK, VALU_FAST_CLASS_PEAK, "bench.py constants (profiles/" + VALU_PEAK_FILE + " missing)"


def cpu_baseline(W, H, C, frames, first_frames_pix, gpu_streams=None):
    """Time the reference binaries (kind 'reference') or, without them, the oracle port, single thread.
    gpu_streams[i] (bytes) is compared with the .dwt the CPU path writes for frame i."""
    ref_enc = os.path.join(ROOT, "oracle", "_ref", "encode")
    ref_dec = os.path.join(ROOT, "oracle", "_ref", "decode")
    port = os.path.join(ROOT, "oracle", "orc_cli")
    kind = "reference" if os.path.exists(ref_enc) and os.path.exists(ref_dec) else "port"
    te = td = 0.0
    ok = True
    same = 0
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
            if gpu_streams is not None and open(dwt, "rb").read() == gpu_streams[i]:
                same += 1
    return {
        "value": round(frames * W * H / (te + td) / 1e6, 4),
        "unit": "Mpixels/s",
        "cores": 1,
        "host_cores": os.cpu_count(),
        "kind": kind,
        "sample": f"{frames} of the benchmark's {W}x{H}x{C} frames, encode+decode CLI round trip incl. PNM file I/O, "
                  f"1 thread ({te:.2f}s encode + {td:.2f}s decode), lossless={ok}",
        "gpu_streams_equal_cpu_bytes": None if gpu_streams is None else f"{same}/{frames}",
    }, (same == frames if gpu_streams is not None else None)


class Runner:
    """The hot path over one workload's resident batch, with the per-step stream gather when world > 1."""

    def __init__(self, ctx, torch, dwt_amd, name, frames, rank, world, dev, gather_mode="packed"):
        self.ctx, self.torch, self.dwt_amd, self.name = ctx, torch, dwt_amd, name
        self.W, self.H, self.C, self.B = WORKLOADS[name]
        if frames > 0:
            self.B = frames
        W, H, C, B = self.W, self.H, self.C, self.B
        self.rank, self.world, self.dev = rank, world, dev
        self.pix = ctx.synth_pixels(B, H, W, C, seed0=rank * B, kind=0)      # resident in HBM before the timed region
        self.stride = ctx.lib.dwtx_encode_bound(W, H, C)
        if B * self.stride > (2 << 30):
            # The worst-case bound (3 bytes per sample) times the batch is gigabytes of output rows (tens of them for a thousand
            # frames), and the decoder's chunk tables are laid out per stride (45 GB for 128 gray frames at the bound, 11 GB
            # this way; the speed is the same either way, tools/time_codec.py TIGHT_STRIDE=1): rows sized from the streams of a
            # few probe frames, half as much again (a stream that did not fit would be clipped like a CAPACITY and fail the
            # lossless check below)
            probe, pinfo = ctx.encode_device(self.pix[:8])
            self.stride = (int(ctx.stream_lengths(pinfo).max().item()) * 3 // 2 + 64 + 7) // 8 * 8
            del probe, pinfo
            torch.cuda.empty_cache()
        slots = 2 if world > 1 else 1
        self.out = [torch.empty((B, self.stride), dtype=torch.uint8, device=dev) for _ in range(slots)]
        self.info = [torch.empty((B, ctypes.sizeof(dwt_amd.StreamInfo)), dtype=torch.uint8, device=dev) for _ in range(slots)]
        self.dec = torch.empty((B, W * H * C), dtype=torch.uint8, device=dev)
        self.gather = None
        if world > 1:
            from dwt_amd.dist import StreamGather
            # one message per peer and step: the rank's streams packed by dwtx_pack_streams ("rows": one send per frame, no copy)
            self.gather = StreamGather(B, dev, dst=0, slots=2, mode=gather_mode, packer=lambda st, ln, out: ctx.pack_streams(st, ln, out))
        self.k = 0

    def step(self):
        ctx, g, k = self.ctx, self.gather, self.k
        s = k % len(self.out)
        if g is not None:
            g.wait(k - 2)                      # the gather that last read this slot's streams is done
        streams, inf = ctx.encode_device(self.pix, out=self.out[s], info=self.info[s])
        lens = ctx.stream_lengths(inf)
        if g is not None:
            # the one exchange step of the path: lengths now, the streams one step later (their lengths
            # are on the host by then), to rank 0 over RCCL/xGMI, overlapping this rank's own work
            g.post(k, streams, lens)
            if k >= 1:
                g.collect(k - 1)
        d, dinfos = ctx.decode_device(streams, lens, self.W, self.H, self.C, out=self.dec)
        self.k += 1
        return streams, lens, d, dinfos

    def drain(self):
        """the last step's streams still have to reach rank 0 (inside the timed region)"""
        if self.gather is not None and self.k >= 1:
            self.gather.collect(self.k - 1)
            self.gather.wait(self.k - 2)
            self.gather.wait(self.k - 1)

    def timed(self, steps, warmup, fence):
        for _ in range(warmup):
            self.step()
        self.drain()
        fence()
        k0 = self.k
        t0 = time.perf_counter()
        for _ in range(steps):
            last = self.step()
        t1 = time.perf_counter()
        self.drain()
        self.torch.cuda.synchronize()
        self.own_s = time.perf_counter() - t0     # this rank's own steps and its part of the last gather, before the barrier
        self.drain_s = time.perf_counter() - t1   # of which: finishing the device work in flight and the last step's gather
        fence()
        return time.perf_counter() - t0, last, k0


def coder_record(ctx, torch, dwt_amd, lin, W, H, C, B, stride, dev, reps=2):
    """The entropy stage on its own (4 B coefficient + stream bytes per sample and direction)."""
    cstreams = torch.empty((B, stride), dtype=torch.uint8, device=dev)
    cinfo = torch.empty((B, ctypes.sizeof(dwt_amd.StreamInfo)), dtype=torch.uint8, device=dev)
    hinfo = (dwt_amd.DecodeInfo * B)()
    samples = B * W * H * C

    def enc():
        rc = ctx.lib.dwtx_encode_planes(ctx.h, lin.data_ptr(), W, H, C, B, 0, cstreams.data_ptr(), stride, cinfo.data_ptr())
        assert rc == 0, rc

    enc()
    clens = ctx.stream_lengths(cinfo)
    lin_out = torch.empty_like(lin)

    def dec():
        rc = ctx.lib.dwtx_decode_planes(ctx.h, lin_out.data_ptr(), cstreams.data_ptr(), stride, clens.data_ptr(), W, H, C, B, -1,
                                        ctypes.cast(hinfo, ctypes.c_void_p))
        assert rc == 0, rc

    dec()
    torch.cuda.synchronize()
    best = {"encode": 1e30, "decode": 1e30}
    for _ in range(reps):
        c0, c1, c2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        c0.record()
        enc()
        c1.record()
        dec()
        c2.record()
        torch.cuda.synchronize()
        best["encode"] = min(best["encode"], c0.elapsed_time(c1))
        best["decode"] = min(best["decode"], c1.elapsed_time(c2))
    ok = bool(torch.equal(lin_out, lin))
    nbytes = 4 * samples + int(clens.sum().item())
    rec = {"what": f"dwtx_encode_planes / dwtx_decode_planes alone on {B} of the same frames per call (linearised coefficients <-> streams)",
           "frames": B, "algorithmic_bytes_per_step": nbytes, "coefficients_roundtrip": ok}
    for name, ms in best.items():
        rec[name] = {"ms_per_step": round(ms, 3), "achieved_GBs": round(nbytes / (ms * 1e-3) / 1e9, 1),
                     "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    # The stage's instruction roofline.  The per-kernel instruction counts cannot be read from inside this process; they come
    # from the committed rocprofv3 --pmc run over the same kernels (tools/pmc_coder.sh -> profiles/*_coder_insts.json:
    # SQ_INSTS_VALU and SQ_WAVES per kernel for one encode + decode of 4096x4096 gray frames), scaled by the coefficient
    # count; the time is this run's; the peaks are measured (tools/mb/mb_valu.hip).  The counts are stamped with the
    # commit they were taken at: `stale` says whether the kernels' sources have changed since.
    ipath = next((os.path.join(ROOT, "profiles", f) for f in ("r04_coder_insts.json", "r03_coder_insts.json")
                  if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
    if ipath and (W, H, C) == (4096, 4096, 1):
        ij = json.load(open(ipath))
        full_peak, fast_peak, peak_src = valu_peaks()
        for name in ("encode", "decode"):
            per_coef = ij[name]["valu_wave_insts_per_coefficient"]     # wave64 instructions per coefficient (x 64 = lane operations)
            insts = per_coef * samples
            rate = insts / (rec[name]["ms_per_step"] * 1e-3)
            rec[name]["valu_insts_per_coefficient"] = round(per_coef * 64, 1)   # vector operations per coefficient (64 lanes per wave instruction)
            rec[name]["frac_of_valu_full_class_peak"] = round(rate / full_peak, 4)
            rec[name]["frac_of_valu_fast_class_peak"] = round(rate / fast_peak, 4)
        stamp = ij.get("sources_sha16")   # the counts are stamped with a hash of the kernels' sources (tools/pmc_coder.sh)
        stale = (stamp != sources_sha16()) if stamp else None
        rec["instruction_roofline"] = {"peak_full_class_wave_insts_per_s": full_peak, "peak_fast_class_wave_insts_per_s": fast_peak,
                                       "peak_source": peak_src, "counts_source": "profiles/" + os.path.basename(ipath),
                                       "counts_sources_sha16": stamp, "sources_sha16_now": sources_sha16(),
                                       "counts_stale": stale if stamp else "unknown (no source stamp in the file)",
                                       "note": "full class = shifts left, bit-field, permute, three-operand, packed 16-bit, DPP, select, lane reads "
                                               "(4.1 cycles per wave64 instruction per SIMD); fast class = add / sub / logic / shift right (2.4 cycles); "
                                               "the stage's kernels are mostly full class",
                                       "per_kernel": ij.get("per_kernel")}
    return rec


def capacity_record(dwt_amd, torch, local, timed):
    """BASELINE.json configs[3]: one 16384x16384 RGB frame with CAPACITY = 1 MiB (encode.c:150-152,192-217): encode with the
    capacity cut (pack.hip k_cut: segments that start beyond the capacity are never coded) and with the cut switched off
    (everything coded, the stream clipped at the end), decode of the 1 MiB stream, the reference's hashes."""
    W = H = 16384
    C, cap = 3, 1 << 20
    cx = dwt_amd.Context(local)
    try:
        pix = cx.synth_pixels(1, H, W, C, 0, 0)
        stride = (cap + 15) // 8 * 8
        streams = torch.empty((1, stride), dtype=torch.uint8, device=pix.device)
        info = torch.empty((1, ctypes.sizeof(dwt_amd.StreamInfo)), dtype=torch.uint8, device=pix.device)
        rec = {"workload": "one 16384x16384x3 frame, CAPACITY 1048576 (BASELINE.json configs[3])"}
        shas = {}
        for name, off in (("encode_ms", 0), ("encode_ms_all_segments_coded", 1)):
            cx.set_option("no_capacity_cut", off)
            cx.encode_device(pix, capacity=cap, out=streams, info=info)
            torch.cuda.synchronize()
            si = dwt_amd.StreamInfo.from_buffer_copy(info[0].cpu().numpy().tobytes())
            shas[name] = hashlib.sha256(streams[0, : si.nbytes].cpu().numpy().tobytes()).hexdigest()
            rec[name] = round(min(timed(lambda: cx.encode_device(pix, capacity=cap, out=streams, info=info))[0] for _ in range(3)), 3)
            if not off:
                rec["bytes"], rec["segments_coded"], rec["segments_cut"] = int(si.nbytes), int(si.segments), int(si.segments_cut)
        cx.set_option("no_capacity_cut", 0)
        cx.encode_device(pix, capacity=cap, out=streams, info=info)
        lens = cx.stream_lengths(info)
        out = torch.empty((1, W * H * C), dtype=torch.uint8, device=pix.device)
        _, dinfos = cx.decode_device(streams, lens, W, H, C, out=out)
        rec["decode_ms"] = round(min(timed(lambda: cx.decode_device(streams, lens, W, H, C, out=out))[0] for _ in range(3)), 3)
        gpath = os.path.join(ROOT, "tests", "golden", "golden.json")
        if os.path.exists(gpath):
            g = json.load(open(gpath)).get("c16384x16384_cap1MiB")
            if g:
                geo = dwt_amd.geometry(W, H)
                lo = dinfos[0].level + 1
                ow, oh = geo.widths[lo], geo.heights[lo]
                h = hashlib.sha256()
                flat = out[0, : ow * oh * C]
                for i in range(0, flat.numel(), 1 << 27):
                    h.update(flat[i:i + (1 << 27)].cpu().numpy().tobytes())
                rec["stream_matches_reference_golden"] = shas["encode_ms"] == g["dwt_sha256"] == shas["encode_ms_all_segments_coded"]
                rec["decoded_picture_matches_reference_golden"] = (ow, oh) == (g["dec_W"], g["dec_H"]) and h.hexdigest() == g["dec_sha256"]
        rec["Mpixels_per_s_roundtrip"] = round(W * H / ((rec["encode_ms"] + rec["decode_ms"]) * 1e-3) / 1e6, 1)
        rec["note"] = ("the floor of this encode is the part CAPACITY cannot cut: the whole forward transform and one pass over all "
                       "coefficients for the plane counts (encode.c:159-165 run in full in the reference too)")
        return rec
    finally:
        cx.close()


def golden_check(name, stream0):
    gpath = os.path.join(ROOT, "tests", "golden", "golden.json")
    gname = GOLDEN_OF.get(name)
    if not (gname and os.path.exists(gpath)):
        return None
    rec = json.load(open(gpath))[gname]
    return len(stream0) == rec["dwt_len"] and hashlib.sha256(stream0).hexdigest() == rec["dwt_sha256"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="gray4096", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (default per workload)")
    ap.add_argument("--cpu-frames", type=int, default=5, help="frames timed on the CPU reference (0 = skip)")
    ap.add_argument("--lift-reps", type=int, default=20)
    ap.add_argument("--extras", type=int, default=1, help="0: skip the workloads / single_frame sub-records")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--gather", default="packed", choices=["packed", "rows"],
                    help="N>1: one packed message per peer and step (dwtx_pack_streams), or one zero-copy send per frame")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: all ranks share cuda:0 (use with --backend gloo)")
    ap.add_argument("--dump-gathered", default="", help="rank 0 writes the last step's gathered streams here (tests)")
    ap.add_argument("--geometry", type=int, nargs=3, metavar=("W", "H", "C"), default=None,
                    help="tests: replace the workload's frame geometry (the line then names it; not a BASELINE config)")
    args = ap.parse_args()
    if args.geometry:
        W_, H_, C_ = args.geometry
        WORKLOADS[args.workload] = (W_, H_, C_, WORKLOADS[args.workload][3])
        CONFIG_OF[args.workload] = "test geometry, no BASELINE config"
        GOLDEN_OF.pop(args.workload, None)
        args.extras = 0

    from dwt_amd import launch
    if launch.needs_spawn(args.gpus):
        # GPU-free parent: start one process per GPU and hand their status on
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    world = launch.check_world(args.gpus)

    import torch
    import torch.distributed as dist

    import dwt_amd

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run = Runner(ctx, torch, dwt_amd, args.workload, args.frames, rank, world, dev, args.gather)
    W, H, C, B = run.W, run.H, run.C, run.B
    elapsed, (streams, lens, d, dinfos), k0 = run.timed(args.steps, args.warmup, fence)
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # what each rank spent on its own (so that a scaling run explains itself: a slow rank, or rank 0 paying for the gather)
        mine = torch.tensor([run.own_s, run.drain_s], dtype=torch.float64, device=dev)
        every = torch.zeros((world, 2), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every.view(-1), mine)
        per_rank = every.cpu().tolist()

    # ---- verification (outside the timed region) ---------------------------------
    pix = run.pix
    lossless = bool(torch.equal(d.view(B, H, W, C), pix)) and all(i.status == 0 and not i.truncated for i in dinfos)
    lens_host = lens.cpu().tolist()
    golden_ok = None
    gathered = None
    if rank == 0:
        golden_ok = golden_check(args.workload, streams[0, : lens_host[0]].cpu().numpy().tobytes())
    if world > 1:
        flag = torch.tensor([1 if lossless else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        lossless = bool(flag.item())
        got = run.gather.result(run.k - 1)
        all_lens = got.lens
        if rank == 0:
            # rank 0's own rows of the gathered set must be the streams it encoded
            mine_ok = all(bool(torch.equal(got.stream(0, i).to(dev), streams[i, : lens_host[i]])) for i in range(B))
            rccl_env = {k: os.environ[k] for k in sorted(os.environ) if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC"))}
            gathered = {"world": world, "backend": args.backend, "mode": args.gather, "frames": world * B, "own_rows_match": mine_ok,
                        "p2p_operations_posted_by_rank0": run.gather.messages_posted,
                        "p2p_operations_per_step_on_rank0": (world - 1) * (1 if args.gather == "packed" else B),
                        # how much of the chip RCCL may take on rank 0 is decided by these (unset = RCCL's defaults):
                        # a slow rank 0 in per_rank_ms_per_step with many channels allowed = its receive kernels beside the codec's
                        "rccl_env": rccl_env,
                        "per_rank_ms_per_step": [round(o / args.steps * 1e3, 3) for o, _ in per_rank],
                        "per_rank_drain_ms": [round(dr * 1e3, 3) for _, dr in per_rank],
                        "note": "per_rank_ms_per_step = a rank's own wall time for its steps incl. its share of the gather, before the "
                                "closing barrier; per_rank_drain_ms = after the last step was queued: device work still in flight "
                                "plus the last step's gather (the only one that cannot hide behind a following step)",
                        "bytes_last_step": int(all_lens.sum()),
                        "bytes_all_steps_incl_warmup": run.gather.bytes_gathered, "bytes_per_rank_last_step": [int(all_lens[r * B:(r + 1) * B].sum()) for r in range(world)]}
            if args.dump_gathered:
                import numpy as np
                width = int(max(int(v) for v in all_lens)) if len(all_lens) else 0
                rows = {}
                for r in range(world):
                    a = np.zeros((B, max(width, 1)), dtype=np.uint8)
                    for i in range(B):
                        v = got.stream(r, i).cpu().numpy()
                        a[i, : v.size] = v
                    rows[f"rank{r}"] = a
                np.savez(args.dump_gathered, lens=all_lens.numpy(), **rows)

    # ---- roofline of the lifting kernels, HIP events on the kernels' stream --------
    # (fresh allocations: three planes-sized tensors carved out of one cached multi-gigabyte block make the forward
    # kernel 12 % slower than separately allocated ones — tools/lift_offsets.py vs tools/time_lift.py)
    # The forward kernel's time is bimodal from one ALLOCATION to the next (34.4 or 39.4 us per 4096x4096 plane with
    # the buffers at the very same virtual addresses, tools/lift_lottery.py: the physical pages behind them differ):
    # three attempts on fresh buffers, all reported; the figure is their mean (what a rocprofv3 summary of this run averages to).
    # The kernel-level records (roofline, roofline_codec, coder) keep the 64 frames per launch they have been quoted on since round 1
    # (the rocprofv3 summaries and PMC passes under profiles/ are of that size); the timed step above runs the workload's own batch.
    RB = min(B, KERNEL_RECORD_FRAMES)
    rpix = pix[:RB]
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    attempts, copies = [], []
    lift_ms = None
    for attempt in range(3):
        planes = pyr = back = None
        torch.cuda.empty_cache()
        planes = ctx.planes_from_pixels(rpix)
        pyr = torch.empty_like(planes)
        back = torch.empty_like(planes)
        ctx.transformation_fwd(planes, pyr)
        ctx.transformation_inv(pyr, back)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.lift_reps):
            ctx.transformation_fwd(planes, pyr)
            ctx.transformation_inv(pyr, back)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.lift_reps
        attempts.append(round(ms * 1e3 / RB, 2))
        lift_ms = ms if lift_ms is None else lift_ms + ms
        # a plain copy of the same planes in the same buffers (8 B per sample): the streaming ceiling of this allocation
        e0.record()
        for _ in range(args.lift_reps):
            back.copy_(planes)
        e1.record()
        torch.cuda.synchronize()
        copies.append(round(e0.elapsed_time(e1) / args.lift_reps * 1e3 / RB, 2))
    lift_ms /= len(attempts)
    # (the separate forward / inverse loops below run on the last attempt's buffers)
    lift_ok = bool(torch.equal(back, planes))
    samples = RB * W * H * C
    achieved = LIFT_BYTES_PER_SAMPLE * samples / (lift_ms * 1e-3) / 1e9
    # each direction on its own (same kernels, separate loops)
    e0.record()
    for _ in range(args.lift_reps):
        ctx.transformation_fwd(planes, pyr)
    e1.record()
    for _ in range(args.lift_reps):
        ctx.transformation_inv(pyr, back)
    e2.record()
    torch.cuda.synchronize()
    fwd_ms, inv_ms = e0.elapsed_time(e1) / args.lift_reps, e1.elapsed_time(e2) / args.lift_reps

    # ---- the transform as the codec runs it (dwtx_transformation_fwd_pixels / _inv_pixels): u8 pixels in, the five finest rings
    # as int16, the tiles' histograms riding along; u8 pixels out.  Algorithmic bytes per sample: forward 1 (pixel) + 2 (int16
    # coefficient), inverse 2 + 1 = 6 for the pair (the int32 levels below the fifth are 1/1024 of the samples).
    roofline_codec = None
    if W % 4 == 0 and min(W, H) > 64:
        del back
        pyr_px, r16, m16 = ctx.transformation_fwd_pixels(rpix)
        out_px = ctx.transformation_inv_pixels(pyr_px, r16, m16, C)
        px_ok = bool(torch.equal(out_px, rpix))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.lift_reps):
            ctx.transformation_fwd_pixels(rpix, out=(pyr_px, r16))
        e1.record()
        for _ in range(args.lift_reps):
            ctx.transformation_inv_pixels(pyr_px, r16, m16, C, out=out_px)
        e2.record()
        torch.cuda.synchronize()
        fpx, ipx = e0.elapsed_time(e1) / args.lift_reps, e1.elapsed_time(e2) / args.lift_reps
        CODEC_LIFT_BYTES = 6
        ach = CODEC_LIFT_BYTES * samples / ((fpx + ipx) * 1e-3) / 1e9
        roofline_codec = {
            "kernel": "k_fwd_pixels_w<u8 | Rgb8, histograms> + k_fwd_level_w<int16> ... / k_inv_level_w<..., int16> ... k_inv_level_w<u8 | rgb>: "
                      "the transform inside dwtx_encode_device / dwtx_decode_device (dwtx_transformation_fwd_pixels / _inv_pixels)",
            "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "frames": RB, "bytes_per_sample": CODEC_LIFT_BYTES,
            "bytes_per_sample_note": "forward: 1 B pixel in + 2 B int16 coefficient out; inverse: 2 B in + 1 B out",
            "algorithmic_bytes": CODEC_LIFT_BYTES * samples,
            "forward_us_per_frame": round(fpx * 1e3 / RB, 2), "inverse_us_per_frame": round(ipx * 1e3 / RB, 2),
            "rings_as_int16_mask": m16, "roundtrip": px_ok, "traffic": None,
        }
        u8path = os.path.join(ROOT, "profiles", "r04_lift8_traffic_pmc.json")
        if os.path.exists(u8path) and C == 1:
            roofline_codec["finest_level_kernels_pmc"] = dict(json.load(open(u8path))["per_kernel"], source="profiles/r04_lift8_traffic_pmc.json (tools/pmc_lift8.sh)")
        del pyr_px, r16, out_px
        back = None
    lin = ctx.linearization(pyr)
    del back, planes
    coder = coder_record(ctx, torch, dwt_amd, lin, W, H, C, RB, run.stride, dev)
    del lin, pyr

    # ---- stage breakdown (one extra untimed pass with events) ----------------------
    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b), r

    enc_ms, _ = timed(lambda: ctx.encode_device(pix, out=run.out[0], info=run.info[0]))
    dec_ms, _ = timed(lambda: ctx.decode_device(run.out[0], ctx.stream_lengths(run.info[0]), W, H, C, out=run.dec))

    # HBM bytes of the same kernels from rocprofv3 PMC counters (FETCH_SIZE, WRITE_SIZE in separate passes),
    # collected in a separate profiling run (profiles/): the counters cannot be read from inside this process
    traffic = traffic_detail = None
    for tname in ("r04_lift_traffic_pmc.json", "r03_lift_traffic_pmc.json", "r02_lift_traffic_pmc.json", "r01_lift_traffic_pmc.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            per_sample = tj["traffic_bytes_per_sample"]
            traffic = int(per_sample * samples)   # HBM bytes of one forward+inverse pass over the step's planes, like `achieved`
            traffic_detail = {"bytes_per_sample": round(per_sample, 2), "algorithmic_bytes_per_sample": LIFT_BYTES_PER_SAMPLE,
                              "source": f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the "
                                        "same kernels; tools/pmc_lift.sh)"}
            for key in ("per_kernel", "planes"):
                if key in tj:
                    traffic_detail[key] = tj[key]
            # the finest level's u8 kernels as the codec runs them (pixels in; LL as int32, the detail bands as int16 out and back): 3.5 B per sample
            u8path = os.path.join(ROOT, "profiles", "r04_lift8_traffic_pmc.json")
            if os.path.exists(u8path):
                traffic_detail["u8_finest_level_kernels"] = dict(json.load(open(u8path))["per_kernel"], source="profiles/r04_lift8_traffic_pmc.json (tools/pmc_lift8.sh)")
            break

    result = None
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
            # what the path computes in: int32 like the reference's `int` (every level of the roofline kernels and the lower levels
            # of the codec); the codec's finest level runs in packed 16-bit arithmetic on u8 pixels and keeps the detail rings of the
            # five finest levels as int16 — provably and testedly the same values (an 8-bit source cannot leave 16 bits there)
            "dtype": "int32",
            "dtype_detail": "roofline kernels: int32 in, int32 out.  Timed codec step: u8 pixels in / out, finest lifting level in packed "
                            "int16 arithmetic, detail rings of the five finest levels stored as int16, everything else int32; identical results",
            "data": "synthetic",
            "config": {
                "workload": f"{W}x{H}x{C} 8-bit synthetic smooth+noise frames, lossless, {B} frames per GPU per step "
                            f"({CONFIG_OF[args.workload]})",
                "frames_per_gpu": B,
                "storage": "u8 pixels; int16 detail rings on the five finest levels, int32 below; .dwt streams",
                "parallelism": f"frames sharded over {world} GPU(s), one process per GPU, streams gathered to rank 0 "
                               f"({args.backend}) one step behind the encoder" if world > 1 else "1 GPU",
            },
            "per_gpu_value": round(B * W * H * args.steps / elapsed / 1e6, 3),
            "bit_exact": {"roundtrip_lossless": lossless, "stream0_matches_reference_golden": golden_ok,
                          "lifting_roundtrip": lift_ok},
            "bytes_per_frame": int(sum(lens_host) / len(lens_host)),
            "stage_ms_per_step": {"encode": round(enc_ms, 3), "decode": round(dec_ms, 3)},
            "coder": coder,
            "roofline": {
                "kernel": "k_fwd2_level_w + k_inv2_level_w (two levels per pass; + the LDS tail): forward+inverse CDF 5/3, all levels, on int32 planes — the "
                          "stage-level transform of include/dwtx.h (dwtx_transformation_fwd / _inv), SURVEY 8d's 16 B per sample; the codec's own "
                          "u8 / int16 kernels are priced in roofline_codec",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "read_only_frac": round(LIFT_READ_BYTES_PER_SAMPLE * samples / (lift_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_detail": traffic_detail,
                "algorithmic_bytes": LIFT_BYTES_PER_SAMPLE * samples,
                "bytes_per_sample": LIFT_BYTES_PER_SAMPLE,
                "frames": RB,
                "us_per_frame": round(lift_ms * 1e3 / RB, 2),
                "us_per_frame_attempts": attempts,
                "plain_copy_us_per_frame_same_buffers": copies,   # 8 B per sample each; forward + inverse are two such passes plus the pyramid's 1/3
                "vs_two_plain_copies": round(2 * sum(copies) / len(copies) / (lift_ms * 1e3 / RB), 3),
                "forward_us_per_frame": round(fwd_ms * 1e3 / RB, 2),
                "inverse_us_per_frame": round(inv_ms * 1e3 / RB, 2),
            },
        }
        result["roofline_codec"] = roofline_codec
        if gathered is not None:
            result["gathered"] = gathered

    def with_index(ctx, streams_dev, lens_dev, Wd, Hd, Cd, nd, out_dev, pix_dev):
        """The same decode offered the sidecar index an earlier decode of these streams produced (include/dwtx.h
        dwtx_index): ms per decode without / with it, and whether the pixels still come back."""
        made = ctx.set_index(None, nd)
        ctx.decode_device(streams_dev, lens_dev, Wd, Hd, Cd, out=out_dev)
        torch.cuda.synchronize()
        ctx.set_index()
        plain = min(timed(lambda: ctx.decode_device(streams_dev, lens_dev, Wd, Hd, Cd, out=out_dev))[0] for _ in range(3))
        ctx.set_index(made, 0)
        ctx.decode_device(streams_dev, lens_dev, Wd, Hd, Cd, out=out_dev)
        fast = min(timed(lambda: ctx.decode_device(streams_dev, lens_dev, Wd, Hd, Cd, out=out_dev))[0] for _ in range(3))
        ctx.set_index()
        same = bool(torch.equal(out_dev.view(nd, Hd, Wd, Cd), pix_dev))
        return {"decode_ms": round(plain, 3), "decode_ms_with_sidecar_index": round(fast, 3), "segments_indexed": int(made[0].nsegs),
                "lossless_with_index": same}

    if world == 1 and args.extras:
        # BASELINE.json configs[1] literally (one frame), configs[2], configs[3] and configs[4] geometry (short runs);
        # each on a context of its own, the main workload's scratch given back first (configs[2] needs most of the HBM)
        extras = {}
        del run, d, dinfos
        ctx.close()
        torch.cuda.empty_cache()
        ctx = dwt_amd.Context(local)
        one = Runner(ctx, torch, dwt_amd, "gray4096", 1, 0, 1, dev)
        t1, (s1, l1, d1, i1), _ = one.timed(5, 2, fence)
        em = min(timed(lambda: ctx.encode_device(one.pix, out=one.out[0], info=one.info[0]))[0] for _ in range(3))
        dm = min(timed(lambda: ctx.decode_device(one.out[0], ctx.stream_lengths(one.info[0]), 4096, 4096, 1, out=one.dec))[0] for _ in range(3))
        result["single_frame"] = {
            "workload": "one 4096x4096x1 frame per step (BASELINE.json configs[1] literally)",
            "value": round(5 * 4096 * 4096 / t1 / 1e6, 1), "unit": "Mpixels/s",
            "encode_ms": round(em, 3), "decode_ms": round(dm, 3),
            "lossless": bool(torch.equal(d1.view(1, 4096, 4096, 1), one.pix)),
            "matches_reference_golden": golden_check("gray4096", s1[0, : int(l1[0])].cpu().numpy().tobytes()),
        }
        result["single_frame"]["sidecar_index"] = with_index(ctx, one.out[0], ctx.stream_lengths(one.info[0]), 4096, 4096, 1, 1, one.dec, one.pix)
        del one, s1, d1
        if args.workload == "gray4096" and B != KERNEL_RECORD_FRAMES:
            # the batch the headline was quoted on until round 3, for continuity between the rounds' lines
            b64 = Runner(ctx, torch, dwt_amd, "gray4096", KERNEL_RECORD_FRAMES, 0, 1, dev)
            t64, (_, _, d64, _), _ = b64.timed(5, 2, fence)
            result["batch64"] = {"workload": f"{KERNEL_RECORD_FRAMES} frames per step, otherwise the main workload (rounds 1-3 quoted this batch)",
                                 "value": round(5 * KERNEL_RECORD_FRAMES * 4096 * 4096 / t64 / 1e6, 1), "unit": "Mpixels/s",
                                 "ms_per_step": round(t64 / 5 * 1e3, 3),
                                 "lossless": bool(torch.equal(d64.view(KERNEL_RECORD_FRAMES, 4096, 4096, 1), b64.pix))}
            del b64, d64
        ctx.close()
        torch.cuda.empty_cache()
        for name in ("rgb1080p", "rgb4096"):
            if name == args.workload:
                continue
            torch.cuda.empty_cache()
            # a context of its own: scratch sized and placed for this workload, as when it is the main one (inside the
            # scratch the gray batch left behind, 16 frames of 4096x4096 RGB ran anywhere between 18.5 and 27 ms per step)
            def attempt(frames):
                cx = dwt_amd.Context(local)
                try:
                    r = Runner(cx, torch, dwt_amd, name, frames, 0, 1, dev)
                    r.step()   # (sizes every scratch buffer)
                    torch.cuda.synchronize()
                    return cx, r, None
                except torch.cuda.OutOfMemoryError as err:
                    cx.close()
                    return None, None, f"torch.cuda.OutOfMemoryError: {str(err)[:200]}"
                except dwt_amd.DwtxError as err:
                    cx.close()
                    if err.rc != -5:     # DWTX_ERR_NOMEM is "did not fit"; anything else (a device fault above all) ends the run
                        raise
                    return None, None, f"DWTX_ERR_NOMEM: {str(err)[:200]}"

            cx, r2, why = attempt(0)
            if r2 is None:   # the configuration's batch did not fit beside what is resident: no smaller stand-in is timed
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                extras[name] = {"workload": f"{WORKLOADS[name][0]}x{WORKLOADS[name][1]}x{WORKLOADS[name][2]}, {WORKLOADS[name][3]} frames per step "
                                            f"({CONFIG_OF[name]})", "skipped": why}
                continue
            XS = 6   # steps of a side workload (2 warm-up steps: a 3-step run once caught a cold start and read 30 % low)
            t2, (s2, l2, d2, i2), _ = r2.timed(XS, 2, fence)
            ok2 = bool(torch.equal(d2.view(r2.B, r2.H, r2.W, r2.C), r2.pix)) and all(i.status == 0 and not i.truncated for i in i2)
            g2 = golden_check(name, s2[0, : int(l2[0])].cpu().numpy().tobytes())
            rec = {
                "workload": f"{r2.W}x{r2.H}x{r2.C}, {r2.B} frames per step, {XS} steps ({CONFIG_OF[name]}); the coder_* fields below: the "
                            f"entropy stage alone on {min(r2.B, 256)} of those frames",
                "value": round(XS * r2.B * r2.W * r2.H / t2 / 1e6, 1), "unit": "Mpixels/s",
                "Msamples_per_s": round(XS * r2.B * r2.W * r2.H * r2.C / t2 / 1e6, 1),
                "ms_per_step": round(t2 / XS * 1e3, 3), "bytes_per_frame": int(l2.sum().item() / r2.B),
                "roundtrip_lossless": ok2, "stream0_matches_reference_golden": g2,
                "sidecar_index": with_index(cx, r2.out[0], cx.stream_lengths(r2.info[0]), r2.W, r2.H, r2.C, r2.B, r2.dec, r2.pix),
            }
            # the entropy stage alone, on a context of its own once the workload's scratch is given back (at most 256 frames:
            # the three planes-sized tensors of the staged path would not fit beside a thousand frames' scratch)
            nb = min(r2.B, 256)
            cpix = r2.pix[:nb].clone()
            geo = (r2.W, r2.H, r2.C, r2.stride)
            del r2, s2, d2
            cx.close()
            del cx
            torch.cuda.empty_cache()
            cy = dwt_amd.Context(local)
            pl = cy.planes_from_pixels(cpix)
            lin2 = cy.linearization(cy.transformation_fwd(pl))
            del pl
            cr = coder_record(cy, torch, dwt_amd, lin2, geo[0], geo[1], geo[2], nb, geo[3], dev, reps=1)
            del lin2, cpix
            cy.close()
            del cy
            rec.update({"coder_frames": nb, "coder_encode_frac_of_hbm_peak": cr["encode"]["frac_of_hbm_peak"],
                        "coder_decode_frac_of_hbm_peak": cr["decode"]["frac_of_hbm_peak"],
                        "coder_encode_ms": cr["encode"]["ms_per_step"], "coder_decode_ms": cr["decode"]["ms_per_step"]})
            extras[name] = rec
        torch.cuda.empty_cache()
        extras["rgb16384_cap1MiB"] = capacity_record(dwt_amd, torch, local, timed)
        result["workloads"] = extras

    if rank == 0:
        if world == 1 and args.cpu_frames > 0:
            nf = min(args.cpu_frames, B)
            sample = pix[:nf].cpu().numpy()
            gs = [streams[i, : lens_host[i]].cpu().numpy().tobytes() for i in range(nf)]
            result["cpu_baseline"], same = cpu_baseline(W, H, C, nf, sample, gs)
            result["bit_exact"]["streams_equal_reference_bytes"] = result["cpu_baseline"]["gpu_streams_equal_cpu_bytes"]
            result["vs_cpu_baseline"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
