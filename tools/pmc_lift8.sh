#!/bin/bash
# HBM traffic of the finest-level kernels the pipelines actually run on 8-bit gray pixels (k_fwd_pixels_w<unsigned char, true> with
# the fused histogram, k_inv2_level_w<unsigned char, true> = the two finest levels in one pass; the finest ring as 16-bit values), per sample: FETCH_SIZE / WRITE_SIZE in
# separate rocprofv3 passes over tools/time_codec.py 4096 4096 1 16, calibrated like tools/pmc_lift.sh on plain copies
# (4-byte accesses: the kernels' loads and most of their stores are one word per lane).   tools/pmc_lift8.sh > profiles/rNN_lift8_traffic_pmc.json
export DWTX_ONE_STREAM=1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
	rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc8_$c; rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc8_$c -- python3 $GRAFT_REPO_ROOT/tools/time_codec.py 4096 4096 1 16 > $GRAFT_REPO_ROOT/gpurun_out/pmc8_$c.log 2>&1 || { echo 'rocprofv3 failed:' >&2; tail -20 $GRAFT_REPO_ROOT/gpurun_out/pmc8_$c.log >&2; exit 1; }
	rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc8cal_$c; rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc8cal_$c -- $GRAFT_REPO_ROOT/tools/mb/mb_copy > $GRAFT_REPO_ROOT/gpurun_out/pmc8cal_$c.log 2>&1 || { echo 'rocprofv3 failed:' >&2; tail -20 $GRAFT_REPO_ROOT/gpurun_out/pmc8cal_$c.log >&2; exit 1; }
done
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import csv, glob, json, collections
def rows(d, counter):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            yield r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], float(r["Counter_Value"])
cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for n, v in rows(f"pmc8cal_{c}", c):
        acc[n].append(v)
    for n, v in acc.items():
        if n in ("copy16", "copy8", "copy4"):
            cal[(c, n)] = (1 << 20) / (sum(v) / len(v))
width = {"k_fwd_pixels_w<unsigned char, true>": (4, 4), "k_inv2_level_w<unsigned char, true>": (4, 4)}
out = {"what": "HBM traffic per sample of the finest-level kernels of dwtx_encode_device / dwtx_decode_device on 4096x4096 8-bit gray frames "
               "(16 frames per call; forward: 1 B of pixels in, LL out as int16 = 0.5 B per sample, the three detail bands as int16 = 1.5 B; inverse, two levels per pass: the second level's LL in as int32 = 0.25 B per pixel, both levels' details as int16 = 1.5 + 0.375 B, pixels out 1 B)",
       "calibration_true_over_reported": {f"{c}/{n}": round(v, 3) for (c, n), v in cal.items()}, "per_kernel": {}}
for kn, (lw, sw) in width.items():
    f = [v for n, v in rows("pmc8_FETCH_SIZE", "FETCH_SIZE") if n == kn]
    wv = [v for n, v in rows("pmc8_WRITE_SIZE", "WRITE_SIZE") if n == kn]
    samples = 16 * 4096 * 4096
    rd = sum(f) / len(f) * cal[("FETCH_SIZE", f"copy{lw}")] * 1024 / samples
    wr = sum(wv) / len(wv) * cal[("WRITE_SIZE", f"copy{sw}")] * 1024 / samples
    out["per_kernel"][kn] = {"launches": len(f), "read_bytes_per_sample": round(rd, 3), "write_bytes_per_sample": round(wr, 3),
                             "algorithmic_bytes_per_sample": 3.0 if kn.startswith("k_fwd") else 3.125}
print(json.dumps(out, indent=1))
PY
