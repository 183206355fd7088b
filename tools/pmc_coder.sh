#!/bin/bash
# Vector-instruction counts of the entropy stage's kernels (bench.py's instruction roofline): one rocprofv3 --pmc pass
# (no tracing flags with it) over tools/coder_only.py, SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVES per kernel, summed per
# encode / decode call and divided by the coefficients.  Run on the GPU box from the repo root:
#   tools/pmc_coder.sh [frames] > profiles/rNN_coder_insts.json
export DWTX_ONE_STREAM=1
F=${1:-16}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_coder
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_coder -- python3 $GRAFT_REPO_ROOT/tools/coder_only.py 4096 4096 1 $F 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_coder.log 2>&1 || { echo "rocprofv3 failed:" >&2; tail -20 $GRAFT_REPO_ROOT/gpurun_out/pmc_coder.log >&2; exit 1; }
cd $GRAFT_REPO_ROOT && python3 - "$F" <<'PY'
import csv, glob, collections, hashlib, json, sys
frames, reps = int(sys.argv[1]), 2
stamp = hashlib.sha256(open("dwt_amd/csrc/pack.hip", "rb").read() + open("dwt_amd/csrc/unpack.hip", "rb").read()).hexdigest()[:16]
ENC = ("k_hist", "k_plan", "k_entries", "k_cut", "k_stage_zero", "k_code", "k_carry", "k_gorder", "k_lut", "k_chain", "k_bitscan", "k_clear_stream", "k_emit", "k_refcopy", "k_order_emit")
DEC = ("k_nch", "k_peek", "k_clear_bitmaps", "k_tiles_init", "k_spec", "k_link", "k_scan", "k_tokenize", "k_hopbits", "k_rank", "k_count", "k_apply_all", "k_seg", "k_part_reset")
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob("gpurun_out/pmc_coder/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        tot[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            calls[n] += 1
coefs = frames * 4096 * 4096
out = {"what": f"rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES over tools/coder_only.py 4096 4096 1 {frames} {reps} "
               "(dwtx_encode_planes / dwtx_decode_planes on linearised coefficients; decoder on one stream)",
       "frames": frames, "sources_sha16": stamp, "per_kernel": {}}
for side, names in (("encode", ENC), ("decode", DEC)):
    valu = salu = 0.0
    for n, c in tot.items():
        if n.startswith(names) and not n.startswith("k_planes_"):   # (k_plan is the coder's, k_planes_from_pixels the tool's preparation)
            v, s_, wv = c["SQ_INSTS_VALU"] / reps, c["SQ_INSTS_SALU"] / reps, c["SQ_WAVES"] / reps
            valu += v
            salu += s_
            if v / coefs * 64 >= 0.5:
                out["per_kernel"][n] = {"side": side, "valu_ops_per_coefficient": round(v * 64 / coefs, 2), "valu_per_wave": round(v / max(wv, 1), 1),
                                        "salu_per_wave": round(s_ / max(wv, 1), 1), "waves_per_call": int(wv), "launches_per_call": calls[n] // reps}
    out[side] = {"valu_wave_insts_per_coefficient": valu / coefs, "valu_ops_per_coefficient": round(valu * 64 / coefs, 1),
                 "salu_wave_insts_per_coefficient": salu / coefs}
print(json.dumps(out, indent=1))
PY
