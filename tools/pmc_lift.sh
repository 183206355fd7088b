#!/bin/bash
# HBM traffic of the lifting kernels from PMC counters, separate passes (run on the GPU box from the repo root):
#   tools/pmc_lift.sh [planes]  ->  gpurun_out/lift_traffic_pmc.json
# FETCH_SIZE / WRITE_SIZE are in KB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a wide
# (16 B per lane) coalesced streaming read; other widths are uncalibrated, so the factors for 16-, 8- and 4-byte
# accesses are measured here on plain copies of a known size (tools/mb/mb_copy) and applied per kernel:
# forward kernels load 16 B per lane, inverse kernels 8 B per lane.
P=${1:-64}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/time_lift.py 4096 $P > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1 || { echo 'rocprofv3 failed:' >&2; tail -20 $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log >&2; exit 1; }
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_write; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/time_lift.py 4096 $P > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1 || { echo 'rocprofv3 failed:' >&2; tail -20 $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log >&2; exit 1; }
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_cal_f; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_cal_f -- $GRAFT_REPO_ROOT/tools/mb/mb_copy > $GRAFT_REPO_ROOT/gpurun_out/pmc_cal_f.log 2>&1 || { echo 'rocprofv3 failed:' >&2; tail -20 $GRAFT_REPO_ROOT/gpurun_out/pmc_cal_f.log >&2; exit 1; }
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_cal_w; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_cal_w -- $GRAFT_REPO_ROOT/tools/mb/mb_copy > $GRAFT_REPO_ROOT/gpurun_out/pmc_cal_w.log 2>&1 || { echo 'rocprofv3 failed:' >&2; tail -20 $GRAFT_REPO_ROOT/gpurun_out/pmc_cal_w.log >&2; exit 1; }
cd $GRAFT_REPO_ROOT && python3 - $P <<'PY'
import csv, glob, json, sys, collections
P = int(sys.argv[1])
def rows(d, counter):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            yield r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], float(r["Counter_Value"])
# calibration: a 1 GiB copy moves 2^20 KB each way
cal = {}
for d, c in (("pmc_cal_f", "FETCH_SIZE"), ("pmc_cal_w", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for n, v in rows(d, c):
        acc[n].append(v)
    for n, v in acc.items():
        if n in ("copy16", "copy8", "copy4"):
            cal[(c, n)] = (1 << 20) / (sum(v) / len(v))      # true KB / reported KB
ff = {16: cal[("FETCH_SIZE", "copy16")], 8: cal[("FETCH_SIZE", "copy8")], 4: cal[("FETCH_SIZE", "copy4")]}
wf = {16: cal[("WRITE_SIZE", "copy16")], 8: cal[("WRITE_SIZE", "copy8")], 4: cal[("WRITE_SIZE", "copy4")]}
load_width = lambda n: 16 if n.startswith(("k_fwd_level_w", "k_fwd2_level_w")) else 8 if n.startswith(("k_inv_level_w", "k_inv2_level_w")) else 4
store_width = lambda n: 8 if n.startswith(("k_fwd_level_w", "k_fwd2_level_w")) else 16 if n.startswith(("k_inv_level_w", "k_inv2_level_w")) else 4
per = collections.defaultdict(lambda: {"fetch_kb": 0.0, "write_kb": 0.0, "launches": 0})
pairs = 0
for n, v in rows("pmc_fetch", "FETCH_SIZE"):
    if "k_fwd" in n or "k_inv" in n:
        per[n]["fetch_kb"] += v
        per[n]["launches"] += 1
        pairs += n.startswith("k_fwd_tail")
for n, v in rows("pmc_write", "WRITE_SIZE"):
    if "k_fwd" in n or "k_inv" in n:
        per[n]["write_kb"] += v
samples = P * 4096 * 4096
out_k = {}
tot = tot_raw2 = 0.0
for n, d in per.items():
    rd = d["fetch_kb"] * ff[load_width(n)] * 1024 / pairs
    wr = d["write_kb"] * wf[store_width(n)] * 1024 / pairs
    out_k[n] = {"load_bytes_per_lane": load_width(n), "fetch_factor": round(ff[load_width(n)], 3), "write_factor": round(wf[store_width(n)], 3),
                "read_bytes_per_sample": round(rd / samples, 3), "write_bytes_per_sample": round(wr / samples, 3),
                "launches_per_pair": d["launches"] // pairs}
    tot += rd + wr
    tot_raw2 += (2 * d["fetch_kb"] + d["write_kb"]) * 1024 / pairs
out = {"what": f"HBM traffic of one forward+inverse multi-level CDF 5/3 of {P} planes 4096x4096 int32 (all k_fwd_*/k_inv_* launches)",
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of tools/time_lift.py 4096 P (tools/pmc_lift.sh), units KB, "
                 "summed per kernel over all levels and divided by the number of forward+inverse pairs; each kernel's counters are scaled by the "
                 "factor (true bytes / reported bytes) measured in the same session on plain 1 GiB copies with the kernel's own access width "
                 "(tools/mb/mb_copy: copy16 / copy8 / copy4)",
       "planes": P, "pairs": pairs, "calibration_true_over_reported": {"FETCH_SIZE": {str(k): round(v, 3) for k, v in ff.items()},
                                                                        "WRITE_SIZE": {str(k): round(v, 3) for k, v in wf.items()}},
       "per_kernel": out_k, "traffic_bytes": tot, "algorithmic_bytes": 16 * samples, "traffic_bytes_per_sample": tot / samples,
       "traffic_bytes_per_sample_with_blanket_x2_on_fetch": tot_raw2 / samples}
json.dump(out, open("gpurun_out/lift_traffic_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
