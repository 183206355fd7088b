#!/bin/bash
# HBM traffic of the lifting kernels from PMC counters, two separate passes (run on the GPU box from the repo root):
#   tools/pmc_lift.sh [planes]  ->  gpurun_out/lift_traffic_pmc.json
# FETCH_SIZE is doubled for gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes; both counters are in KB.
P=${1:-8}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/time_lift.py 4096 $P > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/time_lift.py 4096 $P > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 - $P <<'PY'
import csv, glob, json, sys
P = int(sys.argv[1])
def total(d, counter):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)[0]
    s = 0.0; fwd0 = 0
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if r["Counter_Name"] == counter and ("k_fwd_" in n or "k_inv_" in n):
            s += float(r["Counter_Value"])
            if "k_fwd_tail" in n: fwd0 += 1
    return s, fwd0
fetch, nf = total("pmc_fetch", "FETCH_SIZE")
write, nw = total("pmc_write", "WRITE_SIZE")
assert nf == nw and nf > 0
samples = P * 4096 * 4096
traffic = (2 * fetch + write) * 1024 / nf
out = {"what": f"HBM traffic of one forward+inverse multi-level CDF 5/3 of {P} planes 4096x4096 int32 (all k_fwd_*/k_inv_* launches)",
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of tools/time_lift.py 4096 P (tools/pmc_lift.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); units KB; summed over all lifting launches and divided by the number of forward+inverse pairs",
       "pairs": nf, "fetch_size_kb_per_pair": fetch / nf, "write_size_kb_per_pair": write / nf,
       "traffic_bytes": traffic, "algorithmic_bytes": 16 * samples, "traffic_bytes_per_sample": traffic / samples}
json.dump(out, open("gpurun_out/lift_traffic_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
