#!/bin/bash
# Per-kernel utilisation counters of one encode+decode of 16 frames (run on the GPU box from the repo root):
#   tools/pmc_kernels.sh  ->  table on stdout.  Each counter group is its own rocprofv3 pass.
export DWTX_ONE_STREAM=1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "VALUBusy SALUBusy" "LDSBankConflict MemUnitStalled" "MemUnitBusy OccupancyPercent"; do
	rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pk$i -- python3 $GRAFT_REPO_ROOT/tools/time_codec.py 4096 4096 1 16 > /dev/null 2>&1
	i=$((i+1))
done
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pk0", "pk1", "pk2"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            vals[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["VALUBusy", "SALUBusy", "LDSBankConflict", "MemUnitStalled", "MemUnitBusy", "OccupancyPercent"]
print("kernel".ljust(34), " ".join(c[:12].rjust(12) for c in cols))
for n in sorted(vals, key=lambda k: -len(vals[k].get("VALUBusy", []))):
    if n.startswith("k_") or "ring" in n:
        print(n[:34].ljust(34), " ".join((f"{sum(vals[n][c])/len(vals[n][c]):12.1f}" if vals[n].get(c) else " " * 12) for c in cols))
PY
