#!/bin/bash
# tools/pmc_one.sh <kernel-substring> <cmd...>: SQ_INSTS_VALU / SALU / LDS / waves / wave cycles of one kernel (GPU box)
k=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pmc1 -- "$@" > /dev/null 2>&1
python3 - "$k" <<'PY'
import csv, glob, sys, collections
v = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(v["SQ_WAVES"]) or 1
print(" ".join(f"{c.replace('SQ_', '')}={sum(x) / w:.1f}" for c, x in sorted(v.items()) if c != "SQ_WAVES"), f"waves={w / max(1, len(v['SQ_WAVES'])):.0f}")
PY
