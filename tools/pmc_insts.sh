#!/bin/bash
# Instruction mix and utilisation of the entropy-stage kernels for one encode+decode of 16 frames
# (run on the GPU box from the repo root):  tools/pmc_insts.sh [outfile]  -> table on stdout.
# Each counter group is its own rocprofv3 pass (no tracing flags together with --pmc).
export DWTX_ONE_STREAM=1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "VALUBusy OccupancyPercent" "LDSBankConflict MemUnitStalled"; do
	rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pi$i -- python3 $GRAFT_REPO_ROOT/tools/time_codec.py 4096 4096 1 16 > /dev/null 2>&1
	i=$((i+1))
done
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pi0", "pi1", "pi2", "pi3"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            vals[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES",
        "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "VALUBusy", "OccupancyPercent", "LDSBankConflict", "MemUnitStalled"]
print("kernel (per-wave averages for the SQ_INSTS/CYCLES columns)".ljust(30), " ".join(c.replace("SQ_", "")[:11].rjust(11) for c in cols))
for n in sorted(vals):
    if not (n.startswith("k_")):
        continue
    v = vals[n]
    waves = sum(v["SQ_WAVES"]) / max(1, len(v["SQ_WAVES"])) if v.get("SQ_WAVES") else 0
    out = []
    for c in cols:
        if not v.get(c):
            out.append(" " * 11); continue
        a = sum(v[c]) / len(v[c])
        if c.startswith("SQ_") and c != "SQ_WAVES" and waves:
            a = a / waves
        out.append(f"{a:11.1f}")
    print(n[:30].ljust(30), " ".join(out))
PY
