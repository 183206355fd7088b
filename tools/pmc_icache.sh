#!/bin/bash
export DWTX_ONE_STREAM=1
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pic
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pic -- python3 $GRAFT_REPO_ROOT/tools/time_codec.py 4096 4096 1 16 > $GRAFT_REPO_ROOT/gpurun_out/pic.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pic.log; exit 1; }
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        vals[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in sorted(vals):
    if n.startswith("k_tokenize") or n.startswith("k_link_first") or n.startswith("k_code"):
        print(n, {c: round(sum(v)/len(v)) for c, v in vals[n].items()})
PY
