#!/bin/bash
# usage: tools/prof_top.sh <outdir-name> <time_codec args...>   (run on the GPU box from the repo root)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/tools/time_codec.py "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 - "$out" <<'PY'
import csv,glob,sys
f=glob.glob(f"gpurun_out/{sys.argv[1]}/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:30]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:34]
    print(n.ljust(34), r["Calls"].rjust(5), f'{float(r["AverageNs"])/1e3:10.1f} us', r["Percentage"].rjust(7))
PY
