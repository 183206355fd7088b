#!/bin/bash
# per-kernel average durations of tools/time_codec.py at two batch sizes, side by side (run on the GPU box from the repo root):
#   tools/prof_batch.sh W H C n1 n2 [one_stream]  ->  gpurun_out/prof_batch_*.txt
W=$1; H=$2; C=$3; A=$4; B=$5; ONE=${6:-0}
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
[ "$ONE" = 1 ] && export DWTX_ONE_STREAM=1
for n in $A $B; do
	rm -rf $O/pb_$n
	(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/pb_$n -- python3 $GRAFT_REPO_ROOT/tools/time_codec.py $W $H $C $n > $O/pb_$n.out 2> $O/pb_$n.err) || { tail -5 $O/pb_$n.err; exit 1; }
	tail -1 $O/pb_$n.out
done
python3 - $A $B $O <<'PY'
import csv, glob, sys
A, B, O = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
def stats(n):
    f = glob.glob(f"{O}/pb_{n}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f))}
a, b = stats(A), stats(B)
print(f"{'kernel':40s} {'calls':>6s} {'us/frame @'+str(A):>16s} {'us/frame @'+str(B):>16s}  ratio")
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1]):
    if k in b and a[k][0] == b[k][0]:
        pa, pb = a[k][1] / A, b[k][1] / B
        print(f"{k[:40]:40s} {a[k][0]:6d} {pa:16.2f} {pb:16.2f}  {pb / pa:5.2f}")
PY
