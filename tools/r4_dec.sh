#!/bin/bash
# decoder experiment sweep on the GPU box (development aid): per-kernel table on one stream, then timings with several
# speculation starts
O=$GRAFT_REPO_ROOT/gpurun_out
DWTX_ONE_STREAM=1 tools/prof_top.sh r4_prof_b 4096 4096 1 64 > $O/r4_prof_b.txt 2>&1
for s in 0 32 64 80 96; do
	echo "spec_start $s"; DWTX_SPEC_START=$s python3 tools/time_codec.py 4096 4096 1 64; DWTX_SPEC_START=$s python3 tools/time_codec.py 4096 4096 1 1
	DWTX_SPEC_START=$s DWTX_ONE_STREAM=1 python3 tools/time_codec.py 4096 4096 1 64
done
