#!/bin/bash
# Everything profiles/r04_* is made of (run on the GPU box from the repo root): gpurun_out/r4prof/
O=$GRAFT_REPO_ROOT/gpurun_out/r4prof; mkdir -p $O
prof() { # name cmd...
	n=$1; shift
	rm -rf $O/$n
	cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- "$@" > $O/$n.out 2> $O/$n.err || { echo "rocprofv3 failed for $n" >&2; tail -5 $O/$n.err >&2; }
	cd $GRAFT_REPO_ROOT && cp $(ls $O/$n/*/*kernel_stats.csv | head -1) $O/$n.csv
}
tools/pmc_coder.sh 16 > $O/r04_coder_insts.json 2> $O/pmc_coder.err
mkdir -p profiles && cp $O/r04_coder_insts.json profiles/r04_coder_insts.json   # (bench.py reads it for the instruction roofline)
prof r04_bench_gray4096_128frames_kernel_stats python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-frames 0 --extras 0
cp $O/r04_bench_gray4096_128frames_kernel_stats.out $O/r04_bench_line_under_rocprof.json
prof r04_lifting_only_64planes_kernel_stats python3 $GRAFT_REPO_ROOT/tools/time_lift.py 4096 64
export DWTX_ONE_STREAM=1
prof r04_codec_one_stream_gray4096_64frames_kernel_stats python3 $GRAFT_REPO_ROOT/tools/time_codec.py 4096 4096 1 64
prof r04_codec_one_stream_rgb1080p_256frames_kernel_stats python3 $GRAFT_REPO_ROOT/tools/time_codec.py 1920 1080 3 256
prof r04_codec_one_stream_rgb4096_32frames_kernel_stats python3 $GRAFT_REPO_ROOT/tools/time_codec.py 4096 4096 3 32
unset DWTX_ONE_STREAM
tools/pmc_lift.sh 64 > $O/pmc_lift.out 2>&1; cp gpurun_out/lift_traffic_pmc.json $O/r04_lift_traffic_pmc.json
tools/pmc_lift8.sh > $O/r04_lift8_traffic_pmc.json 2> $O/pmc_lift8.err
tools/pmc_insts.sh > $O/r04_pmc_kernels.txt 2> $O/pmc_insts.err
echo done > $O/DONE
