#!/bin/bash
# Everything profiles/r03_* is made of (run on the GPU box from the repo root): gpurun_out/r3prof/
O=$GRAFT_REPO_ROOT/gpurun_out/r3prof; mkdir -p $O
prof() { # name cmd...
	n=$1; shift
	cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- "$@" > $O/$n.out 2> $O/$n.err
	cd $GRAFT_REPO_ROOT && cp $(ls $O/$n/*/*kernel_stats.csv | head -1) $O/$n.csv
}
tools/pmc_coder.sh 16 > $O/r03_coder_insts.json 2> $O/pmc_coder.err
mkdir -p profiles && cp $O/r03_coder_insts.json profiles/r03_coder_insts.json
prof r03_bench_gray4096_64frames_kernel_stats python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-frames 0 --extras 0
cp $O/r03_bench_gray4096_64frames_kernel_stats.out $O/r03_bench_line_under_rocprof.json
prof r03_lifting_only_64planes_kernel_stats python3 $GRAFT_REPO_ROOT/tools/time_lift.py 4096 64
export DWTX_ONE_STREAM=1
prof r03_codec_one_stream_gray4096_64frames_kernel_stats python3 $GRAFT_REPO_ROOT/tools/time_codec.py 4096 4096 1 64
prof r03_codec_one_stream_rgb1080p_256frames_kernel_stats python3 $GRAFT_REPO_ROOT/tools/time_codec.py 1920 1080 3 256
prof r03_codec_one_stream_rgb4096_32frames_kernel_stats python3 $GRAFT_REPO_ROOT/tools/time_codec.py 4096 4096 3 32
prof r03_capacity_16384_rgb_1MiB_kernel_stats python3 $GRAFT_REPO_ROOT/tools/time_capacity.py
unset DWTX_ONE_STREAM
tools/pmc_lift.sh 64 > $O/pmc_lift.out 2>&1; cp gpurun_out/lift_traffic_pmc.json $O/r03_lift_traffic_pmc.json
tools/pmc_lift8.sh > $O/r03_lift8_traffic_pmc.json 2> $O/pmc_lift8.err
tools/pmc_insts.sh > $O/r03_pmc_kernels.txt 2> $O/pmc_insts.err
python3 tools/fuzz_codec.py 31 120 > $O/r03_fuzz_codec_seed31.log 2>&1
python3 tools/fuzz_decode.py 32 60 > $O/r03_fuzz_decode_seed32.log 2>&1
python3 tools/fuzz_decode.py 33 12 big > $O/r03_fuzz_decode_seed33_big.log 2>&1
echo done > $O/DONE
