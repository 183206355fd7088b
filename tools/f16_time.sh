O=gpurun_out/f16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && DWTX_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof2 -- python3 $GRAFT_REPO_ROOT/tools/time_codec.py 1920 1080 3 256 > $GRAFT_REPO_ROOT/$O/prof2.out 2>&1
cd $GRAFT_REPO_ROOT && cp $(ls $O/prof2/*/*kernel_stats.csv | head -1) $O/rgb1080_stats.csv
cat $O/prof2.out | grep -v amdgpu
