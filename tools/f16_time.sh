O=gpurun_out/ih; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_transform_gpu.py tests/test_codec_gpu.py tests/test_sweeps_gpu.py -x -q -k "not 16384 and not 1024_frames" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -1 $O/t.log
for r in 1 2; do python3 tools/time_lift.py 4096 64 2>&1 | grep -v amdgpu | tr '\n' ' '; echo; done
python3 tools/time_codec.py 4096 4096 1 64; python3 tools/time_codec.py 1920 1080 3 256
(cd /tmp && export TMPDIR=/tmp && DWTX_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/p -- python3 $GRAFT_REPO_ROOT/tools/time_codec.py 4096 4096 1 64 > $GRAFT_REPO_ROOT/$O/p.out 2>&1)
cp $(ls $O/p/*/*kernel_stats.csv | head -1) $O/g.csv
