O=gpurun_out/p16; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_codec_gpu.py tests/test_cli_gpu.py tests/test_sweeps_gpu.py -x -q -k "not 16384 and not 1024_frames" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
python3 tools/time_codec.py 4096 4096 1 64; python3 tools/time_codec.py 1920 1080 3 256; python3 tools/time_codec.py 4096 4096 3 32
DWTX_NO_FINE16=1 python3 tools/time_codec.py 4096 4096 1 64
for cfg in "4096 4096 1 64 g"; do set -- $cfg
(cd /tmp && export TMPDIR=/tmp && DWTX_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/p$5 -- python3 $GRAFT_REPO_ROOT/tools/time_codec.py $1 $2 $3 $4 > $GRAFT_REPO_ROOT/$O/p$5.out 2>&1)
cp $(ls $O/p$5/*/*kernel_stats.csv | head -1) $O/$5.csv
done
