#!/bin/bash
# The round's closing run on the GPU box: all GPU tests, the default bench line, the host-buffer (PCIe) rates, one codec sweep.
O=$GRAFT_REPO_ROOT/gpurun_out/r3final; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
python3 bench.py > $O/r03_bench_line.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 tools/time_host.py > $O/time_host.log 2>&1 || { tail -20 $O/time_host.log; exit 1; }
python3 tools/fuzz_codec.py 31 120 > $O/r03_fuzz_codec_seed31.log 2>&1 || { tail -20 $O/r03_fuzz_codec_seed31.log; exit 1; }
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
echo done > $O/DONE
