#!/bin/bash
# registers / LDS / scratch of the kernels of one source file (development aid; runs in the build container):
#   tools/kinfo.sh unpack [pattern]
f=$1; pat=${2:-.}
d=$(mktemp -d); cd $d
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function --cuda-device-only -save-temps -c /root/repo/dwt_amd/csrc/$f.hip -o x.o 2>/dev/null
grep -E "^\s+\.(name|vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill_count):" *.s | sed 's/^[^:]*://' | paste - - - - - - | sed 's/\s\+/ /g' | grep -E "$pat"
rm -rf $d
