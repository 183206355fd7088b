cp dwt_amd/libdwtx.so /tmp/keep.so
for v in L0 L2 keep L0 L2; do
  if [ $v = keep ]; then cp /tmp/keep.so dwt_amd/libdwtx.so; else cp exp/libdwtx_$v.so dwt_amd/libdwtx.so; fi
  echo "== $v"
  python3 tools/time_lift.py 4096 64 2>&1 | grep -v amdgpu | tr '\n' ' '; echo
done
cp /tmp/keep.so dwt_amd/libdwtx.so
