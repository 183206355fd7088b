for cfg in "4096 4096 1 64" "1920 1080 3 256" "4096 4096 3 32" "1920 1080 3 1024"; do
  echo "== $cfg parts"; python3 tools/time_encode.py $cfg 2>&1 | grep -v amdgpu
  echo "== $cfg one stream"; DWTX_ONE_STREAM=1 python3 tools/time_encode.py $cfg 2>&1 | grep -v amdgpu
done
