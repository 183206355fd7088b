for d in 0 8 3 11; do echo "dbg $d"; DWTX_PART_IMAGES=$d python tools/time_lift.py 4096 64 2>&1 | grep "^fwd:"; done
