set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_conv3x3_gpu.py tests/test_rrdbnet_gpu.py -x -q -m gpu > $o/t5.log 2>&1 || { tail -40 $o/t5.log; exit 1; }
tail -2 $o/t5.log
timeout -k 10 900 python tools/ab_variants.py "slot10=" "slot0=-DFW_RES_SLOT=0" "slot20=-DFW_RES_SLOT=20" "slot10_b=" "slot10_none=ENV:FW_RRDB_ABL_RDB3=7" > $o/ab5.log 2>&1
grep -E "^slot" $o/ab5.log | cut -c1-200
