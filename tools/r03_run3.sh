set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_conv3x3_gpu.py tests/test_rrdbnet_gpu.py tests/test_pipeline_gpu.py tests/test_rife_gpu.py -x -q -m gpu > $o/t3.log 2>&1 || { tail -40 $o/t3.log; exit 1; }
tail -3 $o/t3.log
export FW_AB_TIME_UP=1
timeout -k 10 900 python tools/ab_variants.py "lines1_lo8=-DFW_UP_FULL_LINES=1" "lines0_lo8=-DFW_UP_FULL_LINES=0" "lines1_lo16=-DFW_UP_FULL_LINES=1 ENV:FW_RRDB_LO8=0" "lines1_lo8_b=-DFW_UP_FULL_LINES=1" "lines1_lo16_b=-DFW_UP_FULL_LINES=1 ENV:FW_RRDB_LO8=0" > $o/ab3.log 2>&1
cat $o/ab3.log
