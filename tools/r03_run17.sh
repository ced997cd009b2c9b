cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_rrdbnet_gpu.py -x -q -s -k "winograd" > $o/t17.log 2>&1; echo rc=$?; grep -E "winograd|passed|failed|Error|error" $o/t17.log | tail -12
timeout -k 10 600 python tools/ab_variants.py "direct=" "wino=ENV:FW_RRDB_C5_WINO=1" "direct_b=" "wino_b=ENV:FW_RRDB_C5_WINO=1" > $o/ab17.log 2>&1
cut -c1-200 $o/ab17.log
