set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_conv3x3_gpu.py tests/test_rrdbnet_gpu.py -x -q -m gpu > $o/t7.log 2>&1 || { tail -40 $o/t7.log; exit 1; }
tail -2 $o/t7.log
timeout -k 10 600 python tools/ab_variants.py "base=" "c5rev=ENV:FW_RRDB_C5_REV=1" "base_b=" "c5rev_b=ENV:FW_RRDB_C5_REV=1" > $o/ab7.log 2>&1
cut -c1-200 $o/ab7.log
bash tools/power_trace.sh $o/power_base.log
tail -12 $o/power_base.log; cat $o/power_base_bench.json | cut -c1-300
