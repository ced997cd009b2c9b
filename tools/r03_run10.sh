set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
timeout -k 10 400 python -m pytest tests/test_conv3x3_gpu.py -x -q -m gpu -k "pair" > $o/t10.log 2>&1 || { tail -40 $o/t10.log; exit 1; }
tail -2 $o/t10.log
timeout -k 10 900 python tools/ab_variants.py "slide32=ENV:FW_PAIR_SLIDE32=1" "nox=-DFW_P32_NOX ENV:FW_PAIR_SLIDE32=1" "slide30=" "slide32_b=ENV:FW_PAIR_SLIDE32=1" "slide30_b=" > $o/ab10.log 2>&1
cut -c1-200 $o/ab10.log
