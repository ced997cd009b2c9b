cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_restormer_gpu.py tests/test_tap_gpu.py -x -q -m gpu > $o/t13.log 2>&1 || { tail -60 $o/t13.log; exit 1; }
tail -3 $o/t13.log
FW_AB_CHILD=restormer timeout -k 10 300 python tools/ab_front.py pw_dw_fused.hip "rest_mfma=" "rest_valu=ENV:FW_PW_DW_MFMA=0" "rest_mfma_b=" "rest_valu_b=ENV:FW_PW_DW_MFMA=0" > $o/ab13.log 2>&1
timeout -k 10 300 python tools/ab_front.py pw_dw_fused.hip "naf_mfma=" "naf_valu=ENV:FW_PW_DW_MFMA=0" "naf_mfma_b=" "naf_valu_b=ENV:FW_PW_DW_MFMA=0" >> $o/ab13.log 2>&1
cat $o/ab13.log
FW_STAMP_CHILD=restormer FW_STAMP_TAIL=20000 timeout -k 10 300 python tools/front_stamps.py > $o/stamps_mfma2.log 2>&1
grep "cg=6 mode 2 wg 3 wave 5" $o/stamps_mfma2.log | tail -2
grep "cg=6 mode 0 wg 3 wave 5" $o/stamps_mfma2.log | tail -2
