set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
timeout -k 10 900 python tools/ab_variants.py "base=" "rdb3_nolo_write=ENV:FW_RRDB_ABL_RDB3=1" "rdb3_no_rlo=ENV:FW_RRDB_ABL_RDB3=2" "rdb3_no_rhi=ENV:FW_RRDB_ABL_RDB3=4" "rdb3_none=ENV:FW_RRDB_ABL_RDB3=7" "base_b=" > $o/ab4.log 2>&1
grep -E "^(base|rdb3)" $o/ab4.log | cut -c1-200
