set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
export FW_AB_TIME_UP=1
timeout -k 10 1000 python tools/ab_variants.py "base=" "ncat4=ENV:FW_RRDB_NCAT4=1" "upslot8=-DFW_UP_SLOT_W=8" "upslot16=-DFW_UP_SLOT_W=16" "base_b=" "ncat4_b=ENV:FW_RRDB_NCAT4=1" > $o/ab6.log 2>&1
grep -E "^(base|ncat|upslot)" $o/ab6.log | cut -c1-260
