#!/bin/bash
set -e
o=gpurun_out/r03; mkdir -p $o
cd "$(dirname "$0")/.."
python3 -m pytest tests/test_tap_gpu.py tests/test_restormer_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $o/t27.log
rm -rf $o/stats_tap27
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_tap27 -- python3 tools/profile_nafnet.py > $o/stats_tap27.log 2>&1
python3 - <<'PY' | tee $o/ab27.log
import csv, glob
f = glob.glob('gpurun_out/r03/stats_tap27/**/*_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'pointwise_mfma' in r['Name'] or 'pw_gemm' in r['Name']:
        print(r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3)
PY
FW_SUBMIT_NOTHREADS=1 python3 tools/host_submit.py nafnet 2>/dev/null | tee -a $o/ab27.log
