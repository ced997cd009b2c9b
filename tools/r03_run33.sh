#!/bin/bash
set -e
o=gpurun_out/r03; mkdir -p $o
cd "$(dirname "$0")/.."
python3 -m pytest tests/test_rife_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $o/t33.log
run() { python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d.get('parity'))" | tee -a $o/ab33.log; }
rm -f $o/ab33.log
for i in 1 2; do
  for s in 1 0; do
    echo "FW_IFNET_SKIP_PAD=$s" | tee -a $o/ab33.log
    FW_IFNET_SKIP_PAD=$s run --config rife
    FW_IFNET_SKIP_PAD=$s FW_RIFE_PAIR_STREAMS=1 run --config rife
  done
done
