#!/bin/bash
# Samples rocm-smi (average socket power, sclk, temperature) twice a second while the headline bench runs: what the chip does under
# the conv kernels (the kernels run against the power limit).  Usage (on the GPU box): bash tools/power_trace.sh OUT.log
out=${1:-gpurun_out/power_trace.log}
python3 bench.py --steps 150 --warmup 3 --no-cpu-baseline --no-host-path > "${out%.log}_bench.json" 2> /dev/null &
pid=$!
: > "$out"
while kill -0 $pid 2> /dev/null; do
    echo "t=$(date +%s.%N)" >> "$out"
    rocm-smi --showpower --showclocks --showtemp --showperflevel 2> /dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)|Performance Level" >> "$out"
    sleep 0.5
done
wait $pid
