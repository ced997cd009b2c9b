#!/bin/bash
# Copies what profiles/refresh.sh wrote under gpurun_out/r02/ (merged back from the GPU box) into profiles/ under the names the docs
# cite: the four bench lines, the NEWEST kernel-stats file of each path (gpurun_out/ accumulates over calls), the PMC summary and
# the traffic file.  Run from the repo root after `gpurun -- bash profiles/refresh.sh`.
set -e
R=${FW_ROUND:-r03}
o=gpurun_out/$R
for c in sr rife tap chain; do cp $o/bench_$c.json profiles/${R}_bench_$c.json; done
for c in sr tap rife restormer; do
    f=$(find $o/stats_$c -name "*_kernel_stats.csv" -printf "%T@ %p\n" | sort -n | tail -1 | cut -d" " -f2)
    cp "$f" profiles/${R}_${c}_kernel_stats.csv
done
cp $o/pmc_summary.json profiles/${R}_pmc_summary.json
cp $o/${R}_traffic.json profiles/${R}_traffic.json
cp $o/${R}_traffic_tap.json $o/${R}_traffic_restormer.json $o/${R}_traffic_rife.json profiles/
cp $o/power_trace.json profiles/${R}_power_trace.json
cp $o/energy.json profiles/${R}_energy.json
cp $o/energy_table.md profiles/${R}_energy_table.md
tail -1 $o/secondary_paths.json > profiles/${R}_secondary_paths.json
