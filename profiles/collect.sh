#!/bin/bash
# Copies what profiles/refresh.sh wrote under gpurun_out/r02/ (merged back from the GPU box) into profiles/ under the names the docs
# cite: the four bench lines, the NEWEST kernel-stats file of each path (gpurun_out/ accumulates over calls), the PMC summary and
# the traffic file.  Run from the repo root after `gpurun -- bash profiles/refresh.sh`.
set -e
o=gpurun_out/r02
for c in sr rife tap chain; do cp $o/bench_$c.json profiles/r02_bench_$c.json; done
for c in sr tap rife restormer; do
    f=$(find $o/stats_$c -name "*_kernel_stats.csv" -printf "%T@ %p\n" | sort -n | tail -1 | cut -d" " -f2)
    cp "$f" profiles/r02_${c}_kernel_stats.csv
done
cp $o/pmc_summary.json profiles/r02_pmc_summary.json
cp $o/r02_traffic.json profiles/r02_traffic.json
cp $o/r02_traffic_tap.json $o/r02_traffic_restormer.json profiles/
