#!/bin/bash
# Regenerates the judged profile artefacts on the GPU box (run from the repo root through gpurun):
#   gpurun_out/stats/   rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2`
#   gpurun_out/pmc/p1,p2  FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1 --warmup 1`
#   gpurun_out/bench.json  the default bench line
set -e
root=$GRAFT_REPO_ROOT
mkdir -p "$root/gpurun_out"
python3 "$root/bench.py" > "$root/gpurun_out/bench.json" 2> "$root/gpurun_out/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/stats" -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$root/gpurun_out/stats.log" 2>&1
bash "$root/profiles/pmc_pass.sh" "$root/gpurun_out/pmc/p1" FETCH_SIZE
bash "$root/profiles/pmc_pass.sh" "$root/gpurun_out/pmc/p2" WRITE_SIZE
