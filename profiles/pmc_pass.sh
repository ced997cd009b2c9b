#!/bin/bash
# usage: pmc_pass.sh <outdir> <counters...>   (run from the repo root on the GPU box)
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-host-path > "$out/bench.log" 2>&1
