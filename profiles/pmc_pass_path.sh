#!/bin/bash
# usage: pmc_pass_path.sh <outdir> <tools/profile_*.py> <counters...>   (run from the repo root on the GPU box)
# The PMC pass of profiles/pmc_pass.sh for one of the secondary paths: a profile target (a few forwards of one engine) instead of bench.py.
set -e
out=$1; target=$2; shift; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out" -- python3 "$GRAFT_REPO_ROOT/$target" > "$out/run.log" 2>&1
