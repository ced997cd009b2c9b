#!/bin/bash
# Regenerates the judged profile artefacts on the GPU box (run from the repo root through gpurun):
#   gpurun_out/pmc/p1,p2      FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1 --warmup 1` (separate passes)
#   gpurun_out/pmc/p3         MFMA-busy / LDS-conflict counters
#   gpurun_out/r01_traffic.json  per-launch HBM traffic from p1 + p2 (also written to profiles/ so that the bench line
#                                that follows quotes the traffic of THIS build)
#   gpurun_out/bench.json     the default bench line
#   gpurun_out/stats/         rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2`
set -e
root=$GRAFT_REPO_ROOT
mkdir -p "$root/gpurun_out"
bash "$root/profiles/pmc_pass.sh" "$root/gpurun_out/pmc/p1" FETCH_SIZE
bash "$root/profiles/pmc_pass.sh" "$root/gpurun_out/pmc/p2" WRITE_SIZE
bash "$root/profiles/pmc_pass.sh" "$root/gpurun_out/pmc/p3" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
cd "$root"
python3 tools/make_traffic.py gpurun_out/pmc profiles/r01_traffic.json > /dev/null
cp profiles/r01_traffic.json gpurun_out/r01_traffic.json
python3 bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/stats" -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$root/gpurun_out/stats.log" 2>&1
