#!/bin/bash
# Regenerates the judged profile artefacts of a round (FW_ROUND, default r03) on the GPU box (run from the repo root through gpurun; ~9 min):
#   gpurun_out/$R/pmc/p1,p2   FETCH_SIZE / WRITE_SIZE passes of the headline bench (separate passes), p3: MFMA-busy / LDS counters
#   profiles/${R}_traffic.json  per-launch and per-frame HBM traffic from p1 + p2, tagged with the build's digest (framewright_amd.build.source_digest(): sources + flags; written first,
#                              so that the bench line that follows quotes the traffic of THIS build)
#   gpurun_out/$R/bench_{sr,rife,tap,chain}.json   the four bench lines
#   gpurun_out/$R/stats_{sr,tap,rife,restormer}/   rocprofv3 --kernel-trace --stats summaries
set -e
root=$GRAFT_REPO_ROOT
R=${FW_ROUND:-r03}
o="$root/gpurun_out/$R"
mkdir -p "$o"
bash "$root/profiles/pmc_pass.sh" "$o/pmc/p1" FETCH_SIZE
bash "$root/profiles/pmc_pass.sh" "$o/pmc/p2" WRITE_SIZE
bash "$root/profiles/pmc_pass.sh" "$o/pmc/p3" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
cd "$root"
python3 tools/make_traffic.py "$o/pmc" profiles/${R}_traffic.json f16 > /dev/null
cp profiles/${R}_traffic.json "$o/${R}_traffic.json"
python3 profiles/pmc_summarize.py "$o/pmc" "$o/pmc_summary.json" || true
# HBM traffic of one NAFNet 1080p forward and of one Restormer 512x512 tile (all kernels), for the `tap` bench line and DESIGN section 6
bash "$root/profiles/pmc_pass_path.sh" "$o/pmc_tap/p1" tools/profile_nafnet.py FETCH_SIZE
bash "$root/profiles/pmc_pass_path.sh" "$o/pmc_tap/p2" tools/profile_nafnet.py WRITE_SIZE
bash "$root/profiles/pmc_pass_path.sh" "$o/pmc_rife/p1" tools/profile_ifnet.py FETCH_SIZE
bash "$root/profiles/pmc_pass_path.sh" "$o/pmc_rife/p2" tools/profile_ifnet.py WRITE_SIZE
bash "$root/profiles/pmc_pass_path.sh" "$o/pmc_restormer/p1" tools/profile_restormer.py FETCH_SIZE
bash "$root/profiles/pmc_pass_path.sh" "$o/pmc_restormer/p2" tools/profile_restormer.py WRITE_SIZE
cd "$root"
python3 tools/make_traffic_path.py "$o/pmc_tap" profiles/${R}_traffic_tap.json 4 f16 > /dev/null
python3 tools/make_traffic_path.py "$o/pmc_restormer" profiles/${R}_traffic_restormer.json 3 f16 > /dev/null
python3 tools/make_traffic_path.py "$o/pmc_rife" profiles/${R}_traffic_rife.json 5 f16 > /dev/null
cp profiles/${R}_traffic_tap.json profiles/${R}_traffic_restormer.json profiles/${R}_traffic_rife.json "$o/"
python3 bench.py > "$o/bench_sr.json" 2> "$o/bench_sr.err"
python3 bench.py --config rife > "$o/bench_rife.json" 2> "$o/bench_rife.err"
python3 bench.py --config tap > "$o/bench_tap.json" 2> "$o/bench_tap.err"
python3 bench.py --config chain > "$o/bench_chain.json" 2> "$o/bench_chain.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$o/stats_sr" -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > "$o/stats_sr.log" 2>&1
cd "$root"
# socket power under the headline bench (same box as the kernel trace above) -> joules per frame and per launch class; the other paths' timings
bash tools/power_trace.sh "$o/power_trace.log"
python3 tools/energy_table.py "$o/power_trace.log" "$o/power_trace_bench.json" "$(find $o/stats_sr -name "*_kernel_trace.csv" -printf "%T@ %p\n" | sort -n | tail -1 | cut -d" " -f2)" "$o/power_trace.json" "$o/energy.json" > "$o/energy_table.md"
python3 tools/time_paths.py > "$o/secondary_paths.json" 2> "$o/secondary_paths.err" || true
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$o/stats_tap" -- python3 "$root/tools/profile_nafnet.py" > "$o/stats_tap.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$o/stats_rife" -- python3 "$root/tools/profile_ifnet.py" > "$o/stats_rife.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$o/stats_restormer" -- python3 "$root/tools/profile_restormer.py" > "$o/stats_restormer.log" 2>&1
