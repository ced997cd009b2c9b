# round-3 baseline: the GPU suite, the headline bench line and its kernel trace (run through gpurun from the repo root)
cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $o/t_all.log 2>&1; echo "pytest rc=$?"; tail -5 $o/t_all.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $o/bench_sr_base.json 2> $o/bench_sr_base.err && cat $o/bench_sr_base.json
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$o/stats_sr" -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > "$root/$o/stats_sr.log" 2>&1
cd "$root"
find $o/stats_sr -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $o/sr_kernel_stats.csv
cut -c1-180 $o/sr_kernel_stats.csv
