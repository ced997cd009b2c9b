set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
timeout -k 10 1000 python -m pytest tests -q -m gpu > $o/t_all.log 2>&1 || { tail -60 $o/t_all.log; }
tail -5 $o/t_all.log
FW_NAF_GRAPH=0 FW_RRDB_GRAPH=0 FW_IFNET_GRAPH=0 timeout -k 10 300 python bench.py --config chain --no-cpu-baseline > $o/bench_chain_nograph.json 2> $o/bench_chain_nograph.err
timeout -k 10 300 python bench.py --config chain --no-cpu-baseline > $o/bench_chain_graph.json 2> $o/bench_chain_graph.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $o/bench_sr_a.json 2> $o/bench_sr_a.err
python - <<'PY'
import json
for f in ["chain_nograph","chain_graph","sr_a"]:
    d=json.loads(open(f"gpurun_out/r03/bench_{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("parity"), d.get("host_to_host"))
PY
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$o/stats_sr" -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > "$root/$o/stats_sr.log" 2>&1
cd "$root"
find $o/stats_sr -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $o/sr_kernel_stats.csv
head -12 $o/sr_kernel_stats.csv | cut -c1-200
