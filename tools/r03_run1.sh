set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_conv3x3_gpu.py -x -q -k "up2x" > $o/t_up.log 2>&1 || { tail -30 $o/t_up.log; exit 1; }
tail -3 $o/t_up.log
timeout -k 10 400 python -m pytest tests/test_rrdbnet_gpu.py -x -q > $o/t_rrdb.log 2>&1 || { tail -30 $o/t_rrdb.log; exit 1; }
tail -3 $o/t_rrdb.log
FW_RRDB_UP_PHASE=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > $o/bench_up0.json 2> $o/bench_up0.err
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > $o/bench_up1.json 2> $o/bench_up1.err
FW_RRDB_UP_PHASE=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > $o/bench_up0b.json 2> $o/bench_up0b.err
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > $o/bench_up1b.json 2> $o/bench_up1b.err
python - <<'PY'
import json
for f in ["up0","up1","up0b","up1b"]:
    d=json.loads(open(f"gpurun_out/r03/bench_{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("parity"))
PY
