# one sample of the headline on whatever box gpurun hands out (profiles/r03_box_spread.json): frames/s and the clock the chip holds at its power cap
cd $GRAFT_REPO_ROOT
o=gpurun_out/r03/box_$1
mkdir -p $o
python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-host-path > $o/bench.json 2> $o/bench.err
bash tools/power_trace.sh $o/power.log
python3 - <<PY
import json, re, statistics as st
b = json.loads(open("$o/bench.json").read().strip().splitlines()[-1])
t = open("$o/power.log").read()
pw = [float(x) for x in re.findall(r"Package Power \(W\): ([\d.]+)", t)]
sc = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", t)]
load = [s for p, s in zip(pw, sc) if p > 0.9 * max(pw)]
print(json.dumps({"frames_per_s": b["value"], "ms_per_frame": b["ms_per_step"], "frac": b["roofline"]["frac"], "sclk_mhz_under_load": st.median(load), "power_w": st.median([p for p in pw if p > 0.9 * max(pw)])}))
PY
