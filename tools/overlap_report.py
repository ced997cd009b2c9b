#!/usr/bin/env python3
"""How much do kernels of different streams overlap?  Reads a rocprofv3 `*_kernel_trace.csv` and prints, for the second half of the
trace (past the warm-up): the span, the sum of kernel durations, the time at least one / two / three kernels were running, and the
launches per hardware queue.

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x -- python3 bench.py --config rife --steps 30 --no-cpu-baseline
  python3 tools/overlap_report.py $(find gpurun_out/x -name '*_kernel_trace.csv')
"""
import csv
import sys
from collections import Counter


def main(path: str) -> None:
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows)
    ev = ev[len(ev) // 2:]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    pts = []
    for s, e, _, _ in ev:
        pts.append((s, 1))
        pts.append((e, -1))
    pts.sort()
    depth, last, at = 0, pts[0][0], Counter()
    for t, d in pts:
        at[depth] += t - last
        last = t
        depth += d
    span = t1 - t0
    total = sum(e - s for s, e, _, _ in ev)
    print(f"kernels {len(ev)}  span {span / 1e6:.3f} ms  sum of durations {total / 1e6:.3f} ms  ratio {total / span:.2f}")
    for k in sorted(at):
        print(f"  {k} kernel(s) running: {at[k] / 1e6:8.3f} ms  ({100.0 * at[k] / span:5.1f} %)")
    print("launches per queue:", dict(Counter(q for _, _, q, _ in ev)))


if __name__ == "__main__":
    main(sys.argv[1])
