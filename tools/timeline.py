#!/usr/bin/env python
"""Kernel timeline of steady-state steps out of a rocprofv3 --kernel-trace CSV: per dispatch start / end relative to the
K1 launch of its step, and the gaps on the compute stream.  usage: timeline.py <kernel_trace.csv> [n_steps_from_end]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 6
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^void ", "", n).replace("amr::", "")
    return n[:34]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows), key=lambda e: e[0])
k1 = [i for i, e in enumerate(ev) if e[2].startswith("k1t_demod") or e[2].startswith("k1_demod")]
if len(k1) < n_last + 2:
    sys.exit("too few K1 launches")
first = k1[-(n_last + 1)]
t0 = ev[first][0]
prev_end = {}
for s, e, n, q in ev[first:]:
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else float("nan")
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:7.1f} us  queue {q:>3}  gap on its queue {gap:7.1f}  {n}")
    prev_end[q] = e
    if n.startswith("k1"):
        pass
k1s = [ev[i][0] for i in k1[-(n_last + 1):]]
d = [(b - a) / 1e3 for a, b in zip(k1s, k1s[1:])]
print("K1-to-K1 period (us):", " ".join(f"{x:.1f}" for x in d), " mean", round(sum(d) / len(d), 1))
