#!/usr/bin/env python3
"""Per-launch summary of a K1 workgroup timeline (diagnostic builds: tools/build_variant.sh tl "-DAMR_K1T_CLK=1", run with
AMR_K1_TIMELINE=file; amr_destroy writes one line per workgroup of the last 64 launches).

For every launch: when its workgroups started (spread after the first), how long they ran, when they ended, per XCC.
usage: k1_timeline_report.py file [launches per batch]"""
import sys
import numpy as np

def main():
    path = sys.argv[1]
    per_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    d = np.loadtxt(path, dtype=np.uint64, comments="#", ndmin=2)
    slots = {}
    for s in np.unique(d[:, 0]):
        r = d[d[:, 0] == s]
        slots[int(s)] = r
    order = sorted(slots, key=lambda s: slots[s][:, 2].min())     # by time: the ring wraps
    print(f"# {path}: {len(order)} launches; us; start = after the launch's first workgroup; dur = per workgroup")
    print("# launch  wgs   gap   span | start p50  p90  p99  max | dur p1   p50   p99 | late>2us | per XCC: mean dur (wgs)")
    prev_end = None
    agg = {}
    for k, s in enumerate(order):
        r = slots[s]
        st = r[:, 2].astype(np.float64) * 0.01; en = r[:, 3].astype(np.float64) * 0.01
        t0 = st.min(); span = en.max() - t0
        rel = st - t0; dur = en - st
        gap = (t0 - prev_end) if prev_end is not None else float("nan")
        prev_end = en.max()
        xcc = (r[:, 4] & np.uint64(0xf)).astype(int)
        per = " ".join(f"{dur[xcc == x].mean():.0f}({(xcc == x).sum()})" for x in range(8) if (xcc == x).any())
        pc = lambda v, q: np.percentile(v, q)
        print(f"{k:4d}/{s:2d} {len(r):5d} {gap:6.1f} {span:6.1f} | {pc(rel,50):6.1f} {pc(rel,90):5.1f} {pc(rel,99):5.1f} {rel.max():5.1f} | "
              f"{pc(dur,1):6.1f} {pc(dur,50):6.1f} {pc(dur,99):6.1f} | {(rel > 2).sum():5d} | {per}")
        if len(r) >= 1024 and k >= per_batch:     # whole launches only, the first batch in the ring may be cut
            a = agg.setdefault(s % per_batch if per_batch > 1 else 0, [])
            a.append((span, pc(rel, 99), rel.max(), pc(dur, 50), (rel > 2).sum(), gap))
    for ph, a in sorted(agg.items()):
        a = np.array(a)
        print(f"# launch {ph} of {per_batch} per batch: n {len(a)}  span mean {a[:,0].mean():.1f} (min {a[:,0].min():.1f} max {a[:,0].max():.1f})  "
              f"start p99 {a[:,1].mean():.1f}  start max {a[:,2].mean():.1f}  dur p50 {a[:,3].mean():.1f}  late>2us {a[:,4].mean():.0f}  gap before {np.nanmean(a[:,5]):.1f}")

if __name__ == "__main__":
    main()
