#!/usr/bin/env python
"""Summarise a gpurun_out/<tag>/ directory of this round's GPU jobs: bench lines and rocprofv3 kernel stats."""
import csv, glob, json, os, sys
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "bench_*.log"))):
    try:
        j = json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e:
        print(os.path.basename(f), "NO JSON:", open(f).read()[-300:].replace("\n", " | "))
        continue
    r = j["roofline"]
    bad = [k for k, v in j["config"]["checks"].items() if "MISMATCH" in str(v)]
    print(f"{os.path.basename(f)[6:-4]:16s} {j['value']/1e6:6.3f}e6 step {j['ms_per_step']:.4f} steady {j['steady_ms_per_step']:.4f} "
          f"k1 {r['k1_ms']:.4f} frac {r['frac']:.3f} search {r['search_ms']:.4f} whole {r['whole_path_frac']:.3f} {'BAD ' + str(bad) if bad else 'checks ok'}")
for p in sorted(glob.glob(os.path.join(d, "prof_*", "prof_kernel_stats.csv"))):
    print("==", p.split("/")[-2])
    for r in list(csv.DictReader(open(p)))[:6]:
        print(f"   {r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f} us")
