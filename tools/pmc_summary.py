#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel.  usage: pmc_summary.py <dir>... [--kernel substr]"""
import csv, collections, glob, sys
dirs = [a for a in sys.argv[1:] if not a.startswith('--')]
sub = None
if '--kernel' in sys.argv: sub = sys.argv[sys.argv.index('--kernel') + 1]; dirs = [d for d in dirs if d != sub]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if sub and sub not in k: continue
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f'   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})')
