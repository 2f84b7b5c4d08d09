#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) -> profiles/<tag>/ + profiles/k1_hbm_traffic.json.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes, are reported in KiB, and on gfx950 FETCH_SIZE shows exactly half of the bytes of a wide
(16 B per lane) coalesced streaming read, which is K1's access pattern (global_load_lds_dwordx4): reads are doubled.
WRITE_SIZE is checked against the known byte count of K1's output (1 bit per sample).
"""
import csv, collections, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(root, 'profiles', tag)
os.makedirs(dst, exist_ok=True)

def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg

summary = collections.defaultdict(dict)
for sub in ('pmc_fetch', 'pmc_write', 'pmc_sq_a', 'pmc_sq_b'):
    for k, cs in counters(sub).items():
        for c, v in cs.items():
            summary[k][c] = {'mean': sum(v) / len(v), 'launches': len(v)}
json.dump(summary, open(os.path.join(dst, 'pmc_summary.json'), 'w'), indent=1, sort_keys=True)
for f in glob.glob(os.path.join(src, 'stats', '**', '*kernel_stats.csv'), recursive=True):
    shutil.copy(f, os.path.join(dst, 'kernel_stats.csv'))
import re
names = [os.path.basename(p) for p in glob.glob(os.path.join(src, 'bench_*.log'))]
for name in names:
    p = os.path.join(src, name)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith('{')]
        if lines: open(os.path.join(dst, name.replace('.log', '.json')), 'w').write(lines[-1])
k1 = next((k for k in summary if 'k1t_demod<72, false' in k), None) or next((k for k in summary if 'k1_demod<72, false>' in k), None)
if k1 and 'FETCH_SIZE' in summary[k1] and 'WRITE_SIZE' in summary[k1]:
    fetch_kib, write_kib = summary[k1]['FETCH_SIZE']['mean'], summary[k1]['WRITE_SIZE']['mean']
    rd, wr = fetch_kib * 1024 * 2, write_kib * 1024
    out = {'kernel': k1, 'bytes_per_launch': rd + wr, 'read_bytes': rd, 'write_bytes': wr,
           'FETCH_SIZE_KiB_raw': fetch_kib, 'WRITE_SIZE_KiB_raw': write_kib,
           'correction': 'FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads), WRITE_SIZE x1; separate --pmc passes',
           'profile': 'profiles/' + tag + '/pmc_summary.json', 'tag': tag}
    json.dump(out, open(os.path.join(root, 'profiles', 'k1_hbm_traffic.json'), 'w'), indent=1)
    print(json.dumps(out))
print(open(os.path.join(dst, 'kernel_stats.csv')).read())
