#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/collect_profiles_r03.sh) -> profiles/<tag>/ + profiles/k1_hbm_traffic.json.

Per workload (cfg2, cfg3, cfg5): kernel_stats_<w>.csv (pipelined), kernel_stats_<w>_depth1.csv (the kernels one after
the other), pmc_summary_<w>.json (mean counter value per kernel and launch).  HBM traffic follows
/opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE come from separate --pmc
passes, are reported in KiB, and on gfx950 FETCH_SIZE shows exactly half of the bytes of a wide (16 B per lane)
coalesced streaming read, which is K1's access pattern (global_load_lds_dwordx4): K1's reads are doubled.
"""
import csv, collections, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(root, 'profiles', tag)
os.makedirs(dst, exist_ok=True)
for w in ('cfg2', 'cfg3', 'cfg5'):
    summary = collections.defaultdict(dict)
    for sub in ('pmc_fetch', 'pmc_write', 'pmc_sq_a', 'pmc_sq_b'):   # (round 4: the PMC passes run bench.py --depth 1)
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(src, w, sub, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in agg.items():
            for c, v in cs.items():
                summary[k][c] = {'mean': sum(v) / len(v), 'launches': len(v)}
    if summary:
        json.dump(summary, open(os.path.join(dst, f'pmc_summary_{w}.json'), 'w'), indent=1, sort_keys=True)
    for sub, name in (('stats', f'kernel_stats_{w}.csv'), ('stats_iso', f'kernel_stats_{w}_depth1.csv')):
        for f in glob.glob(os.path.join(src, w, sub, '**', '*kernel_stats.csv'), recursive=True):
            shutil.copy(f, os.path.join(dst, name))
    if w == 'cfg2':
        k1 = next((k for k in summary if 'k1t_demod<72, false' in k), None)
        if k1 and 'FETCH_SIZE' in summary[k1] and 'WRITE_SIZE' in summary[k1]:
            fetch_kib, write_kib = summary[k1]['FETCH_SIZE']['mean'], summary[k1]['WRITE_SIZE']['mean']
            rd, wr = fetch_kib * 1024 * 2, write_kib * 1024
            out = {'kernel': k1, 'bytes_per_launch': rd + wr, 'read_bytes': rd, 'write_bytes': wr,
                   'FETCH_SIZE_KiB_raw': fetch_kib, 'WRITE_SIZE_KiB_raw': write_kib,
                   'correction': 'FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads), WRITE_SIZE x1; separate --pmc passes',
                   'profile': f'profiles/{tag}/pmc_summary_cfg2.json', 'tag': tag}
            json.dump(out, open(os.path.join(root, 'profiles', 'k1_hbm_traffic.json'), 'w'), indent=1)
            print(json.dumps(out))
# round 4: the LDS counters of K1 on the second input distribution (uniform random bytes) next to the synthetic noise
uni = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, 'uniform', 'pmc_sq_b', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        uni[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
if uni:
    json.dump({k: {c: {'mean': sum(v) / len(v), 'launches': len(v)} for c, v in cs.items()} for k, cs in uni.items()},
              open(os.path.join(dst, 'pmc_summary_cfg2_uniform.json'), 'w'), indent=1, sort_keys=True)
for p in glob.glob(os.path.join(src, 'bench_*.log')):
    lines = [l for l in open(p) if l.startswith('{')]
    if lines:
        open(os.path.join(dst, os.path.basename(p).replace('.log', '.json')), 'w').write(lines[-1])
for name in ('fresh_runs.txt', 'single_block.txt', 'timeline_cfg2.txt', 'timeline_cfg5.txt', 'bench_gpus2_refusal.txt'):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
for w in ('cfg2', 'cfg3', 'cfg5'):
    for suffix in ('', '_depth1'):
        p = os.path.join(dst, f'kernel_stats_{w}{suffix}.csv')
        if os.path.exists(p):
            print('==', os.path.basename(p))
            for r in list(csv.DictReader(open(p)))[:5]:
                print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
