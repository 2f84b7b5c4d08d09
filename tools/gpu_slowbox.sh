#!/bin/bash
# what is slow on a "slow box"?  K1 alone (synchronous decode), K1 in the pipelined bench with the full read-back
# (290 k hits = 7 MB per step), and with the validated read-back (4 k hits)
cd $GRAFT_REPO_ROOT
python tools/placement_probe.py 2 2>/dev/null | grep "round 0" | cut -c1-110
one() { tag=$1; shift; "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$tag', j['value'], 'steady', j['steady_ms_per_step'], 'k1', r['k1_ms'], r['frac'], 'k2', r['search_ms'])"; }
one "pipelined raw      " python bench.py --no-cpu-baseline --steps 100
one "pipelined validated" python bench.py --no-cpu-baseline --steps 100 --validate
one "pipelined raw      " python bench.py --no-cpu-baseline --steps 100
one "depth 1            " python bench.py --no-cpu-baseline --steps 50 --depth 1
one "depth 2            " python bench.py --no-cpu-baseline --steps 100 --depth 2
