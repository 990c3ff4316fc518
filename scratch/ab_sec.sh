#!/bin/bash
# scratch/ab_sec.sh <lib> ...: A/B over the solver-bound workloads (config 2 script, then bench with secondaries)
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  echo "== $lib"
  RR_LIB=$PWD/$lib python scratch/config2.py 2>&1 | grep "^N 1024"
  RR_LIB=$PWD/$lib python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('   headline %.4f ms %.3f M | heavy solves %.3f light %.4f collide %.4f' % (d['ms_per_step'], d['value']/1e6, k['k_solve_heavy']['avg_ms'], k['k_solve']['avg_ms'], k['k_collide']['avg_ms']))
for s in d['secondary']:
    if s.get('value'): print('   %-58s %.3f M %s ms' % (s['workload'][:58], s['value']/1e6, s.get('ms_per_step')))
"
done
