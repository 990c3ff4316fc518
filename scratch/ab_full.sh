#!/bin/bash
# A/B of two library builds on one box over the bench's headline + secondary workloads: scratch/ab_full.sh <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  RR_LIB=$PWD/$lib python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('== $lib headline %.4f ms %.3f M | heavy solves %.3f light %.4f collide %.4f raster %.4f shade %.4f' % (d['ms_per_step'], d['value']/1e6, k['k_solve_heavy']['avg_ms'], k['k_solve']['avg_ms'], k['k_collide']['avg_ms'], k['k_raster']['avg_ms'], k['k_shade']['avg_ms']))
for s in d['secondary']:
    if s.get('value'): print('   %-58s %.3f M %s ms' % (s['workload'][:58], s['value']/1e6, s.get('ms_per_step')))
    else: print('   %-58s %s' % (s['workload'][:58], {k: v for k, v in s.items() if k in ('camera_off','camera_on')}))
"
done
