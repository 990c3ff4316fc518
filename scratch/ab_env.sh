#!/bin/bash
# A/B of environment knobs on one box: scratch/ab_env.sh "VAR=val VAR2=val" ["..." ...]  ("-" = defaults); BENCH_ARGS="--command-scale 0.5" for another workload
cd $GRAFT_REPO_ROOT
for kv in "$@"; do
  for rep in 1 2; do
  ( if [ "$kv" != "-" ]; then export $kv; fi
    python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 20 $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('%-40s %s %.4f ms | collide %.4f prep %.4f raster %.4f shade %.4f heavy solves %.3f' % ('$kv', '$BENCH_ARGS' or 'headline', d['ms_per_step'], k['k_collide']['avg_ms'], k['k_prep']['avg_ms'], k['k_raster']['avg_ms'], k['k_shade']['avg_ms'], k['k_solve_heavy']['avg_ms']))" )
  done
done
