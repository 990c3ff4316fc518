#!/bin/bash
# A/B of two builds of the library on one box: headline step time and the per-kernel HIP-event times (k_solve_heavy = heavy + very heavy solves)
cd $GRAFT_REPO_ROOT
run() {
  python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('$1', d['value'], d['ms_per_step'], 'heavy solves', k['k_solve_heavy']['avg_ms'], 'light', k['k_solve']['avg_ms'], 'collide', k['k_collide']['avg_ms'], 'prep', k['k_prep']['avg_ms'])"
}
for lib in scratch/variants/old.so real_robots_amd/csrc/librealrobot_hip.so; do
  export RR_LIB=$PWD/$lib
  run "$lib coop"
  RR_NO_COOP=1 run "$lib NO_COOP"
done
