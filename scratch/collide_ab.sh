#!/bin/bash
# scratch/collide_ab.sh: collision pass timeline + phase clock (stats build) and a short headline A/B against scratch/variants/head.so
cd $GRAFT_REPO_ROOT
python scratch/cwgtime.py 2>&1 | grep -v amdgpu.ids | grep -A1 "rep [02]"
python scratch/cprof.py 2>&1 | grep -v amdgpu.ids
for lib in scratch/variants/head.so real_robots_amd/csrc/librealrobot_hip.so scratch/variants/head.so real_robots_amd/csrc/librealrobot_hip.so; do
  RR_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('== $lib headline %.4f ms %.3f M | heavy solves %.3f light %.4f collide %.4f prep %.4f raster %.4f shade %.4f' % (d['ms_per_step'], d['value']/1e6, k['k_solve_heavy']['avg_ms'], k['k_solve']['avg_ms'], k['k_collide']['avg_ms'], k['k_prep']['avg_ms'], k['k_raster']['avg_ms'], k['k_shade']['avg_ms']))"
done
