#!/bin/bash
cd $GRAFT_REPO_ROOT
lib=real_robots_amd/csrc/librealrobot_hip_stats.so
for E in 776 768 524 396; do
  echo "=== new env $E"
  N=1024 RR_LIB=$PWD/$lib RR_ABLATE=$(( ((E / 4) << 16) | 16384 )) python scratch/sprof.py 1.0 400 per_sweep 2>&1 | grep -v amdgpu | tail -14 | head -10
done
scratch/gs_one.sh $lib 300 | grep -v amdgpu
