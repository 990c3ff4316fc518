#!/bin/bash
# phase clock (cycles per sweep) of ONE env in the one-env-per-wave form (1024 envs): scratch/gs_env.sh <env multiple of 4> <T>
cd $GRAFT_REPO_ROOT
E=${1:-776}; T=${2:-400}
for lib in scratch/variants/old_stats.so real_robots_amd/csrc/librealrobot_hip_stats.so; do
  echo "=== $lib env $E"
  N=1024 RR_LIB=$PWD/$lib RR_ABLATE=$(( ((E / 4) << 16) | 16384 )) python scratch/sprof.py 1.0 $T per_sweep 2>&1 | grep -v amdgpu | tail -14 | head -10
done
