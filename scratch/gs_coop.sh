#!/bin/bash
cd $GRAFT_REPO_ROOT
for lib in scratch/variants/old_stats.so real_robots_amd/csrc/librealrobot_hip_stats.so; do
  echo "=== $lib (1024 envs, one env per wave)"
  N=1024 RR_LIB=$PWD/$lib python scratch/solve_blocks.py x 1.0 400 2>&1 | grep -v amdgpu | head -9
done
