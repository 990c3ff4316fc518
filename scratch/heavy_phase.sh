#!/bin/bash
# scratch/heavy_phase.sh [T]: the costliest envs of the 1024-env workload at step T (one env per wave) and the phase clock of the sweep of
# the costliest ones that sit in wave 0 of their workgroup (the phase clock reads wave 0 of one workgroup)
cd $GRAFT_REPO_ROOT
T=${1:-400}
N=1024 python scratch/solve_blocks.py x 1.0 $T 2>&1 | grep -v amdgpu | head -14 > /tmp/sb.txt; cat /tmp/sb.txt
for U in $(grep "^  wg" /tmp/sb.txt | awk '{print $2}'); do
  if [ $((U % 4)) -eq 0 ]; then
    echo "=== unit $U"; grep "wg *$U " /tmp/sb.txt
    N=1024 RR_ABLATE=$(( ((U / 4) << 16) | 16384 )) python scratch/sprof.py 1.0 $T per_sweep 2>&1 | grep -v amdgpu | tail -14 | head -10
  fi
done
