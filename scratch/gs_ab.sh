#!/bin/bash
# A/B of the solver's phase clock on the costliest solver workgroup of the headline workload: scratch/gs_ab.sh <steps>
T=${1:-300}
cd $GRAFT_REPO_ROOT
for lib in scratch/variants/old_stats.so real_robots_amd/csrc/librealrobot_hip_stats.so; do
  echo "=== $lib"
  RR_LIB=$PWD/$lib python scratch/solve_blocks.py x 1.0 $T > /tmp/sb.txt 2>&1; head -8 /tmp/sb.txt
  WG=$(grep -m1 "  wg " /tmp/sb.txt | awk '{print $2}')
  RR_LIB=$PWD/$lib RR_ABLATE=$(( (WG << 16) | 16384 )) python scratch/sprof.py 1.0 $T 2>&1 | tail -14
done
