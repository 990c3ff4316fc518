#!/bin/bash
# phase clock of one solver workgroup (the costliest of the headline workload at step T) for one stats build: scratch/gs_one.sh <lib> <T>
cd $GRAFT_REPO_ROOT
lib=$1; T=${2:-300}
RR_LIB=$PWD/$lib python scratch/solve_blocks.py x 1.0 $T > /tmp/sb.txt 2>&1; grep -m3 "workgroups\|  wg" /tmp/sb.txt
WG=$(grep -m1 "  wg " /tmp/sb.txt | awk '{print $2}')
RR_LIB=$PWD/$lib RR_ABLATE=$(( (WG << 16) | 16384 )) python scratch/sprof.py 1.0 $T 2>&1 | tail -14
