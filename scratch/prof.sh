#!/bin/bash
# usage: scratch/prof.sh <tag> [bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o run -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$TAG/bench.log 2>&1
tail -2 $R/gpurun_out/$TAG/bench.log
find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1 | xargs cat | head -20
