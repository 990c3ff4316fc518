#!/bin/bash
# k_raster phase ablations (development build: make -C real_robots_amd/csrc stats)
export RR_LIB=$PWD/real_robots_amd/csrc/librealrobot_hip_stats.so
for a in ${ABL:-0 1 2 3 8}; do
  echo "ablate=$a: $(RR_ABLATE=$a python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items() if k=='k_raster'})")"
done
