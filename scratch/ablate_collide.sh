#!/bin/bash
# k_collide phase ablations (development build: make -C real_robots_amd/csrc stats); physics is wrong under ablation, only time matters
export RR_LIB=$PWD/real_robots_amd/csrc/librealrobot_hip_stats.so
for a in 2048; do
  echo "ablate=$a: $(RR_ABLATE=$a python bench.py --no-cpu-baseline --no-render --steps 30 --warmup 5 --presettle 150 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})")"
done
