#!/bin/bash
# A/B of library variants: bench headline (+ optional args) for each scratch/variants/lib_*.so and the shipped library
mkdir -p gpurun_out/var
P='import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"]), d["ms_per_step"], {k: v["avg_ms"] for k, v in d["roofline"]["kernels"].items()})'
for f in scratch/variants/lib_*.so; do
  echo "== $f"; RR_LIB=$PWD/$f python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary "$@" 2>gpurun_out/var/err.log | python -c "$P"
done
echo "== shipped"; python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "$P"
