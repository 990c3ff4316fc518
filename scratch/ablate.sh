#!/bin/bash
python -m pytest tests -q -m gpu -k "raster or camera or full_size or flags" 2>&1 | tail -1
for sa in 64 32; do
  echo "small_area=$sa: $(RR_SMALL_AREA=$sa python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items() if k=='k_raster'})")"
done
