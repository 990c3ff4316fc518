#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raster or full_size or flags" 2>&1 | tail -3
for a in 0 1 2 4 3; do
  echo "ablate=$a: $(RR_ABLATE=$a python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})")"
done
echo "no static: $(RR_NO_STATIC_LAYER=1 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})")"
