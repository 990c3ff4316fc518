#!/bin/bash
python scratch/cullcheck.py
python -m pytest tests/test_gpu_parity.py -q -m gpu 2>&1 | grep -E "^E  |passed|failed" | head -20
run() { echo "$*: $(python bench.py --no-cpu-baseline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})")"; }
run
