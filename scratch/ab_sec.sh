#!/bin/bash
# A/B of library variants on the bench headline AND its secondary workloads (late window, macro actions, ...: the list kernels)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/var
P='import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"]), d["ms_per_step"]); [print("   ", s.get("workload", s.get("name")), s.get("value"), s.get("ms_per_step")) for s in d.get("secondary", [])]'
for f in "" scratch/variants/lib_*.so; do
  echo "== ${f:-shipped}"
  if [ -n "$f" ]; then export RR_LIB=$PWD/$f; else unset RR_LIB; fi
  timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/var/bench_$(basename ${f:-shipped}).json | python -c "$P"
done
