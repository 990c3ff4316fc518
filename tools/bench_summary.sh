#!/bin/bash
# bench.py without the CPU baseline, summarised: value, ms/step, per-kernel ms of the headline and the secondary workloads
mkdir -p gpurun_out/bs
python bench.py --steps ${1:-100} --warmup 10 --no-cpu-baseline > gpurun_out/bs/bench.json 2> gpurun_out/bs/bench.err || { tail -5 gpurun_out/bs/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/bs/bench.json"))
print("headline %.0f env-steps/s %.4f ms/step" % (d["value"], d["ms_per_step"]), {k: v["avg_ms"] for k, v in d["roofline"]["kernels"].items()})
for s in d.get("secondary", []):
    print("  %.0f env-steps/s %.4f ms/step" % (s["value"], s["ms_per_step"]), s["kernels_ms"], "|", s["workload"][:40])
PY
