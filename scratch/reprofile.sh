set -x
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r06 > gpurun_out/round_log.txt 2>&1
O=gpurun_out/round
cp $O/traffic_latest.json $O/sq_latest.json profiles/
bash scratch/prof_prep.sh > gpurun_out/round/r06_prep_forms.txt 2>&1
python scratch/timeline.py gpurun_out/round/stats 30 > gpurun_out/round/r06_timeline.txt 2>&1
python bench.py --steps 200 --warmup 20 > gpurun_out/round/r06_bench.json 2> gpurun_out/round/bench2.err
tail -c 300 gpurun_out/round/r06_bench.json
python - <<'PY'
import json, bench
d = json.loads(open('gpurun_out/round/r06_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'traffic', d['roofline'].get('traffic'), 'valu frac', d['roofline'].get('valu', {}).get('frac'))
print('hash', bench.kernel_source_hash() == json.load(open('profiles/traffic_latest.json'))['source_sha256'])
print({k: (v.get('value') if isinstance(v, dict) else v) for k, v in d.get('secondary', {}).items()} if isinstance(d.get('secondary'), dict) else type(d.get('secondary')))
PY
