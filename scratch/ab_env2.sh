#!/bin/bash
# scratch/ab_env2.sh VAR=VAL ...: headline + secondaries with and without an environment setting, alternating, same box
cd $GRAFT_REPO_ROOT
run() {
  python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  headline %.4f ms' % d['ms_per_step'], ' | '.join('%s %.4f' % (s['workload'][:22], s['ms_per_step']) for s in d['secondary'] if s.get('ms_per_step')))"
}
for i in 1 2; do
  echo "default"; run
  echo "$@"; env "$@" bash -c "$(declare -f run); run"
done
