#!/bin/bash
# kernel durations of the two preparation forms under rocprofv3 (headline workload + config 2 script)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_prep; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for form in 0 1; do
  export RR_PREP_SCALAR=$form
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/h$form -o run -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > $O/h$form.log 2>&1
  echo "== headline, RR_PREP_SCALAR=$form"; grep -E "k_prep|k_collide|k_solve_light" $O/h$form/run_kernel_stats.csv | cut -d, -f1-8
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/c$form -o run -- python3 $R/scratch/config2.py > $O/c$form.log 2>&1
  echo "== config2 script, RR_PREP_SCALAR=$form"; grep -E "k_prep|k_collide|k_solve" $O/c$form/run_kernel_stats.csv | cut -d, -f1-8
done
