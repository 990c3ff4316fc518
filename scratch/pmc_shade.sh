#!/bin/bash
# PMC passes for k_shade (and k_raster beside it): cache hits / misses, VMEM instructions, LDS / VMEM waits.  Separate passes, kernel trace only.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/pmc_shade; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5"
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -o run -- $B > $O/$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r05/pmc_shade'
acc=collections.defaultdict(list)
for f in glob.glob(O+'/*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'].split('(')[0]
        if k in ('k_shade','k_raster','k_collide'): acc[(k,row['Counter_Name'])].append(float(row['Counter_Value']))
for (k,c),v in sorted(acc.items()): print('%-10s %-34s mean per launch %.4g  (n %d)' % (k,c,sum(v)/len(v),len(v)))
PY
