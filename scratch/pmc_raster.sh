#!/bin/bash
# PMC passes over k_raster alone (scratch/render_only.py); output gpurun_out/ctr/<pass>/ ; never combined with trace domains other than --kernel-trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ctr; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/list.txt 2>&1
grep -oE "\b(SQ|SQC|TCP|TCC|TA|TD|GRBM|SPI|CPC)_[A-Z0-9_]+" $O/list.txt | sort -u > $O/names.txt; wc -l $O/names.txt
i=0
while read -r line; do
  i=$((i+1))
  timeout 170 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $O/p$i -o run -- python3 $R/scratch/render_only.py 12 > $O/p$i.log 2>&1
  echo "pass $i rc=$? : $line"; tail -2 $O/p$i.log
done <<'PASSES'
SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_LDS
SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH SQ_WAIT_ANY
SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAVES_EQ_64 SQ_ACTIVE_INST_VMEM
PASSES
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ctr'
for d in sorted(glob.glob(O+'/p*/')):
    f=glob.glob(d+'*counter_collection.csv')
    if not f: print(d, 'no csv'); continue
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        k=row['Kernel_Name'].split('(')[0]
        if k in ('k_raster','k_shade'): acc[(k,row['Counter_Name'])].append(float(row['Counter_Value']))
    for (k,c),v in sorted(acc.items()): print('%-10s %-28s n=%3d mean=%.4g'%(k,c,len(v),sum(v[-10:])/len(v[-10:])))
PY
