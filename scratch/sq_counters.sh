#!/bin/bash
# SQ counters of the bench kernels (one pass, 8 SQ slots): wave cycles, VALU / LDS activity, wait buckets.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O -o run -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --presettle 150 > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os, json, collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/sq'
f=glob.glob(O+'/*counter_collection.csv')
acc=collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    acc[(row['Kernel_Name'].split('(')[0], row['Counter_Name'])].append(float(row['Counter_Value']))
res={}
for (k,c),v in acc.items(): res.setdefault(k,{})[c]=sum(v)/len(v)
json.dump(res, open(O+'/sq_summary.json','w'), indent=1)
for k in ('k_raster','k_shade','k_solve','k_collide'):
    r=res.get(k,{})
    if not r: continue
    wc=r.get('SQ_WAVE_CYCLES',0) or 1
    print(k, {c: round(x) for c,x in r.items()})
    print('   VALU active / wave cycles %.3f  LDS active / wave cycles %.3f  wait_any %.3f  wait_inst_any %.3f  VALU insts per wave-cycle x4 %.3f' % (
        r.get('SQ_ACTIVE_INST_VALU',0)/wc, r.get('SQ_ACTIVE_INST_LDS',0)/wc, r.get('SQ_WAIT_ANY',0)/wc, r.get('SQ_WAIT_INST_ANY',0)/wc, 4*r.get('SQ_INSTS_VALU',0)/wc))
PY
