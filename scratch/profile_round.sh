#!/bin/bash
# Collects the round's evidence: bench JSON (with cpu_baseline), rocprofv3 kernel stats, PMC traffic of the dominant kernel.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O
cd $R && python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 > $O/stats_bench.log 2>&1
cat $O/stats/run_kernel_stats.csv | head -8
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --presettle 150 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --presettle 150 > $O/pmc_write.log 2>&1
ls $O/pmc_fetch $O/pmc_write
python3 - <<'PY'
import csv, glob, os, json, collections
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/round'
res={}
for name in ('fetch','write'):
    f=glob.glob(O+'/pmc_%s/*counter_collection.csv'%name)
    if not f: print('no counter csv for',name); continue
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        acc[(row['Kernel_Name'].split('(')[0], row['Counter_Name'])].append(float(row['Counter_Value']))
    for k,v in acc.items():
        res['%s:%s'%k]=dict(mean=sum(v)/len(v), n=len(v))
        print(k, 'mean', sum(v)/len(v), 'n', len(v))
json.dump(res, open(O+'/pmc_summary.json','w'), indent=1)
PY
