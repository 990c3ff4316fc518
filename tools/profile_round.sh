#!/bin/bash
# Collects a round's evidence on a GPU box (run through gpurun): the bench JSON (with cpu_baseline + secondary workloads),
# rocprofv3 kernel stats, PMC HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, never combined with other trace
# domains), SQ VALU counters.  Output: gpurun_out/round/ ; traffic_latest.json and sq_latest.json carry the run
# configuration so that bench.py only attaches them to runs of that configuration.  Usage: tools/profile_round.sh [tag]
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O
cd $R && python bench.py --steps 200 --warmup 20 > $O/${TAG}_bench.json 2> $O/bench.err; tail -c 600 $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $B --steps 100 --warmup 10 > $O/stats_bench.log 2>&1
cp $O/stats/run_kernel_stats.csv $O/${TAG}_kernel_stats.csv; head -12 $O/${TAG}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- $B --steps 20 --warmup 5 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- $B --steps 20 --warmup 5 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/sq -o run -- $B --steps 20 --warmup 5 > $O/sq.log 2>&1
TAG=$TAG python3 - <<'PY'
import csv, glob, os, json, collections
R=os.environ['GRAFT_REPO_ROOT']; O=R+'/gpurun_out/round'; TAG=os.environ['TAG']
cfg={"envs": 4096, "objects": 3, "width": 128, "height": 128, "render": True, "command_scale": 1.0, "solver_iters": 50}
import importlib.util
_sp=importlib.util.spec_from_file_location('bench_mod', R+'/bench.py'); _bm=importlib.util.module_from_spec(_sp); _sp.loader.exec_module(_bm)
SRC_SHA=_bm.kernel_source_hash()   # bench.load_profile only quotes a profile of the very kernel sources it runs
def means(d):
    """mean per dispatch and, for kernels launched more than once per step, the number of dispatches per step"""
    f=glob.glob(O+'/%s/*counter_collection.csv'%d)
    acc=collections.defaultdict(list)
    if not f: print('no counter csv in', d); return {}
    for row in csv.DictReader(open(f[0])):
        acc[(row['Kernel_Name'].split('(')[0], row['Counter_Name'])].append(float(row['Counter_Value']))
    steps=max([len(v) for (k,c),v in acc.items() if k=='k_raster'] or [1])
    out={}
    for (k,c),v in acc.items(): out.setdefault(k,{})[c]={'mean': sum(v)/len(v), 'n': len(v), 'per_step': sum(v)/steps}
    return out
fe, wr, sq = means('pmc_fetch'), means('pmc_write'), means('sq')
json.dump({'fetch': fe, 'write': wr}, open(O+'/%s_pmc_summary.json'%TAG,'w'), indent=1)
json.dump(sq, open(O+'/%s_sq_counters.json'%TAG,'w'), indent=1)
# HBM bytes per launch: (2*FETCH_SIZE + WRITE_SIZE) KB (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts half of wide coalesced reads)
tr={}; lo={}
for k in set(fe)|set(wr):
    # bytes per STEP (a kernel launched for the light and again for the heavy envs counts with all its launches)
    f=fe.get(k,{}).get('FETCH_SIZE',{}).get('per_step',0.0); w=wr.get(k,{}).get('WRITE_SIZE',{}).get('per_step',0.0)
    tr[k]=round((2*f+w)*1024); lo[k]=round((f+w)*1024)
for d in (tr, lo):
    d['render_stage']=d.get('k_raster',0)+d.get('k_shade',0)+d.get('k_render_list',0)+d.get('k_raster_list',0)+d.get('k_render_setup',0)
    d['k_prep']=sum(d.get(k,0) for k in ('k_prep_a','k_prep_b','k_prep_ab','k_prep_a16','k_prep_b16','k_prep_ab16'))
    d['k_solve']=d.get('k_solve',0)+d.get('k_solve_rs',0)      # (k_solve_rs: the heavy classes' solve of a step that draws -- the same kernel with the render set-up in its tail)
tr['lower_bound']=lo
tr['config']=cfg; tr['source']=TAG+'_pmc_summary.json'; tr['source_sha256']=SRC_SHA
tr['_note']="HBM bytes per step from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), all launches of a kernel in a step added up (the first untimed frames included in the mean). Top level: (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- the guide's gfx950 correction, which is calibrated for 16-byte-per-lane coalesced streaming reads only (here: k_solve's contact records and row stream, k_static_copy); `lower_bound`: (FETCH_SIZE + WRITE_SIZE) * 1024, no correction -- for kernels whose reads are scattered 4/8/16-byte records (k_raster, k_shade, k_collide, k_prep_*) the truth lies between the two. render_stage = k_render_setup + k_raster + k_shade + the heavy envs' k_render_list / k_raster_list; only valid for `config`"
json.dump(tr, open(O+'/traffic_latest.json','w'), indent=1)
sv={'config': cfg, 'source': TAG+'_sq_counters.json', 'source_sha256': SRC_SHA, 'valu_wave_instr_per_launch': {k: round(v['SQ_INSTS_VALU']['mean']) for k,v in sq.items() if 'SQ_INSTS_VALU' in v}}
json.dump(sv, open(O+'/sq_latest.json','w'), indent=1)
for k in ('k_raster','k_shade','k_solve','k_solve_rs','k_collide','k_prep_a','k_prep_b','k_prep_ab','k_prep_a16','k_prep_b16','k_prep_ab16','k_solve_light','k_solve_light_ow','k_render_setup','k_render_list'):
    r={c: x['mean'] for c,x in sq.get(k,{}).items()}
    if not r: continue
    wc=r.get('SQ_WAVE_CYCLES',0) or 1
    print(k, 'traffic MB %.1f'%(tr.get(k,0)/1e6), 'VALU wave-instr %.3g'%r.get('SQ_INSTS_VALU',0), 'VALU active/wave-cycle %.3f wait_any %.3f'%(r.get('SQ_ACTIVE_INST_VALU',0)/wc, r.get('SQ_WAIT_ANY',0)/wc))
PY
