#!/bin/bash
# SQ counters of the acoustic stage's kernels on the final build: matrix-pipe busy, issue-active, wait fractions (separate rocprofv3 --pmc passes, never with a trace);
# durations from a --kernel-trace --stats pass of the same probes in the same call.  -> gpurun_out/pmc_sq_r05/pmc_sq.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_sq_r05; rm -rf $O; mkdir -p $O
P="rocprofv3 --output-format csv"
for w in flow hift; do
  case $w in flow) C="python tools/flow_probe.py --utts 4 --iters 1";; hift) C="python tools/hift_probe.py --iters 1";; esac
  timeout 600 $P --kernel-trace --stats -d $O/${w}_stats -- $C > $O/${w}_stats.log 2>&1
  timeout 600 $P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/${w}_sq1 -- $C > $O/${w}_sq1.log 2>&1
  timeout 600 $P --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/${w}_sq2 -- $C > $O/${w}_sq2.log 2>&1
done
python - <<PY
import csv, glob, json, collections
O='$O'
out={'_how': 'tools/pmc_sq_r05.sh: rocprofv3 --pmc in two passes per probe (tools/flow_probe.py --utts 4, tools/hift_probe.py), durations from a separate --kernel-trace --stats pass of the same call; per kernel: counter means per dispatch; derived: mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs) (nominal clock: the chip runs lower under load), the wave-cycle fractions = counter / SQ_WAVE_CYCLES', 'kernels': {}}
for w in ('flow','hift'):
    dur={}
    for f in glob.glob(O+'/%s_stats/**/*kernel_stats.csv' % w, recursive=True):
        for r in csv.DictReader(open(f)): dur[r['Name']]=float(r['AverageNs'])/1e3
    agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
    for p in ('sq1','sq2'):
        for f in glob.glob(O+'/%s_%s/**/*counter_collection.csv' % (w,p), recursive=True):
            rd=csv.DictReader(open(f)); cols={c.lower():c for c in rd.fieldnames}
            for r in rd:
                a=agg[r[cols['kernel_name']]][r[cols['counter_name']]]; a[0]+=float(r[cols['counter_value']]); a[1]+=1
    for k,cs in agg.items():
        if 'hvx' not in k: continue
        d={c: round(v[0]/v[1],1) for c,v in cs.items()}
        d['dispatches']=max(v[1] for v in cs.values())
        us=dur.get(k)
        der={}
        if us: der['avg_duration_us']=round(us,1)
        wc=d.get('SQ_WAVE_CYCLES')
        if us and 'SQ_VALU_MFMA_BUSY_CYCLES' in d: der['mfma_busy_frac_at_2p4GHz']=round(d['SQ_VALU_MFMA_BUSY_CYCLES']/(us*2400*1024),3)
        if wc:
            for c,n in (('SQ_ACTIVE_INST_ANY','issue_active_frac'),('SQ_WAIT_INST_ANY','wait_inst_any_frac'),('SQ_WAIT_ANY','wait_any_frac'),('SQ_ACTIVE_INST_VALU','valu_active_frac'),('SQ_ACTIVE_INST_LDS','lds_active_frac')):
                if c in d: der[n]=round(d[c]/wc,3)
        if d.get('SQ_LDS_IDX_ACTIVE'): der['lds_bank_conflict_frac_of_lds_cycles']=round(d.get('SQ_LDS_BANK_CONFLICT',0)/d['SQ_LDS_IDX_ACTIVE'],3)
        d['derived']=der
        if d['dispatches']>=8: out['kernels'][k]=d
json.dump(out, open(O+'/pmc_sq.json','w'), indent=1)
for k,v in sorted(out['kernels'].items(), key=lambda kv: -kv[1]['derived'].get('avg_duration_us',0)*kv[1]['dispatches'])[:8]:
    print(k[:70], v['dispatches'], v['derived'])
PY
rm -rf $O/*_stats $O/*_sq1 $O/*_sq2
