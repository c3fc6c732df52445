cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04i; rm -rf $O; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace -d $O/tr -- python tools/bench_decode.py --seqs 64 --steps 20 > $O/dec.log 2>&1
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find the last log_softmax, print the launches between the previous log_softmax and it, after the last layer
idx=[i for i,r in enumerate(rows) if 'log_softmax' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
seg=rows[a+1:b+1]
t0=int(seg[0]['Start_Timestamp'])
print('step launches',len(seg),'span us',(int(seg[-1]['End_Timestamp'])-t0)/1e3)
for r in seg[-22:]:
    print('%8.1f %7.1f  %s' % ((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r['Kernel_Name'][:100]))
print('-- first layer')
for r in seg[:10]:
    print('%8.1f %7.1f  %s' % ((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r['Kernel_Name'][:100]))
PY
rm -rf $O/tr
timeout 1200 python -m pytest tests/test_gpu_cv3w.py tests/test_gpu_fullsize.py tests/test_gpu_models.py -m gpu -x -q > gpurun_out/pytest_r04_i.log 2>&1; tail -3 gpurun_out/pytest_r04_i.log
