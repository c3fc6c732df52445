cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03c; mkdir -p $O
P="rocprofv3 --output-format csv"
$P --kernel-trace --stats -d $O/flow_half -- python tools/flow_probe.py --iters 3 > $O/flow_half.log 2>&1
HVX_FLOW_HALF_STREAM=0 $P --kernel-trace --stats -d $O/flow_f32 -- python tools/flow_probe.py --iters 3 > $O/flow_f32.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for d in flow_half flow_f32; do echo $d; tail -1 $O/$d.log; f=$(find $O/$d -name "*kernel_stats.csv" | head -1); head -7 $f | cut -c1-150; done
