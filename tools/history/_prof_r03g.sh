cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03g; rm -rf $O; mkdir -p $O
P="rocprofv3 --output-format csv"
$P --kernel-trace --stats -d $O/a -- python tools/flow_probe.py --utts 4 --iters 2 > $O/a.log 2>&1
HVX_FLOW_F16_LINEARS=1 $P --kernel-trace --stats -d $O/b -- python tools/flow_probe.py --utts 4 --iters 2 > $O/b.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for d in a b; do echo $d; tail -1 $O/$d.log; head -7 $(find $O/$d -name "*kernel_stats.csv") | cut -c1-165; done
