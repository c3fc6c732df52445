# round 3, after the fp16 residual stream / register-pipelined K-loop / resident-row conv: kernel stats and PMC passes (separate runs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03d; mkdir -p $O
P="rocprofv3 --output-format csv"
$P --pmc FETCH_SIZE -d $O/flow_fetch -- python tools/flow_probe.py --utts 4 --iters 1 > $O/flow_fetch.log 2>&1
$P --pmc WRITE_SIZE -d $O/flow_write -- python tools/flow_probe.py --utts 4 --iters 1 > $O/flow_write.log 2>&1
$P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/flow_sq1 -- python tools/flow_probe.py --utts 4 --iters 1 > $O/flow_sq1.log 2>&1
$P --kernel-trace --stats -d $O/hift_stats -- python tools/hift_probe.py --iters 3 > $O/hift_stats.log 2>&1
$P --pmc FETCH_SIZE -d $O/hift_fetch -- python tools/hift_probe.py --iters 1 > $O/hift_fetch.log 2>&1
$P --pmc WRITE_SIZE -d $O/hift_write -- python tools/hift_probe.py --iters 1 > $O/hift_write.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; tail -1 $O/*.log | cut -c1-200
