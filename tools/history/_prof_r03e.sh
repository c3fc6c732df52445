cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03d; mkdir -p $O
P="rocprofv3 --output-format csv"
rm -rf $O/flow_fetch $O/flow_write $O/flow_sq1
python tools/flow_probe.py --iters 2 2>&1 | tail -1
$P --kernel-trace --stats -d $O/flow_stats -- python tools/flow_probe.py --utts 4 --iters 3 > $O/flow_stats.log 2>&1
$P --pmc FETCH_SIZE -d $O/flow_fetch -- python tools/flow_probe.py --utts 4 --iters 1 > $O/flow_fetch.log 2>&1
$P --pmc WRITE_SIZE -d $O/flow_write -- python tools/flow_probe.py --utts 4 --iters 1 > $O/flow_write.log 2>&1
$P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/flow_sq1 -- python tools/flow_probe.py --utts 4 --iters 1 > $O/flow_sq1.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
head -6 $(find $O/flow_stats -name "*kernel_stats.csv") | cut -c1-150
timeout 700 python -m pytest tests -m gpu -x -q --timeout 300 -k "flow or dit or estimator or cfm or cv3 or gemm or linear" 2>&1 | tail -3
