# round-3 profiling passes (one gpurun call).  Kernel-trace stats and PMC passes are SEPARATE rocprofv3 runs.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03; mkdir -p $O
P="rocprofv3 --output-format csv"
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*\|FETCH_SIZE\|WRITE_SIZE" | sort -u > $O/counters_available.txt
# --- flow (gemm_big, attn_dit) ---
$P --kernel-trace --stats -d $O/flow_stats -- python tools/flow_probe.py --utts 4 --iters 2 > $O/flow_stats.log 2>&1
$P --pmc FETCH_SIZE -d $O/flow_fetch -- python tools/flow_probe.py --utts 4 --iters 1 > $O/flow_fetch.log 2>&1
$P --pmc WRITE_SIZE -d $O/flow_write -- python tools/flow_probe.py --utts 4 --iters 1 > $O/flow_write.log 2>&1
$P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/flow_sq1 -- python tools/flow_probe.py --utts 4 --iters 1 > $O/flow_sq1.log 2>&1
$P --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $O/flow_sq2 -- python tools/flow_probe.py --utts 4 --iters 1 > $O/flow_sq2.log 2>&1
# --- decode step (64 sequences x 2 heads, ctx 1536) ---
$P --kernel-trace --stats -d $O/dec_stats -- python tools/bench_decode.py --seqs 64 --steps 100 > $O/dec_stats.log 2>&1
$P --pmc FETCH_SIZE -d $O/dec_fetch -- python tools/bench_decode.py --seqs 64 --steps 10 > $O/dec_fetch.log 2>&1
$P --pmc WRITE_SIZE -d $O/dec_write -- python tools/bench_decode.py --seqs 64 --steps 10 > $O/dec_write.log 2>&1
# --- vocoder ---
$P --kernel-trace --stats -d $O/hift_stats -- python tools/hift_probe.py --iters 3 > $O/hift_stats.log 2>&1
$P --pmc FETCH_SIZE -d $O/hift_fetch -- python tools/hift_probe.py --iters 1 > $O/hift_fetch.log 2>&1
$P --pmc WRITE_SIZE -d $O/hift_write -- python tools/hift_probe.py --iters 1 > $O/hift_write.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; find $O -name "*.csv" | head -40; tail -2 $O/*.log | cut -c1-300
