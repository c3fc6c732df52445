# round-4 profiling passes (one gpurun call).  Kernel-trace stats and PMC passes are SEPARATE rocprofv3 runs.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04c; rm -rf $O; mkdir -p $O
P="rocprofv3 --output-format csv"
# new tests of the round
true
# --- decode step (64 sequences x 2 heads, ctx 1536): per-kernel stats, HBM traffic, SQ counters ---
$P --kernel-trace --stats -d $O/dec_stats -- python tools/bench_decode.py --seqs 64 --steps 100 > $O/dec_stats.log 2>&1
$P --pmc FETCH_SIZE -d $O/dec_fetch -- python tools/bench_decode.py --seqs 64 --steps 10 > $O/dec_fetch.log 2>&1
$P --pmc WRITE_SIZE -d $O/dec_write -- python tools/bench_decode.py --seqs 64 --steps 10 > $O/dec_write.log 2>&1
$P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/dec_sq1 -- python tools/bench_decode.py --seqs 64 --steps 10 > $O/dec_sq1.log 2>&1
# --- the bench command under the kernel trace ---
timeout 900 $P --kernel-trace --stats -d $O/bench_stats -- python bench.py --steps 8 --no-cpu-baseline --no-fp32-mode > $O/bench_under_rocprof.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; tail -1 $O/dec_stats.log | cut -c1-200; tail -1 $O/bench_under_rocprof.log | cut -c1-160
