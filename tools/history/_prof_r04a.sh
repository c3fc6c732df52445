# round-4: per-kernel stats of the 128-row decode step (gemm_dec form)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04a; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace --stats -d $O/dec_stats -- python tools/bench_decode.py --seqs 64 --steps 100 > $O/dec_stats.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1); cut -c1-200 $f | head -16; tail -1 $O/dec_stats.log
