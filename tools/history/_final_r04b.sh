# round-4 record, final build, second half (one gpurun call): decode-step stats + PMC traffic (tools/_prof_r04d.sh), the bench command under the kernel trace
bash tools/_prof_r04d.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04e; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/bench_stats -- python bench.py --steps 8 --no-cpu-baseline --no-fp32-mode > $O/bench_under_rocprof.log 2>&1
cp $(find $O/bench_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_8steps.csv; rm -rf $O/bench_stats
head -8 $O/kernel_stats_bench_8steps.csv | cut -c1-150; tail -1 $O/bench_under_rocprof.log | cut -c1-160
