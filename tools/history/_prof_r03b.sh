cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03b; mkdir -p $O
P="rocprofv3 --output-format csv"
timeout 600 python -m pytest tests/test_gpu_matcha.py -m gpu -x -q --timeout 300 2>&1 | tail -2
python tools/bench_matcha.py --decoder cv2 > $O/bench_matcha_cv2.log 2>&1; tail -1 $O/bench_matcha_cv2.log | cut -c1-600
python tools/bench_matcha.py --decoder matcha > $O/bench_matcha.log 2>&1; tail -1 $O/bench_matcha.log | cut -c1-300
$P --kernel-trace --stats -d $O/matcha_stats -- python tools/bench_matcha.py --decoder cv2 --steps 3 > $O/matcha_stats.log 2>&1
$P --kernel-trace --stats -d $O/hift_stats -- python tools/hift_probe.py --iters 3 > $O/hift_stats.log 2>&1
$P --pmc FETCH_SIZE -d $O/hift_fetch -- python tools/hift_probe.py --iters 1 > $O/hift_fetch.log 2>&1
$P --pmc WRITE_SIZE -d $O/hift_write -- python tools/hift_probe.py --iters 1 > $O/hift_write.log 2>&1
$P --pmc FETCH_SIZE -d $O/attn_fetch -- python tools/attn_probe.py 8 > $O/attn_fetch.log 2>&1
$P --pmc WRITE_SIZE -d $O/attn_write -- python tools/attn_probe.py 8 > $O/attn_write.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
