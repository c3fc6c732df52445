set -x
mkdir -p gpurun_out/prof_bench gpurun_out/prof_flow gpurun_out/prof_dec
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2
timeout 250 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_under_rocprof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_flow -- python tools/flow_probe.py --iters 1 > gpurun_out/flow_probe_rocprof.log 2>&1
find gpurun_out/prof_bench gpurun_out/prof_flow -name "*kernel_stats.csv" | head
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*kernel_trace.csv" -delete
du -sh gpurun_out
