cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 400 > gpurun_out/final/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/final/pytest_gpu.log | tail -1
timeout 800 python bench.py --steps 20 > gpurun_out/final/bench.log 2>&1; tail -1 gpurun_out/final/bench.log | cut -c1-160
timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -- python bench.py --steps 8 --no-cpu-baseline --no-fp32-mode > gpurun_out/final/bench_under_rocprof.log 2>&1; tail -1 gpurun_out/final/bench_under_rocprof.log | cut -c1-160
find gpurun_out/final -name "*.db" -delete; find gpurun_out/final -name "*kernel_trace.csv" -delete; find gpurun_out/final -name "*agent_info.csv" -delete
