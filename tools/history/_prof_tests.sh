cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_tests
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tests -- python -m pytest tests -m gpu -x -q > gpurun_out/pytest_under_rocprof.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_under_rocprof.log | tail -2
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*kernel_trace.csv" -delete
find gpurun_out/prof_tests -name "*kernel_stats.csv"
