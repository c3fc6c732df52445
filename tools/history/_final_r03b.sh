cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 800 python bench.py > gpurun_out/final/bench.log 2>&1; tail -1 gpurun_out/final/bench.log | cut -c1-160
HVX_FLOW_F16_LINEARS=0 HVX_FLOW_F32_SMALL=0 timeout 500 python bench.py --no-cpu-baseline --no-fp32-mode > gpurun_out/final/bench_plain_bf16.log 2>&1; tail -1 gpurun_out/final/bench_plain_bf16.log | cut -c1-160
timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -- python bench.py --steps 8 --no-cpu-baseline --no-fp32-mode > gpurun_out/final/bench_under_rocprof.log 2>&1; tail -1 gpurun_out/final/bench_under_rocprof.log | cut -c1-120
find gpurun_out/final -name "*.db" -delete; find gpurun_out/final -name "*kernel_trace.csv" -delete; find gpurun_out/final -name "*agent_info.csv" -delete
