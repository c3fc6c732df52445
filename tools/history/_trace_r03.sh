cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/trc; mkdir -p gpurun_out
timeout 1200 rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -- python bench.py --steps 20 --no-cpu-baseline --no-fp32-mode > gpurun_out/trace_bench.log 2>&1
f=$(find /tmp/trc -name "*kernel_trace.csv" | head -1); ls -la $f; head -1 $f
python tools/trace_gaps.py $f | tee gpurun_out/trace_gaps.log
