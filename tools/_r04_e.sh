cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
for AB in 7 8; do timeout 600 python bench.py --no-cpu-baseline --no-fp32-mode --acoustic-batch $AB > gpurun_out/bench_r04_e$AB.log 2>&1; tail -1 gpurun_out/bench_r04_e$AB.log | cut -c1-200; done
