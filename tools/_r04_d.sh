cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for S in 96 128; do python tools/bench_decode.py --seqs $S --steps 100 2>&1 | tail -1; done
for S in 96 128; do timeout 600 python bench.py --no-cpu-baseline --no-fp32-mode --lm-slots $S > gpurun_out/bench_r04_d$S.log 2>&1; tail -1 gpurun_out/bench_r04_d$S.log | cut -c1-200; done
