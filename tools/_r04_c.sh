cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/dec_ab.py --layers 2 2>&1 | tail -3
python tools/bench_decode.py --seqs 64 --steps 200 2>&1 | tail -1
B="python bench.py --no-cpu-baseline --no-fp32-mode --no-strict"
python bench.py --help | grep -c no-strict
timeout 600 python bench.py --no-cpu-baseline --no-fp32-mode > gpurun_out/bench_r04_c0.log 2>&1; tail -1 gpurun_out/bench_r04_c0.log | cut -c1-200
NARROW="HVX_DEC_GPW_QKV=3 HVX_DEC_GPW_RES=2 HVX_DEC_GPW_MLP=10 HVX_DEC_GPW_DOWN=7 HVX_ATT_CHUNK=512"
env $NARROW timeout 600 python bench.py --no-cpu-baseline --no-fp32-mode --lm-cus 32 --acoustic-batch 5 > gpurun_out/bench_r04_c1.log 2>&1; tail -1 gpurun_out/bench_r04_c1.log | cut -c1-200
env $NARROW timeout 600 python bench.py --no-cpu-baseline --no-fp32-mode --lm-cus 32 --acoustic-batch 4 > gpurun_out/bench_r04_c2.log 2>&1; tail -1 gpurun_out/bench_r04_c2.log | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --no-fp32-mode --acoustic-batch 5 > gpurun_out/bench_r04_c3.log 2>&1; tail -1 gpurun_out/bench_r04_c3.log | cut -c1-200
