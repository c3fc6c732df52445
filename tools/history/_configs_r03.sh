# the other BASELINE configurations and the head_num sweep on the final round-3 build (one GPU, 8 timed steps each)
mkdir -p gpurun_out/cfgs
B="python bench.py --steps 8 --no-cpu-baseline --no-fp32-mode"
timeout 500 $B --heads 1 > gpurun_out/cfgs/k1.log 2>&1
timeout 500 $B --heads 4 > gpurun_out/cfgs/k4.log 2>&1
timeout 900 $B --config stress > gpurun_out/cfgs/stress.log 2>&1
timeout 500 $B --config zero_shot > gpurun_out/cfgs/zero_shot.log 2>&1
timeout 500 $B --config acoustic > gpurun_out/cfgs/acoustic.log 2>&1
timeout 500 python tools/bench_worker.py > gpurun_out/cfgs/worker.log 2>&1
for f in k1 k4 stress zero_shot acoustic worker; do echo $f; tail -1 gpurun_out/cfgs/$f.log | cut -c1-220; done
