# round-4 record on the final build (one gpurun call): GPU test-suite, smoke, the driver's bench line, the other BASELINE configurations, head_num sweep, queue worker
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; echo "bench wall $(( $(date +%s) - T0 )) s"; tail -1 $O/bench.log | cut -c1-200
B="python bench.py --steps 8 --no-cpu-baseline --no-fp32-mode"
# head_num sweep at the driver's step count (8 steps = 64 utterances would not even fill the 128-slot grid of head_num 1)
timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-fp32-mode --heads 1 > $O/k1.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-fp32-mode --heads 4 > $O/k4.log 2>&1
timeout 900 $B --config stress > $O/stress.log 2>&1
timeout 500 $B --config zero_shot > $O/zero_shot.log 2>&1
timeout 500 $B --config acoustic > $O/acoustic.log 2>&1
timeout 500 python tools/bench_worker.py > $O/worker.log 2>&1
for f in k1 k4 stress zero_shot acoustic worker; do echo $f; tail -1 $O/$f.log | cut -c1-220; done
