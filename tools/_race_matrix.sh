export HVX_EXPERIMENTAL_ACOUSTIC_CHAINS=1
run() { echo "== $*"; env "$@" timeout 300 python tools/race_probe.py --lm 1 --acoustic 2 --reps 40 2>&1 | grep -v -i "rccl\|amdgpu\|warn" | tail -1; }
run HVX_X=0
run HVX_DEBUG_SERIALIZE=flow
run HVX_DEBUG_SERIALIZE=hift
run HVX_DEBUG_SERIALIZE=both
run PYTORCH_NO_CUDA_MEMORY_CACHING=1
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=1
run HIP_LAUNCH_BLOCKING=1
