run() { echo "== $*"; env "$@" 2>&1 | grep -v -i "rccl\|amdgpu\|warn" | tail -${TAILN:-4}; }
TAILN=12 run timeout 300 python tools/mfma_interference.py --launches 2000 --only 6
run timeout 300 python tools/platform_probe.py --victim decode --aggressor x3 --iters 3000
run timeout 300 python tools/platform_probe.py --victim decode --aggressor mfma --iters 3000
run timeout 300 python tools/platform_probe.py --victim decode --aggressor x3 --agg-n 128 --agg-m 40000 --iters 3000
run timeout 300 python tools/race_bisect.py --b same --reps 400
run timeout 300 python tools/race_bisect.py --b same --reps 400 --flow-bf16
run timeout 300 python tools/race_probe.py --lm 1 --acoustic 2 --reps 40
