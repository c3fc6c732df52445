run() { echo "== $*"; env "$@" 2>&1 | grep -v -i "rccl\|amdgpu\|warn" | tail -12; }
I=${ITERS:-3000}
run timeout 300 python tools/platform_probe.py --victim decode --aggressor x3 --iters $I
run timeout 300 python tools/platform_probe.py --victim decode --aggressor f32 --iters $I
run timeout 300 python tools/platform_probe.py --victim decode --aggressor none --iters $I
run HVX_HIFT_FP32_MFMA=1 timeout 300 python tools/platform_probe.py --victim decode --aggressor x3 --iters $I
run timeout 300 python tools/platform_probe.py --victim torch --aggressor x3 --iters $I
run timeout 300 python tools/platform_probe.py --victim decode --aggressor matmul --iters $I
run timeout 300 python tools/platform_probe.py --victim decode --aggressor bf16 --iters $I
run timeout 300 python tools/platform_probe.py --victim source --aggressor x3 --iters $I
