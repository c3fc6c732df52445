cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
dec() { echo "== $*"; env "$@" timeout 300 python tools/bench_decode.py --seqs 64 --heads 2 --ctx $CTX 2>&1 | tail -2; }
for CTX in 1536 1024 2560; do
dec HVX_ATT_ONE_TRIP=0
dec HVX_ATT_ONE_TRIP=1
dec HVX_ATT_CHUNK=384
dec HVX_ATT_CHUNK=384 HVX_ATT_ONE_TRIP=0
dec HVX_ATT_CHUNK=512 HVX_ATT_ONE_TRIP=0
done
timeout 900 python -m pytest tests/test_gpu_cv3w.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
