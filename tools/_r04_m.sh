cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python tools/bench_decode.py --seqs 64 --heads 2 --ctx 1536 2>&1 | tail -1 | cut -c1-40; }
run HVX_DEC_HEADS=0
run HVX_DEC_HEADS=2
run HVX_DEC_HEADS=3
run HVX_DEC_HEADS=3 HVX_DEC_GPW_HMLP=6
run HVX_DEC_HEADS=0
run HVX_DEC_HEADS=3
