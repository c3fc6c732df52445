cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python tools/bench_decode.py --seqs 64 --heads 2 --ctx 1536 2>&1 | tail -1 | cut -c1-40; }
run HVX_DEC_HMLP_D=4 HVX_DEC_GPW_HMLP=11
run HVX_DEC_HMLP_D=6 HVX_DEC_GPW_HMLP=11
run HVX_DEC_HMLP_D=8 HVX_DEC_GPW_HMLP=11
run HVX_DEC_HMLP_D=8 HVX_DEC_GPW_HMLP=6
run HVX_DEC_HMLP_D=10 HVX_DEC_GPW_HMLP=11
run HVX_DEC_HMLP_D=10 HVX_DEC_GPW_HMLP=6
run HVX_DEC_HMLP_D=6 HVX_DEC_GPW_HMLP=4
run HVX_DEC_HMLP_D=4 HVX_DEC_GPW_HMLP=3
