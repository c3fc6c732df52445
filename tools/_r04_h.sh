cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "--layers 1 --heads 2" "--layers 24 --heads 2" "--layers 2 --heads 4 --seqs 40" "--layers 2 --heads 1 --seqs 50" "--layers 2 --heads 2 --seqs 17" "--layers 2 --heads 2 --seqs 128"; do echo "== dec_ab $a"; timeout 300 python tools/dec_ab.py $a 2>&1 | grep -v amdgpu.ids; done
for g in 2 3 4 6; do echo "== gpw_out $g"; HVX_DEC_GPW_OUT=$g timeout 300 python tools/bench_decode.py --seqs 64 --heads 2 --ctx 1536 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_gpu_cv3w.py tests/test_gpu_fullsize.py tests/test_gpu_models.py -m gpu -x -q 2>&1 | tail -3
