cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "--layers 1 --heads 2" "--layers 2 --heads 4 --seqs 40" "--layers 2 --heads 1 --seqs 50" "--layers 2 --heads 2 --seqs 33" "--layers 2 --heads 2 --seqs 128" "--layers 2 --heads 5 --seqs 48"; do echo "== dec_ab $a"; timeout 300 python tools/dec_ab.py $a 2>&1 | grep -v amdgpu.ids; done
for g in 6 8 11 16 22; do echo "== gpw_hmlp $g"; HVX_DEC_GPW_HMLP=$g timeout 300 python tools/bench_decode.py --seqs 64 --heads 2 --ctx 1536 2>&1 | tail -1; done
timeout 1200 python -m pytest tests/test_gpu_cv3w.py tests/test_gpu_fullsize.py tests/test_gpu_models.py -m gpu -x -q > gpurun_out/pytest_r04_j.log 2>&1; grep -n 'passed\|failed' gpurun_out/pytest_r04_j.log
