# round 4: fp8 gate / up of the MTP heads — parity test, decode step with and without the codes (same box), kernel durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=gpurun_out/fp8_heads_r04.log; : > $L
timeout 900 python -m pytest tests/test_gpu_cv3w.py -m gpu -x -q -s -k fp8 > gpurun_out/pytest_fp8.log 2>&1; grep -E 'passed|failed|fp8 gate|Error|error' gpurun_out/pytest_fp8.log | head -20 >> $L
for a in "" "--head-fp8" "" "--head-fp8"; do echo "== bench_decode --seqs 64 --heads 2 --ctx 1536 $a" >> $L; timeout 300 python tools/bench_decode.py --seqs 64 --heads 2 --ctx 1536 $a 2>&1 | tail -1 >> $L; done
for a in "--heads 4 --seqs 40" "--heads 4 --seqs 40 --head-fp8"; do echo "== bench_decode --ctx 1536 $a" >> $L; timeout 300 python tools/bench_decode.py --ctx 1536 $a 2>&1 | tail -1 >> $L; done
O=gpurun_out/prof_fp8; rm -rf $O; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace --stats -d $O/tr -- python tools/bench_decode.py --seqs 64 --steps 30 --head-fp8 > $O/dec.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1); echo "== kernel durations with --head-fp8" >> $L; grep -E 'gemm_dec_kernel<2, 3, 0|gemm_mid' $f | cut -c1-150 >> $L
rm -rf $O
cut -c1-230 $L
