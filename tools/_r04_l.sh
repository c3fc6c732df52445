cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in 3 6 11; do
O=gpurun_out/prof_r04l_$g; rm -rf $O; mkdir -p $O
HVX_DEC_GPW_HMLP=$g rocprofv3 --output-format csv --kernel-trace --stats -d $O/tr -- python tools/bench_decode.py --seqs 64 --steps 30 > $O/dec.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
echo "== gpw $g"; grep -E 'gemm_dec_kernel<2, 3, 0|gemm_mid|gemm_skinny' $f | cut -c1-140
rm -rf $O
done
