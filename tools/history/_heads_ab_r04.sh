# round 4: the MTP heads' GEMMs of a wide decode grid, generic kernels against the ring form, same box (HVX_DEC_HEADS: bit 0 = gate / up, bit 1 = output projection)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=gpurun_out/heads_ab_r04.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout 300 python tools/bench_decode.py --seqs 64 --heads 2 --ctx 1536 2>&1 | tail -1 >> $L; }
run HVX_DEC_HEADS=0
run HVX_DEC_HEADS=2
run HVX_DEC_HEADS=3
run HVX_DEC_HEADS=0
run HVX_DEC_HEADS=3
echo "== tools/bin/stream_lab (cold read floor)" >> $L; tools/bin/stream_lab >> $L 2>&1
echo "== tools/bin/dec_lab hmlp 5 64 (one head's gate / up, 79 MB)" >> $L; timeout 300 tools/bin/dec_lab hmlp 5 64 >> $L 2>&1
cat $L | cut -c1-110
