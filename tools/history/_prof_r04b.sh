# round-4: per-kernel stats of the 128-row decode step confined to 32 CUs (wide and narrow launch geometries)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04b; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace --stats -d $O/w -- python tools/bench_decode.py --seqs 64 --steps 50 --cus 32 > $O/w.log 2>&1
HVX_DEC_GPW_QKV=3 HVX_DEC_GPW_RES=2 HVX_DEC_GPW_MLP=10 HVX_DEC_GPW_DOWN=7 HVX_ATT_CHUNK=1024 rocprofv3 --output-format csv --kernel-trace --stats -d $O/n -- python tools/bench_decode.py --seqs 64 --steps 50 --cus 32 > $O/n.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for d in w n; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; cut -c1-150 $f | head -13 | awk -F, '{print $1, $2, $4}'; done
