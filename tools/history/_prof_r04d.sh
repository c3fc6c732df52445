# round-4, final build: decode-step kernel stats + HBM traffic passes (separate rocprofv3 runs), summarised on the box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04d; rm -rf $O; mkdir -p $O
P="rocprofv3 --output-format csv"
$P --kernel-trace --stats -d $O/dec_stats -- python tools/bench_decode.py --seqs 64 --steps 100 > $O/dec_stats.log 2>&1
$P --pmc FETCH_SIZE -d $O/dec_fetch -- python tools/bench_decode.py --seqs 64 --steps 10 > $O/dec_fetch.log 2>&1
$P --pmc WRITE_SIZE -d $O/dec_write -- python tools/bench_decode.py --seqs 64 --steps 10 > $O/dec_write.log 2>&1
F=$(find $O/dec_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/dec_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py --out $O/pmc_traffic.json --how "see profiles/README.md" llm_decode_step=$F,$W,15,15
cp $(find $O/dec_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_decode_step.csv
rm -rf $O/dec_stats $O/dec_fetch $O/dec_write
python - <<PY
import json; d=json.load(open('$O/pmc_traffic.json'))['llm_decode_step']; print(d['hbm_bytes_per_launch'], d['fetch_size_kib'], d['write_size_kib'])
PY
head -14 $O/kernel_stats_decode_step.csv | cut -c1-170; tail -1 $O/dec_stats.log | cut -c1-120
