#!/bin/bash
# Round-6 record on the final build, ONE gpurun call:  gpurun --timeout 3600 -- 'bash tools/final_r06.sh'
#   1. the GPU test-suite and smoke();  2. the driver's bench command;  3. the other BASELINE configurations + the queue worker;
#   4. rocprofv3 --kernel-trace --stats of the bench (8 steps, no extras) and of the isolated decode step / flow / vocoder probes;
#   5. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (never combined with a trace), summarised by tools/pmc_traffic.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r06; rm -rf $O; mkdir -p $O
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "tests wall $(( $(date +%s) - T0 )) s"
T0=$(date +%s); timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; echo "bench wall $(( $(date +%s) - T0 )) s"; tail -1 $O/bench.log | cut -c1-200
B="python bench.py --steps 8 --no-cpu-baseline --no-fp32-mode --no-extras"
timeout 900 $B --config stress > $O/stress.log 2>&1
timeout 500 $B --config zero_shot > $O/zero_shot.log 2>&1
timeout 500 $B --config acoustic > $O/acoustic.log 2>&1
timeout 500 python tools/bench_worker.py > $O/worker.log 2>&1
timeout 500 python tools/bench_matcha.py --decoder cv2 > $O/matcha_cv2.log 2>&1
timeout 500 python tools/bench_matcha.py --decoder matcha > $O/matcha.log 2>&1
timeout 300 python tools/attn_ab.py --forms 16,17,32 --rounds 5 --iters 5 > $O/attn_ab.log 2>&1
timeout 300 python tools/flow_probe.py --utts 4 --iters 4 > $O/flow_probe.log 2>&1
for f in stress zero_shot acoustic worker matcha_cv2 matcha; do echo $f; tail -1 $O/$f.log | cut -c1-220; done
P="rocprofv3 --output-format csv"
timeout 900 $P --kernel-trace --stats -d $O/bench_stats -- $B > $O/bench_under_rocprof.log 2>&1
timeout 600 $P --kernel-trace --stats -d $O/dec_stats -- python tools/bench_decode.py --seqs 64 --steps 100 > $O/dec_stats.log 2>&1
timeout 600 $P --kernel-trace --stats -d $O/dec8_stats -- python tools/bench_decode.py --seqs 8 --steps 100 > $O/dec8_stats.log 2>&1
timeout 600 $P --kernel-trace --stats -d $O/flow_stats -- python tools/flow_probe.py --utts 4 --iters 3 > $O/flow_stats.log 2>&1
timeout 600 $P --kernel-trace --stats -d $O/hift_stats -- python tools/hift_probe.py --iters 3 > $O/hift_stats.log 2>&1
for w in dec flow hift; do
  case $w in dec) C="python tools/bench_decode.py --seqs 64 --steps 10";; flow) C="python tools/flow_probe.py --utts 4 --iters 1";; hift) C="python tools/hift_probe.py --iters 1";; esac
  timeout 600 $P --pmc FETCH_SIZE -d $O/${w}_fetch -- $C > $O/${w}_fetch.log 2>&1
  timeout 600 $P --pmc WRITE_SIZE -d $O/${w}_write -- $C > $O/${w}_write.log 2>&1
done
f() { find $O/$1 -name "*counter_collection.csv" | head -1; }
python tools/pmc_traffic.py --out $O/pmc_traffic.json --how "tools/final_r06.sh: rocprofv3 --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE (never combined with --stats / traces); KiB as rocprofv3 reports them; gfx950 correction (MI355X_MICROARCH.md, HBM): bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024" \
  --grid "llm_decode_step=tools/bench_decode.py --seqs 64 --steps 10: 64 sequences x 2 heads = 128 rows, context 1536, bf16 (15 forward steps incl. warm-up); the bench line's own grid (mean live sequences, mean context) is in its roofline.kernel text" \
  "llm_decode_step=$(f dec_fetch),$(f dec_write),15,15" \
  "dit_gemm_bf16=$(f flow_fetch),$(f flow_write):gemm_big" "dit_attention_bf16=$(f flow_fetch),$(f flow_write):attn_dit" "hift_conv_gemm_f32=$(f hift_fetch),$(f hift_write):x3"
for w in flow hift; do
  case $w in flow) C="python tools/flow_probe.py --utts 4 --iters 1";; hift) C="python tools/hift_probe.py --iters 1";; esac
  timeout 600 $P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/${w}_sq1 -- $C > $O/${w}_sq1.log 2>&1
done
python tools/pmc_sq.py $O > $O/pmc_sq.json 2>$O/pmc_sq.err
for w in bench dec dec8 flow hift; do cp $(find $O/${w}_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${w}.csv 2>/dev/null; done
rm -rf $O/*_stats $O/*_fetch $O/*_write $O/*_sq1
python - <<PY
import json; d=json.load(open('$O/pmc_traffic.json'))
for k,v in d.items():
    if k != '_how': print(k, v['per'], v['hbm_bytes_per_launch'])
PY
head -8 $O/kernel_stats_bench.csv | cut -c1-170; du -sh $O
