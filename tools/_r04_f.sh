cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-fp32-mode > gpurun_out/bench_r04_f_$tag.log 2>&1; python - <<PY
import json
d=json.loads(open('gpurun_out/bench_r04_f_$tag.log').read().strip().splitlines()[-1])
r=d['roofline']
print('$tag', '$*', d['value'], d['ms_per_step'], 'in-job us', r['avg_launch_us'], 'alone', r['alone']['avg_launch_us'], 'llm', d['stage_seconds_per_step']['llm'], 'ac', d['stage_seconds_per_step']['flow+hift'])
PY
}
run base A=1
run g2 HVX_DEC_GPW_QKV=2 HVX_DEC_GPW_RES=2
run g2m5 HVX_DEC_GPW_QKV=2 HVX_DEC_GPW_RES=2 HVX_DEC_GPW_MLP=5 HVX_DEC_GPW_DOWN=4
run att512 HVX_ATT_CHUNK=512
run g2att512 HVX_DEC_GPW_QKV=2 HVX_DEC_GPW_RES=2 HVX_ATT_CHUNK=512
run base2 A=2
