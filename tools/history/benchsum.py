import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    print(f, 'value', d["value"], 'ms/step', d["ms_per_step"], d["stage_seconds_per_step"]["llm"], d["stage_seconds_per_step"]["flow+hift"], 'serial', d["stage_seconds_serial"], 'decode us', d["roofline"]["avg_launch_us"], 'alone', d["roofline"]["alone"]["avg_launch_us"], d["roofline"]["alone"]["frac"])
    print('   ', [(r["kernel"], r["achieved"], r["avg_launch_us"]) for r in d["roofline_other"]], d["kernel_time_share_ms"])
