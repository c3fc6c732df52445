#!/usr/bin/env python3
"""python tools/pmc_sq.py DIR: SQ counters of the acoustic stage's kernels from DIR/{flow,hift}_sq1 (rocprofv3 --pmc pass) and DIR/{flow,hift}_stats (a separate
--kernel-trace --stats pass of the same probes) -> JSON on stdout: per kernel the counter means per dispatch and mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES /
(duration x 2.4 GHz x 1024 SIMDs) (nominal clock: the chip runs lower under load), issue / wait fractions = counter / SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import json
import sys

O = sys.argv[1]
out = {'_how': __doc__.strip(), 'kernels': {}}
for w in ('flow', 'hift'):
    dur = {}
    for f in glob.glob(O + '/%s_stats/**/*kernel_stats.csv' % w, recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r['Name']] = float(r['AverageNs']) / 1e3
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(O + '/%s_sq1/**/*counter_collection.csv' % w, recursive=True):
        rd = csv.DictReader(open(f))
        cols = {c.lower(): c for c in rd.fieldnames}
        for r in rd:
            a = agg[r[cols['kernel_name']]][r[cols['counter_name']]]
            a[0] += float(r[cols['counter_value']])
            a[1] += 1
    for k, cs in agg.items():
        if 'hvx' not in k:
            continue
        d = {c: round(v[0] / v[1], 1) for c, v in cs.items()}
        d['dispatches'] = max(v[1] for v in cs.values())
        us = dur.get(k)
        if us:
            d['avg_us'] = round(us, 1)
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
                d['mfma_busy_frac'] = round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / (us * 1e-6 * 2.4e9 * 1024), 3)
        wc = d.get('SQ_WAVE_CYCLES')
        if wc:
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS'):
                if c in d:
                    d[c.lower() + '_frac'] = round(d[c] / wc, 3)
        if d['dispatches'] >= 4:
            out['kernels'][k[:160]] = d
print(json.dumps(out, indent=1))
