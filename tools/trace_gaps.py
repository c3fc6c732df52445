#!/usr/bin/env python3
"""Where does a decode step's time go beside the acoustic stage?  From a rocprofv3 --kernel-trace CSV of a bench run: the kernels of the decode
engine (one stream, dependent launches) in the middle half of the run — sum of their durations, sum of the gaps between one's end and the next
one's start, per engine step (a step = one ras_sample_kernel) — and the same for the acoustic stream's kernels.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --steps 4 --no-cpu-baseline --no-fp32-mode
    python tools/trace_gaps.py DIR/.../*_kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict

LM = ('attn_fwd', 'gemm_skinny', 'gemm_narrow', 'gemm_mid', 'reduce_rmsnorm', 'attn_combine', 'ras_sample', 'heads_prologue', 'log_softmax', 'embed2',
      'decode_advance', 'heads_')
rows = []
with open(sys.argv[1], newline='') as f:
    rd = csv.DictReader(f)
    cols = {c.lower(): c for c in rd.fieldnames}
    kn, st, en = cols['kernel_name'], cols['start_timestamp'], cols['end_timestamp']
    qid = cols.get('queue_id')
    for r in rd:
        rows.append((int(r[st]), int(r[en]), r[kn], r[qid] if qid else ''))
rows.sort()
# the window: the 1-second bins of the run in which BOTH stages are busy (acoustic kernels in flight > 60 % of the bin and > 100 decode steps) —
# the steady state of the continuous engine; the alone-measurements, the warm-up and the strict_batch8 step fall out
t0 = rows[0][0]
nb = int((rows[-1][1] - t0) / 1e9) + 1
ab = [0.0] * nb
ls = [0] * nb
for r in rows:
    i = int((r[0] - t0) / 1e9)
    if 'ras_sample' in r[2]:
        ls[i] += 1
    elif 'hvx' in r[2] and not any(k in r[2] for k in LM):
        ab[i] += (r[1] - r[0]) / 1e9
print('bins (s: decode steps / acoustic busy): ' + ' '.join('%d:%d/%.2f' % (i, ls[i], ab[i]) for i in range(nb)))
good = [i for i in range(nb) if ab[i] > 0.6 and ls[i] > 100]
if not good:
    raise SystemExit('no steady-state bin found')
# the longest run of consecutive good bins, minus its first and last
runs, cur = [], [good[0]]
for i in good[1:]:
    if i == cur[-1] + 1: cur.append(i)
    else: runs.append(cur); cur = [i]
runs.append(cur)
best = max(runs, key=len)
if len(best) > 2: best = best[1:-1]
lo, hi = t0 + best[0] * 1e9, t0 + (best[-1] + 1) * 1e9
mid = [r for r in rows if lo <= r[0] <= hi]
lm = [r for r in mid if any(k in r[2] for k in LM)]
ac = [r for r in mid if not any(k in r[2] for k in LM) and 'hvx' in r[2]]
steps = sum(1 for r in lm if 'ras_sample' in r[2])
dur = sum(r[1] - r[0] for r in lm)
gaps = [lm[i + 1][0] - lm[i][1] for i in range(len(lm) - 1)]
pos = [g for g in gaps if g > 0]
span = lm[-1][1] - lm[0][0]
print('window %.1f ms: %d decode kernels in %d steps (%.0f per step), queues %s' % ((hi - lo) / 1e6, len(lm), steps, len(lm) / max(steps, 1), sorted(set(r[3] for r in lm))))
print('per step: span %.2f ms = kernel durations %.2f ms + gaps %.2f ms; mean duration %.1f us, mean gap %.1f us, median gap %.1f us' %
      (span / 1e6 / steps, dur / 1e6 / steps, sum(pos) / 1e6 / steps, dur / 1e3 / len(lm), sum(pos) / 1e3 / len(pos), sorted(pos)[len(pos) // 2] / 1e3))
by = defaultdict(lambda: [0, 0.0, 0.0])
for i, r in enumerate(lm[:-1]):
    k = r[2].split('(')[0][-40:]
    by[k][0] += 1; by[k][1] += r[1] - r[0]; by[k][2] += max(gaps[i], 0)
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1] - kv[1][2])[:12]:
    print('  %-42s n/step %5.1f  dur %6.1f us  gap after %6.1f us' % (k, v[0] / steps, v[1] / v[0] / 1e3, v[2] / v[0] / 1e3))
adur = sum(r[1] - r[0] for r in ac)
print('acoustic stream in the same window: %d kernels, busy %.1f %% of the window (sum of durations; queues %s)' % (len(ac), 100.0 * adur / (hi - lo), sorted(set(r[3] for r in ac))))
# how much of the decode kernels' time overlaps an acoustic kernel
import bisect
ast = sorted((r[0], r[1]) for r in ac)
starts = [a[0] for a in ast]
ov = 0
for r in lm:
    i = bisect.bisect_right(starts, r[1])
    for a in ast[max(0, i - 6):i]:
        ov += max(0, min(a[1], r[1]) - max(a[0], r[0]))
print('decode kernel time that runs while an acoustic kernel is in flight: %.0f %%' % (100.0 * min(ov, dur) / dur))
