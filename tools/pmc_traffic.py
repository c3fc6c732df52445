#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in SEPARATE passes: MI355X_MICROARCH.md, rocprofv3 PMC
slots), with the gfx950 correction of the guide's HBM section: FETCH_SIZE counts half the bytes of a wide coalesced read, so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (rocprofv3 reports KiB).

    python tools/pmc_traffic.py --out profiles/r03_pmc_traffic.json  name=fetch.csv,write.csv[,steps_fetch,steps_write][:regex] ...

name = the bench's kernel-class name (llm_decode_step, dit_gemm_bf16, dit_attention_bf16, hift_conv_gemm_f32); regex selects the kernels of the
class in the counter CSVs (default: every hvx kernel); with steps given the class figure is per STEP (sum over the kernels of a run / steps:
the decode step), otherwise per LAUNCH (mean over the dispatches of the selected kernels).  Every entry lists the mangled names it was counted on
("kernels"): bench.py refuses an entry whose kernels are no longer in libhvx.so."""
import argparse
import csv
import json
import re
from collections import defaultdict


def read(path, counter):
    per = defaultdict(lambda: [0.0, 0])
    with open(path, newline='') as f:
        rd = csv.DictReader(f)
        cols = {c.lower(): c for c in rd.fieldnames}
        k, n, v = cols.get('kernel_name'), cols.get('counter_name'), cols.get('counter_value')
        for row in rd:
            if row[n] != counter:
                continue
            per[row[k]][0] += float(row[v])
            per[row[k]][1] += 1
    return per


ap = argparse.ArgumentParser()
ap.add_argument('--out', required=True)
ap.add_argument('--how', default='')
ap.add_argument('--grid', action='append', default=[], help='name=text: the workload the counters of class `name` were taken on (bench.py prints it as traffic_counted_on)')
ap.add_argument('specs', nargs='+')
a = ap.parse_args()
out = {'_how': a.how}
grids = dict(g.split('=', 1) for g in a.grid)
for spec in a.specs:
    name, rest = spec.split('=', 1)
    rx = None
    if ':' in rest:
        rest, rx = rest.split(':', 1)
    parts = rest.split(',')
    fetch, write = read(parts[0], 'FETCH_SIZE'), read(parts[1], 'WRITE_SIZE')
    steps = (int(parts[2]), int(parts[3])) if len(parts) >= 4 else None
    sel = [k for k in fetch if ('hvx' in k) and (rx is None or re.search(rx, k))]
    if steps:
        f = sum(fetch[k][0] for k in sel) / steps[0]
        w = sum(write[k][0] for k in sel if k in write) / steps[1]
        entry = {'per': 'step', 'steps_fetch_pass': steps[0], 'steps_write_pass': steps[1]}
    else:
        nf = sum(fetch[k][1] for k in sel)
        nw = sum(write[k][1] for k in sel if k in write)
        f = sum(fetch[k][0] for k in sel) / max(nf, 1)
        w = sum(write[k][0] for k in sel if k in write) / max(nw, 1)
        entry = {'per': 'launch', 'launches_fetch_pass': nf, 'launches_write_pass': nw}
    entry.update(fetch_size_kib=round(f, 1), write_size_kib=round(w, 1), hbm_bytes_per_launch=int((2 * f + w) * 1024), kernels=sorted(sel))
    if name in grids:
        entry['grid'] = grids[name]
    entry['per_kernel'] = {k: {'dispatches': fetch[k][1], 'fetch_size_kib_mean': round(fetch[k][0] / fetch[k][1], 1),
                               'write_size_kib_mean': round(write[k][0] / write[k][1], 1) if k in write and write[k][1] else None} for k in sel}
    out[name] = entry
json.dump(out, open(a.out, 'w'), indent=1)
print({k: v['hbm_bytes_per_launch'] for k, v in out.items() if isinstance(v, dict)})
