#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name: mean / sum of each counter and dispatch count.

    python tools/pmc_summary.py <counter_collection.csv> [<more.csv> ...] > summary.json
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(.*$', '', name)
    m = re.search(r'hvx::(\w+)|_ZN3hvx\d+(\w+?)(I|E)', name)
    base = (m.group(1) or m.group(2)) if m else name[:60]
    return base


def main():
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in sys.argv[1:]:
        with open(path, newline='') as f:
            rd = csv.DictReader(f)
            cols = rd.fieldnames
            kcol = next(c for c in cols if c.lower() in ('kernel_name', 'kernel-name', 'name'))
            ncol = next(c for c in cols if c.lower() in ('counter_name', 'counter-name'))
            vcol = next(c for c in cols if c.lower() in ('counter_value', 'counter-value', 'value'))
            for row in rd:
                a = agg[row[kcol]][row[ncol]]
                a[0] += float(row[vcol])
                a[1] += 1
    out = {}
    for k, cs in agg.items():
        out[k] = {c: {'mean': v[0] / v[1], 'sum': v[0], 'dispatches': v[1]} for c, v in cs.items()}
    # class roll-up for the decode weight-streaming GEMM
    roll = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for k, cs in agg.items():
        cls = 'gemm_skinny_kernel' if 'gemm_skinny' in k else ('gemm_tiled_kernel' if 'gemm_tiled' in k else ('attn' if 'attn_' in k else 'other'))
        for c, v in cs.items():
            roll[cls][c][0] += v[0]
            roll[cls][c][1] += v[1]
    out['__classes__'] = {cls: {c: {'mean': v[0] / v[1], 'dispatches': v[1]} for c, v in cs.items()} for cls, cs in roll.items()}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
