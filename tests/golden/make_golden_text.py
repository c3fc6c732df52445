#!/usr/bin/env python3
"""Mints tests/golden/segmentation.json: outputs of the reference's own long-text segmentation functions
(server/model_utils/infer_speech_model.py: split_text_by_punctuation :263-315, merge_short_segments :318-354) on seeded texts.
Runs in the build container only (needs /root/reference); the two function definitions are taken from the reference module's AST and
executed here — the module as a whole does not import (torchaudio, hyperpyyaml, ... are absent).  The fixture holds inputs and outputs only.
"""
import ast
import json
import os
import random

REF = '/root/reference/server/model_utils/infer_speech_model.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'segmentation.json')


def reference_functions():
    tree = ast.parse(open(REF, encoding='utf-8').read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('split_text_by_punctuation', 'merge_short_segments')]
    assert len(keep) == 2
    ns = {}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, 'exec'), ns)
    return ns['split_text_by_punctuation'], ns['merge_short_segments']


def texts():
    rng = random.Random(20250929)
    marks = '。！？；，、.!?;,'
    han = ''.join(chr(c) for c in range(0x4e00, 0x4e00 + 400))
    latin = 'abcdefghijklmnopqrstuvwxyz '
    out = ['', 'short', '。', 'no punctuation at all ' * 12, '，，，，，，，，，，，，，，，，，，，，，，，，，，，，，，，，，，，，', 'a.b.c.d.e.f.g.h.i.j.k.l.m.n.o.p.q.r.s.t.u.v.w.x.y.z.' * 3]
    for n in (10, 29, 30, 31, 55, 120, 400, 5200):
        for alphabet in (han, latin):
            s = []
            while len(s) < n:
                run = rng.randint(1, 24)
                s.extend(rng.choice(alphabet) for _ in range(run))
                if rng.random() < 0.8:
                    s.append(rng.choice(marks))
            out.append(''.join(s[:n]))
    return out


def main():
    split, merge = reference_functions()
    cases = []
    for t in texts():
        for mx, mn in ((30, 10), (50, 10), (30, 5), (12, 4)):
            seg = split(t, mx, mn)
            cases.append(dict(text=t, max_length=mx, min_length=mn, split=seg, merged=merge(seg, mn)))
    with open(OUT, 'w', encoding='utf-8') as f:
        json.dump(dict(source='server/model_utils/infer_speech_model.py:263-354', cases=cases), f, ensure_ascii=False)
    print('wrote', OUT, len(cases), 'cases')


if __name__ == '__main__':
    main()
