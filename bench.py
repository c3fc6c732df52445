#!/usr/bin/env python3
"""bench.py — HydraVox-CV3 speech-synthesis hot path on MI355X (BASELINE.json metric / config).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): HydraVox-CV3, inference_head_num=2, batch = 8 x 512-char utterances
per GPU (512 text tokens -> 2816 speech tokens, generation length pinned by min = max token/text ratio 5.5 because random
weights never emit EOS sensibly -> 5632 mel frames -> 112.6 s of 24 kHz audio each), LLM and DiT in bf16, HiFT in fp32,
sampler top_p 0.9 / top_k 10 / win_size 32 / tau_r 0.2, seeded N(0, 0.02) random weights of the [ASSUMED-CV3] architecture.
One step = one batch through llm -> flow -> hift.  N > 1: weak scaling, every rank synthesises its own 8 utterances
(global utterance index = seed) and the finished waveforms are gathered to rank 0 (grouped RCCL send/recv) inside the step.

`value` = speech tokens per second of the whole job (all stages, all ranks); the reference's own per-stage figures
(TPS = tokens / LLM wall, RTF = wall / audio seconds; infer_speech_model.py:563-565, 594-604) ride along as extra keys.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TF = 2500.0       # dense bf16
MFMA_F32_PEAK_TF = 157.3         # f32-input MFMA == fp32 vector rate
# The vocoder convolutions compute in fp32-equivalent arithmetic as THREE bf16 MFMAs per step (gemm_x3.hip): the rate their instruction mix
# allows is a third of the bf16 matrix peak, and that is what their fraction is quoted against (never the 157 TF/s of the fp32 MFMA they replace)
MFMA_X3_PEAK_TF = MFMA_BF16_PEAK_TF / 3.0

KINDS = {0: ('llm_decode_gemm', 'hbm'), 1: ('dit_gemm_bf16', 'mfma'), 2: ('dit_attention_bf16', 'mfma'), 3: ('ras_sampler', 'hbm'),
         4: ('hift_conv_gemm_f32', 'mfma'), 5: ('llm_attention', 'hbm')}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20, help='timed steps of 8 utterances; 20 (160 utterances through 64 decode slots) is the steady state of the continuous engine — 8 would let all utterances start at once and run LM-then-acoustic in series')
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=8, help='utterances per GPU per step')
    ap.add_argument('--chars', type=int, default=512, help='text tokens per utterance')
    ap.add_argument('--heads', type=int, default=2, help='inference_head_num')
    ap.add_argument('--config', choices=['tts', 'stress', 'zero_shot', 'acoustic'], default='tts',
                    help='tts: BASELINE configs[1] (the headline, default); stress: configs[2] (4 heads, batch 32); zero_shot: configs[3] (mixed lengths U{64..512} with a 3 s '
                         'prompt: 75 speech tokens + 150 mel frames + 20 prompt-text tokens, 8 per GPU and step); acoustic: configs[4] (flow + vocoder only on pre-tokenised streams)')
    ap.add_argument('--streams', type=int, default=64, help='(--config acoustic) pre-tokenised speech-token streams per GPU, lengths U{352..2816}')
    ap.add_argument('--tiny', action='store_true', help='toy dimensions (plumbing check only; INVALID as a benchmark)')
    ap.add_argument('--llm-dtype', choices=['bf16', 'fp32'], default='bf16', help='fp32: the speech-token LM on the exact fp32 forms (ids bit-exact against the reference) with the flow decoder / vocoder as in the headline')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fp32-mode', action='store_true', help='skip the extra, untimed step in the parity-exact fp32 mode (the `fp32_mode` object of the line)')
    ap.add_argument('--lm-slots', type=int, default=0, help='sequences decoded in one grid (continuous batching): utterances of later steps join as earlier ones finish; '
                    'default: a grid of 128 rows — 64 sequences at head_num >= 2, 128 at head_num 1 (the LM-bound point of the sweep: 15.1 k -> 16.4 k tokens/s)')
    ap.add_argument('--mode', choices=['continuous', 'chains', 'serial'], default='continuous',
                    help='continuous: one decode grid of --lm-slots sequences + acoustic stage of finished utterances beside it (default); '
                         'chains: the round-1 form, --lm-chains independent decode chains of one step each; serial: stages back to back')
    ap.add_argument('--acoustic-batch', type=int, default=4, help='utterances per padded CFM solve (x CFG 2 rows per estimator call)')
    ap.add_argument('--acoustic-min-batch', type=int, default=4, help='(--mode continuous) the acoustic stage waits for this many finished utterances (throughput over latency)')
    ap.add_argument('--lm-pace', default='', help='(--mode continuous) admission pacing of the decode grid "first,every_steps,more": open with `first` sequences, admit `more` every `every_steps` decode steps (empty: fill all slots at once)')
    ap.add_argument('--acoustic-cus', type=int, default=0, help='(--mode continuous) confine the acoustic stage to this many compute units (the decode engine stays unconfined); 0: all')
    ap.add_argument('--lm-cus-only', type=int, default=0, help='(--mode continuous) confine ONLY the decode engine to this many compute units; the acoustic stage keeps all of them')
    ap.add_argument('--lm-cus', type=int, default=0, help='(--mode continuous) compute units reserved for the decode engine; the acoustic stage runs on the others (0: both share all CUs)')
    ap.add_argument('--acoustic-chains', type=int, default=1, help='(--mode chains) kept for compatibility: more than one concurrent acoustic chain is not supported (clamped to 1)')
    ap.add_argument('--lm-chains', type=int, default=3, help='(--mode chains) batches whose LM decode runs concurrently')
    ap.add_argument('--serial', action='store_true', help='same as --mode serial')
    ap.add_argument('--prof-period', type=int, default=17, help='every n-th launch of a kernel class is bracketed in the profiling step (prime: no aliasing with the 4-GEMM block period)')
    ap.add_argument('--no-extras', action='store_true', help='skip the untimed extra objects of the line: single_request (BASELINE configs[0] shape), head_sweep (head_num 1 / 4), bf16_id_agreement')
    ap.add_argument('--sweep-steps', type=int, default=8, help='steps of 8 utterances per point of the head_num sweep')
    ap.add_argument('--hift-exact', action='store_true', help='the vocoder on the exact fp32 MFMA forms (hvx_hift_config.exact_fp32) instead of split-bf16 pairs')
    ap.add_argument('--config-steps', type=int, default=4, help='steps of each of the other BASELINE configs (stress, zero_shot, acoustic) run behind the headline as the `configs` object; 0: skip')
    ap.add_argument('--lib-opt', action='append', default=[], metavar='NAME=VALUE', help='library option(s) set before anything runs (hvx_set_option; A / B runs, e.g. attn_dit_form=16 = the round-5 attention tile); recorded in config.lib_options')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--stub-pipeline', action='store_true', help=argparse.SUPPRESS)      # plumbing check of the rank logic on CPU / gloo (tools/bench_stub.py); INVALID as a benchmark
    return ap.parse_args()


def read_prof(lib):
    out = {}
    for k, (name, bound) in KINDS.items():
        ms, work, ns, nl, lw = C.c_double(), C.c_double(), C.c_int64(), C.c_int64(), C.c_double()
        lib.hvx_prof_read(k, C.byref(ms), C.byref(work), C.byref(ns), C.byref(nl), C.byref(lw))
        if ns.value:
            out[name] = dict(bound=bound, sampled=ns.value, launches=nl.value, avg_us=1e3 * ms.value / ns.value,
                             est_total_ms=ms.value * nl.value / ns.value, work_per_launch=work.value / ns.value,
                             rate=(work.value / (ms.value * 1e-3)) if ms.value > 0 else 0.0, launched_work=lw.value)
    return out


_LIB_BYTES = None


def _kernel_in_lib(name):
    """is the kernel rocprofv3 named `name` (mangled, or demangled with its template arguments) a symbol of the libhvx.so being benchmarked?"""
    import re
    if name.encode() in _LIB_BYTES:
        return True
    m = re.search(r'(\w+)<([^<>]*)>\(', name)
    if m:                                                   # demangled: rebuild the Itanium template-argument list of integers / booleans
        ident, targs, enc = m.group(1), [t.strip() for t in m.group(2).split(',')], ''
        for t in targs:
            if t in ('true', 'false'):
                enc += 'Lb%dE' % (t == 'true')
            elif re.fullmatch(r'-?\d+', t):
                enc += 'Li%sE' % t.replace('-', 'n')
            else:
                enc = None                                  # a type argument (or a demangler artefact): the identifier alone has to do
                break
        if enc is not None:
            return ('%d%sI%sE' % (len(ident), ident, enc)).encode() in _LIB_BYTES
        return ('%d%s' % (len(ident), ident)).encode() in _LIB_BYTES
    m = re.search(r'(\w+)\(', name)
    return bool(m) and ('%d%s' % (len(m.group(1)), m.group(1))).encode() in _LIB_BYTES


def pmc_traffic(name, with_grid=False):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r*_pmc_traffic.json; FETCH_SIZE / WRITE_SIZE in
    separate passes with the gfx950 x2 FETCH correction).  PMC collection cannot run inside the timed bench.  An entry is used only if
    every kernel it was counted on (its "kernels" list of mangled names) is still a kernel of the libhvx.so being benchmarked — counters
    of kernels that no longer exist are stale evidence and are refused (null)."""
    import glob
    global _LIB_BYTES
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if name not in d:
                continue
            if _LIB_BYTES is None:
                with open(os.path.join(ROOT, 'flowmirror_hydravox_amd', 'libhvx.so'), 'rb') as f:
                    _LIB_BYTES = f.read()
            kernels = d[name].get('kernels')
            if not kernels or not all(_kernel_in_lib(k) for k in kernels):
                return (None, None) if with_grid else None
            b = int(d[name]['hbm_bytes_per_launch'])
            return (b, d[name].get('grid', os.path.basename(path))) if with_grid else b
        except Exception:
            pass
    return (None, None) if with_grid else None


def roofline_of(name, p):
    if p['bound'] == 'hbm':
        ach = p['rate'] / 1e9
        return dict(kernel=name, bound='hbm', achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ach / HBM_PEAK_GBS, 4),
                    traffic=pmc_traffic(name), avg_launch_us=round(p['avg_us'], 2), algorithmic_bytes_per_launch=round(p['work_per_launch']),
                    launches_per_step=p['launches'], launches_per_timed_region=p['launches'] * p.get('steps', 1), sampled_launches=p['sampled'])
    peak = MFMA_X3_PEAK_TF if name.endswith('f32') else MFMA_BF16_PEAK_TF
    ach = p['rate'] / 1e12
    d = dict(kernel=name, bound='mfma', achieved=round(ach, 2), peak=round(peak, 1), unit='TFLOP/s', frac=round(ach / peak, 4), traffic=pmc_traffic(name),
             avg_launch_us=round(p['avg_us'], 2), algorithmic_flops_per_launch=round(p['work_per_launch']),
             launches_per_step=p['launches'], launches_per_timed_region=p['launches'] * p.get('steps', 1), sampled_launches=p['sampled'],
             measured='hipEvent brackets around every %d-th launch of the class in ONE extra profiling step (same kernels, same geometry as a timed step)' % p.get('period', 1))
    if name.endswith('f32'):
        d['peak_note'] = 'fp32-equivalent flops of the convolution; every step is 3 bf16 MFMAs on (hi, lo) operand pairs, so the peak is the dense bf16 peak / 3'
        d['includes'] = 'every split-bf16 GEMM launch of the timed region: the vocoder convolutions and, in the flow\'s reference-precision mode, the fp32 input projection of its estimator calls'
    return d


def cpu_baseline(cfg, pipe_seed, chars, heads):
    """Reference algorithm on the host cores (oracle/ = the fp32 CPU restatement), BASELINE.md §2: the LM with the reference's
    no-KV-cache full-prefix recompute per AR step over a wall-clock budget at the true context offsets (up to 128 tokens), the same LM
    KV-cached ("fair CPU"), the flow decoder at T = 704 and T = 1408 frames extrapolated to the bench length with a*T + b*T^2, HiFT at
    T = 704 scaled linearly.  Bounded: about 80 s of host work."""
    from oracle import llm_ref, flow_ref, hift_ref, sampler_ref
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.pipeline import synthetic_utterance
    # 256 OpenMP threads on a 2-socket host make these small fp32 GEMMs slower, not faster: use at most 32 and say so
    cores = min(os.cpu_count() or 1, int(os.environ.get('HVX_CPU_BASELINE_THREADS', '32')))
    torch.set_num_threads(cores)
    budget = float(os.environ.get('HVX_CPU_BASELINE_LLM_SECONDS', '20'))
    u = synthetic_utterance(cfg, 0, chars)
    sampling = dict(top_p=0.9, top_k=10, win_size=32, tau_r=0.2)
    sd = W.make_llm_state(cfg.llm, seed=pipe_seed, init='normal02')
    ratio = 5.5

    def run_llm(use_kv, seconds, max_tokens):
        n, t0 = 0, time.time()
        for _ in llm_ref.llm_inference(sd, cfg.llm, u.text, sampler_ref.NoiseStream(seed=0), inference_head_num=heads, sampling=sampling,
                                       max_token_text_ratio=ratio, min_token_text_ratio=ratio, use_kv_cache=use_kv):
            n += 1
            if n >= max_tokens or time.time() - t0 > seconds:
                break
        return n, time.time() - t0
    n_nc, t_nc = run_llm(False, budget, 128)                 # reference behaviour
    n_kv, t_kv = run_llm(True, 10.0, 128)                    # "fair CPU": same arithmetic, keys / values kept (includes the prefill of the prefix)
    # BASELINE configs[0] on the host cores (BASELINE.md §2 "config 1 is timed in full"): ONE 64-char utterance at head_num 1 — the KV-cached LM in full
    # here (352 steps); the reference's literal no-cache loop in full only with HVX_CPU_BASELINE_FULL=1 (minutes), else its full timing in the build
    # container rides along from tests/golden/single_cv3.npz (where the reference's own modules produced that fixture)
    single = None
    if chars >= 64:
        u1 = synthetic_utterance(cfg, 0, 64)

        def run_single(use_kv):
            t0 = time.time()
            n = sum(1 for _ in llm_ref.llm_inference(sd, cfg.llm, u1.text, sampler_ref.NoiseStream(seed=0), inference_head_num=1, sampling=sampling,
                                                     max_token_text_ratio=ratio, min_token_text_ratio=ratio, use_kv_cache=use_kv))
            return n, time.time() - t0
        n1, t1 = run_single(True)
        single = {'tokens': n1, 'llm_seconds_kv_cached': round(t1, 2)}
        if os.environ.get('HVX_CPU_BASELINE_FULL') == '1':
            n1u, t1u = run_single(False)
            single['llm_seconds_no_cache_reference_loop'] = round(t1u, 2)
        try:
            import numpy as np
            g = np.load(os.path.join(ROOT, 'tests', 'golden', 'single_cv3.npz'))
            rs = [float(v) for v in g['ref_seconds']]
            single['reference_modules_in_build_container'] = {'threads': int(g['ref_threads']), 'llm_seconds_no_cache': round(rs[0], 1), 'flow_seconds': round(rs[1], 1),
                                                              'hift_seconds': round(rs[2], 1), 'tokens_per_s': round(len(g['tokens']) / sum(rs), 3),
                                                              'note': 'the reference\'s own llm.inference / flow decoder / hift.inference on this shape, timed in full when the fixture was minted (tests/golden/make_golden_fullsize.py: gen_single)'}
        except Exception:
            pass
    del sd
    t_llm = t_nc / max(n_nc, 1)
    # ---- flow: 10 Euler steps x CFG 2 at two lengths -> a*T + b*T^2 (linear layers / attention), extrapolated to the bench length
    sdf = W.make_flow_state(cfg.flow, seed=pipe_seed + 1, init='normal02')
    g = torch.Generator().manual_seed(1)
    n_spk = int(chars * ratio)
    T_full = n_spk * cfg.flow.token_mel_ratio
    t_flow, mel = {}, None
    for n_tok in (352, 704):
        token = torch.randint(0, cfg.flow.vocab, (1, n_tok), generator=g)
        t0 = time.time()
        m = flow_ref.flow_inference(token, u.embedding[None], sdf, cfg.flow)
        t_flow[n_tok * cfg.flow.token_mel_ratio] = time.time() - t0
        if mel is None:
            mel = m
    del sdf
    (T1, f1), (T2, f2) = sorted(t_flow.items())
    qb = max((f2 / T2 - f1 / T1) / (T2 - T1), 0.0)
    qa = f1 / T1 - qb * T1
    flow_full = qa * T_full + qb * T_full * T_full
    sdh = W.make_hift_state(cfg.hift, seed=pipe_seed + 2, init='normal02')
    tables = hift_ref.make_tables(cfg.hift, seed=0, n_samples=mel.shape[-1] * cfg.hift.upsample_total)
    t0 = time.time()
    hift_ref.hift_inference(mel, sdh, cfg.hift, tables)
    t_hift = time.time() - t0
    hift_full = t_hift * T_full / mel.shape[-1]
    total = n_spk * t_llm + flow_full + hift_full            # one bench utterance on the host cores
    if single is not None and T1 == 704:
        # (the flow / vocoder timings above at T = 704 ARE this utterance's length)
        single.update(flow_seconds=round(f1, 2), hift_seconds=round(t_hift * 704 / mel.shape[-1], 2))
        single['latency_s_kv_cached'] = round(single['llm_seconds_kv_cached'] + single['flow_seconds'] + single['hift_seconds'], 2)
        single['tokens_per_s_kv_cached'] = round(single['tokens'] / single['latency_s_kv_cached'], 3)
        single['workload'] = 'BASELINE configs[0]: head_num 1, one 64-char utterance (64 text -> 352 tokens -> 704 frames), oracle/ on %d threads' % cores
    return dict(single_request=single, value=round(n_spk / total, 3), unit='speech-tokens/s', cores=cores, kind='port',
                sample='oracle/ (fp32 CPU restatement of the reference algorithm, torch CPU, %d threads) on one %d-char utterance of the bench workload: '
                       'LM = first %d speech tokens in %.1f s with the reference\'s no-KV-cache full-prefix recompute at the true context offsets '
                       '(context %d..%d; the utterance mean is ~%d, so this favours the CPU); flow (10 Euler steps x CFG 2) measured at T = %d / %d frames '
                       '(%.1f / %.1f s) and extrapolated to T = %d with a*T + b*T^2 (%.0f s); HiFT measured at T = %d (%.1f s), scaled linearly (%.0f s); '
                       'value = %d tokens / (%d * LM seconds per token + flow + HiFT)'
                       % (cores, chars, n_nc, t_nc, chars + 2, chars + 2 + n_nc, chars + 2 + n_spk // 2, T1, T2, f1, f2, T_full, flow_full, mel.shape[-1], t_hift,
                          hift_full, n_spk, n_spk),
                llm_tokens_per_s=round(n_nc / t_nc, 3),
                fair_cpu_kv_cached={'llm_tokens_per_s': round(n_kv / t_kv, 3), 'tokens': n_kv, 'seconds': round(t_kv, 2),
                                    'value': round(n_spk / (n_spk * t_kv / max(n_kv, 1) + flow_full + hift_full), 3),
                                    'note': 'the same LM arithmetic with a KV cache (prefix prefill included in the seconds); flow / HiFT as above'},
                flow_seconds={'T%d' % T1: round(f1, 2), 'T%d' % T2: round(f2, 2), 'T%d_extrapolated' % T_full: round(flow_full, 1)},
                hift_seconds={'T%d' % mel.shape[-1]: round(t_hift, 2), 'T%d_scaled' % T_full: round(hift_full, 1)})


def single_request(pipe, cfg, ratio, chars=64, heads=(1, 2), repeats=3):
    """BASELINE configs[0]'s shape on the GPU, the way the reference serves it (server/model_utils/infer_speech_model.py:612-681: ONE utterance per
    call, llm -> flow -> hift back to back): a 64-char utterance (64 text -> 352 speech tokens -> 704 frames) at head_num 1 and 2, batch 1, on the
    headline's pipeline (bf16 LM on the 16-row decode kernels, bf16 flow, fp32-contract vocoder).  Latency = best of `repeats` after one warm-up."""
    from flowmirror_hydravox_amd.pipeline import synthetic_utterance
    u = synthetic_utterance(cfg, 0, chars)
    k0 = pipe.llm.inference_head_num
    out = {'workload': 'one %d-char utterance per call (%d text -> %d speech tokens -> %d mel frames), batch 1, stages back to back (infer_speech_model.py:612-681); '
                       'best of %d calls after a warm-up' % (chars, chars, int(chars * ratio), 2 * int(chars * ratio), repeats)}
    try:
        for k in heads:
            pipe.llm.inference_head_num = k
            pipe.synthesize([u], max_token_text_ratio=ratio, min_token_text_ratio=ratio)
            best = None
            for _ in range(repeats):
                torch.cuda.synchronize()
                t0 = time.time()
                wavs, st = pipe.synthesize([u], max_token_text_ratio=ratio, min_token_text_ratio=ratio)
                torch.cuda.synchronize()
                dt = time.time() - t0
                if best is None or dt < best[0]:
                    best = (dt, st)
            dt, st = best
            us, by = st.llm.get('decode_step_us', 0.0), st.llm.get('decode_step_bytes', 0.0)
            out['head_num_%d' % k] = {'latency_s': round(dt, 4), 'tokens': st.tokens, 'tokens_per_s': round(st.tokens / dt, 1), 'llm_tokens_per_s': round(st.tps, 1),
                                      'rtf': round(dt / st.audio_seconds, 6), 'stage_seconds': {'llm': round(st.llm_seconds, 4), 'flow': round(st.flow_seconds, 4), 'hift': round(st.hift_seconds, 4)},
                                      'decode_step_us': round(us, 1), 'decode_step_frac_of_hbm': round(by / us / 1e3 / HBM_PEAK_GBS, 4) if us else None}
    finally:
        pipe.llm.inference_head_num = k0
    return out


def head_sweep(pipe, args, make_utt, ratio, k_line, B):
    """north_star: "RTF and speech-tokens/sec at head_num in {1, 2, 4}" — the two points the line's `value` is not, through the same continuous engine
    (grid of 128 ROWS for head_num 1, 64 sequences otherwise), `--sweep-steps` steps of B utterances each, untimed for `value`."""
    out = {'steps': args.sweep_steps, 'utterances': args.sweep_steps * B,
           'note': 'same pipeline and engine as `value` (head_num %d, %d steps); fewer steps: the fill of the first grid weighs more' % (k_line, args.steps)}
    k0 = pipe.llm.inference_head_num
    try:
        for k in (1, 2, 4):
            if k == k_line:
                continue
            pipe.llm.inference_head_num = k
            slots = 128 if k == 1 else 64
            job = [make_utt(g) for g in range(args.sweep_steps * B)]
            torch.cuda.synchronize()
            t0 = time.time()
            tok = 0
            for i, wav, toks in pipe.synthesize_continuous(job, lm_slots=slots, max_token_text_ratio=ratio, min_token_text_ratio=ratio,
                                                           acoustic_batch=args.acoustic_batch, acoustic_min_batch=args.acoustic_min_batch):
                tok += len(toks)
            torch.cuda.synchronize()
            dt = time.time() - t0
            c = dict(pipe.last_continuous)
            us, by = c['llm'].get('decode_step_us', 0.0), c['llm'].get('decode_step_bytes', 0.0)
            out['head_num_%d' % k] = {'value': round(tok / dt, 1), 'unit': 'speech-tokens/s', 'rtf': round(dt / c['audio_seconds'], 6), 'ms_per_step': round(1e3 * dt / args.sweep_steps, 1),
                                      'lm_slots': slots, 'decode_step_us': round(us, 1), 'decode_step_frac_of_hbm': round(by / us / 1e3 / HBM_PEAK_GBS, 4) if us else None,
                                      'mean_active_sequences': round(c['llm'].get('mean_active_sequences', 0.0), 1)}
    finally:
        pipe.llm.inference_head_num = k0
    return out


def run_acoustic(args, cfg, world, rank, lib):
    """BASELINE configs[4] on this rank's share: pre-tokenised speech-token streams (lengths U{352..2816}, seeded by the global stream index)
    through the flow decoder (10 Euler steps x CFG 2, padded solves of --acoustic-batch streams of similar length) and the vocoder.
    value = mel frames per second of the whole job; the roofline object is the DiT matrix work against the bf16 MFMA peak."""
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.flow import HvxFlow
    from flowmirror_hydravox_amd.hift import HvxHift
    from flowmirror_hydravox_amd.dp import gather_waveforms, shard_by_cost, acoustic_cost
    flow = HvxFlow(cfg.flow, W.make_flow_state(cfg.flow, seed=1987, init='normal02'), dtype=torch.bfloat16, max_t=2 * 2816 + 64)
    hift = HvxHift(cfg.hift, W.make_hift_state(cfg.hift, seed=1988, init='normal02'), exact_fp32=args.hift_exact)
    # the GLOBAL list of streams x world streams (global index = seed) dealt longest-first by cost a T + b T^2 (dp.acoustic_cost): every rank gets the same
    # amount of WORK, not the same count; one rank: all of them
    n_global = args.streams * world
    all_lens = [int(torch.randint(352, 2817, (1,), generator=torch.Generator().manual_seed(9_000_011 + i))) for i in range(n_global)]
    ids = shard_by_cost([acoustic_cost(2.0 * m) for m in all_lens], world)[rank]
    n = len(ids)
    lens = [all_lens[i] for i in ids]
    toks = [torch.randint(0, cfg.flow.vocab, (m,), generator=torch.Generator().manual_seed(i), dtype=torch.int32).cuda() for i, m in zip(ids, lens)]
    embs = [torch.randn(cfg.flow.spk_embed_dim, generator=torch.Generator().manual_seed(i)).cuda() for i in ids]
    order = sorted(range(n), key=lambda i: -lens[i])
    groups = [order[i:i + max(1, args.acoustic_batch)] for i in range(0, n, max(1, args.acoustic_batch))]      # length buckets: neighbours in the sorted order

    def run_all():
        wavs = [None] * n
        t_flow = t_hift = 0.0
        for g in groups:
            torch.cuda.synchronize()
            t0 = time.time()
            mels = flow.inference_batch([toks[i] for i in g], [embs[i] for i in g])
            torch.cuda.synchronize()
            t1 = time.time()
            for i, m in zip(g, mels):
                wavs[i] = hift.inference(speech_feat=m)[0][0]
            torch.cuda.synchronize()
            t_flow += t1 - t0
            t_hift += time.time() - t1
        return wavs, t_flow, t_hift

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        run_all()
    barrier()
    t0 = time.time()
    t_flow = t_hift = 0.0
    for _ in range(args.steps):
        wavs, a, b = run_all()
        got = gather_waveforms(wavs, ids, dst=0)
        t_flow, t_hift = t_flow + a, t_hift + b
    barrier()
    elapsed = time.time() - t0
    frames = float(sum(2 * m for m in lens)) * args.steps
    dit_flops = sum(7.56e9 * 2 * m + 1.80e6 * (2 * m) ** 2 for m in lens) * args.steps
    hift_flops = 672e6 * frames
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed, t_flow, t_hift], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_flow, t_hift = (float(v) for v in t)
        t = torch.tensor([frames, dit_flops, hift_flops], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        frames, dit_flops, hift_flops = (float(v) for v in t)
    if rank != 0:
        return
    ach = dit_flops / t_flow / 1e12 / world                     # per GPU, while the flow stage runs
    print(json.dumps({
        'metric': 'mel frames/sec, flow-matching (10 Euler steps x CFG 2) + HiFT vocoder only, pre-tokenised streams', 'value': round(frames / elapsed, 1),
        'unit': 'mel-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 2),
        'rccl_ranks': (torch.distributed.get_world_size() if world > 1 else 1), 'collective_backend': ((torch.distributed.get_backend() + ' (= RCCL on ROCm)') if world > 1 else None),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[4] slice: %d speech-token streams per GPU and step, lengths U{352..2816} (mean %.0f), flow bf16 (DiT 22 x 1024) in padded '
                               'solves of up to %d streams of neighbouring length, HiFT fp32 contract (decode convolutions as split-bf16 MFMA), seeded N(0,0.02) weights'
                               % (n, sum(lens) / len(lens), args.acoustic_batch), 'streams_per_gpu': n, 'parallelism': 'stream-dp%d' % world},
        'utterances_per_s': round(n * world * args.steps / elapsed, 2), 'audio_seconds_per_s': round(frames / 50.0 / elapsed, 1),
        'stage_seconds_per_step': {'flow': round(t_flow / args.steps, 4), 'hift': round(t_hift / args.steps, 4)},
        'roofline': {'kernel': 'DiT estimator (all bf16 GEMMs + attention of the flow stage)', 'bound': 'mfma', 'achieved': round(ach, 1), 'peak': MFMA_BF16_PEAK_TF,
                     'unit': 'TFLOP/s', 'frac': round(ach / MFMA_BF16_PEAK_TF, 4), 'traffic': None,
                     'algorithmic_flops': 'SURVEY.md §8(d): 7.56 GF*T + 1.80 MF*T^2 per stream of T frames, summed over the streams / wall time of the flow stage'},
        'roofline_other': [{'kernel': 'HiFT vocoder', 'bound': 'mfma', 'achieved': round(hift_flops / t_hift / 1e12 / world, 1), 'peak': round(MFMA_X3_PEAK_TF, 1), 'unit': 'TFLOP/s',
                            'frac': round(hift_flops / t_hift / 1e12 / world / MFMA_X3_PEAK_TF, 4),
                            'note': '672 MF of fp32-equivalent convolution per mel frame over the wall time of the whole vocoder stage; the decode convolutions run as 3 bf16 '
                                    'MFMAs per step (gemm_x3.hip): peak = dense bf16 peak / 3'}]}))


def main():
    args = parse()
    if args.serial:
        args.mode = 'serial'
    args.serial = args.mode == 'serial'
    if args.cpu_baseline_worker:
        from flowmirror_hydravox_amd.config import cv3_config, tiny_config
        print(json.dumps(cpu_baseline(tiny_config() if args.tiny else cv3_config(), 1986, args.chars, args.heads)))
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` starts its own ranks, one process per GPU over RCCL, as the reference starts one worker per GPU
        # (server/worker.py:104-127); under torch.distributed.run the environment is already there and this branch is skipped
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))).returncode)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or without torch.distributed.run)' % (args.gpus, world, args.gpus))
    stub = args.stub_pipeline
    if args.lib_opt and not stub:
        from flowmirror_hydravox_amd import _lib as _hvx_lib
        for kv in args.lib_opt:
            _hvx_lib.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    dev = 'cpu' if stub else 'cuda'
    sync = (lambda: None) if stub else torch.cuda.synchronize
    if world > 1:
        import torch.distributed as dist
        if stub:
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        assert dist.get_world_size() == args.gpus and dist.get_backend() == ('gloo' if stub else 'nccl')        # "nccl" is RCCL on ROCm
    elif not stub:
        torch.cuda.set_device(0)
    from flowmirror_hydravox_amd import _lib, cv3_config, tiny_config
    from flowmirror_hydravox_amd.pipeline import HvxPipeline, synthetic_utterance
    from flowmirror_hydravox_amd.dp import gather_waveforms, shard_by_cost, Handoff
    from flowmirror_hydravox_amd.sampling import ras_sampling
    from functools import partial
    if stub:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        from bench_stub import StubPipeline, StubLib
        lib = StubLib()
        args.no_cpu_baseline = args.no_fp32_mode = args.no_extras = True
    else:
        _lib.require_gpu()
        lib = _lib.load()

    cfg = tiny_config() if args.tiny else cv3_config()
    if args.tiny:
        # tiny_config's speech vocabulary (96 codes + 200 stop ids) makes an all-EOS nucleus a 2 % event per draw; with the generation length pinned
        # (EOS refused 100 times = error, llm_multi_head_v3.py:158-166) a toy job dies on it.  Same toy widths, a CV3-like share of stop ids:
        cfg.llm.speech_tokens = cfg.flow.vocab = 4096
    if args.config == 'acoustic':
        return run_acoustic(args, cfg, world, rank, lib)
    if args.config == 'stress' and '--heads' not in sys.argv and '--batch' not in sys.argv:
        args.heads, args.batch = 4, 32                        # BASELINE configs[2]: multi-head accept-rate stress (win_size 32, tau_r 0.2 are the defaults here)
    zero_shot = args.config == 'zero_shot'
    chars, B, K = args.chars, args.batch, args.heads
    if args.lm_slots <= 0:
        args.lm_slots = 128 if K == 1 else 64                 # a decode grid of 128 rows (64 x K for K >= 2: the widest the heads' buffers are sized for stays 256 rows)
    ratio = 5.5
    n_spk = int(chars * ratio)
    P_SPK, P_TXT = (75, 20) if zero_shot else (0, 0)           # configs[3]: 3 s prompt = 75 speech tokens / 150 mel frames, 20 prompt-text tokens

    def n_text_of(index):
        return chars if not zero_shot else int(torch.randint(64, 513, (1,), generator=torch.Generator().manual_seed(7_000_003 + index)))

    def make_utt(index):
        if not zero_shot:
            return synthetic_utterance(cfg, index, chars)
        return synthetic_utterance(cfg, index, n_text_of(index), n_prompt_speech=P_SPK, n_prompt_text=P_TXT)
    max_ctx = 2 + chars + P_TXT + P_SPK + n_spk + K + 32
    sampling = partial(ras_sampling, top_p=0.9, top_k=10, win_size=32, tau_r=0.2)
    t_build = time.time()
    if stub:
        pipe = StubPipeline(cfg, K)
    else:
        pipe = HvxPipeline(cfg, llm_dtype=torch.float32 if args.llm_dtype == 'fp32' else torch.bfloat16, flow_dtype=torch.bfloat16, max_batch=B, max_ctx=max_ctx, max_t=2 * (n_spk + P_SPK) + 64,
                           seed=1986, init='normal02', sampling=sampling, inference_head_num=K, hift_exact_fp32=args.hift_exact)
    pipe.acoustic_batch = max(1, args.acoustic_batch)
    pipe.lm_cus = args.lm_cus
    pipe.lm_cus_only = args.lm_cus_only
    if args.lm_cus_only > 0:
        pipe.llm.cu_range = (0, args.lm_cus_only)
    pipe.acoustic_cus = args.acoustic_cus
    if args.lm_cus > 0:
        pipe.llm.cu_range = (0, args.lm_cus)             # (before the first decode engine exists: the engine keeps its stream)
    t_build = time.time() - t_build
    utts = [make_utt(rank * B + i) for i in range(B)]
    gids = [rank * B + i for i in range(B)]

    def barrier():
        sync()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        sync()

    def step():
        wavs, st = pipe.synthesize(utts, max_token_text_ratio=ratio, min_token_text_ratio=ratio)
        got = gather_waveforms(wavs, gids, dst=0)
        return st, got

    # Warm-up steps run the three stages back to back (this is also where the un-overlapped stage times come from).  The timed
    # steps go through the software pipeline: the flow decoder + vocoder of step i run on a second stream while the LM decodes
    # step i+1; the pipeline is empty when the clock starts and drained before it stops, so the region holds exactly K whole steps.
    serial = None
    for _ in range(args.warmup):
        serial, got = step()
    # BASELINE configs[1] read strictly: ONE batch of B utterances, lock-step decode grid of B sequences, no utterance of another batch in
    # flight, stages back to back — its throughput and its latency (every utterance of the batch is complete after `latency_s`)
    strict = None
    if args.warmup > 0:
        sync()
        t_s = time.time()
        serial, got = step()
        sync()
        t_s = time.time() - t_s
        strict = {'value': round(serial.tokens / t_s, 1), 'unit': 'speech-tokens/s', 'rtf': round(t_s / serial.audio_seconds, 6), 'latency_s': round(t_s, 3),
                  'utterances_in_flight_per_gpu': B, 'schedule': 'one batch of %d, lock-step decode, stages back to back, no continuous batching (one un-timed extra step after the warm-up)' % B,
                  'stage_seconds': {'llm': round(serial.llm_seconds, 4), 'flow': round(serial.flow_seconds, 4), 'hift': round(serial.hift_seconds, 4)}}
    lm_alone = None
    if args.mode == 'continuous' and args.warmup > 0 and not stub:
        # the wide decode grid alone (nothing else on the GPU): --lm-slots utterances through the engine, untimed — `roofline.alone`
        reqs = [dict(text=utts[i % B].text, seed=10_000 + i, tag=i, max_token_text_ratio=ratio, min_token_text_ratio=ratio) for i in range(args.lm_slots)]
        for _ in pipe.llm.generate_stream(iter(reqs), n_slots=args.lm_slots):
            pass
        lm_alone = dict(pipe.llm.last_stats)
    barrier()
    stats = []
    cont = None
    t0 = time.time()
    if args.mode == 'serial':
        for _ in range(args.steps):
            st, got = step()
            stats.append(st)
    elif args.mode == 'chains':
        for wavs, st in pipe.synthesize_pipelined([utts] * args.steps, max_token_text_ratio=ratio, min_token_text_ratio=ratio,
                                                  lm_chains=args.lm_chains, acoustic_chains=args.acoustic_chains):
            got = gather_waveforms(wavs, gids, dst=0)
            stats.append(st)
    else:
        # K steps of B utterances each = K * B utterances through the continuous-batching engine; finished waveforms are handed to rank 0
        # one step's worth (B utterances) at a time, inside the timed region
        # the job is the GLOBAL list of steps x B x world utterances (global index = sampler seed), dealt longest-first across the ranks
        # (dp.shard_by_cost): with mixed lengths every rank gets the same amount of text, with equal lengths the deal is
        # round-robin; no rank ever needs another's utterances, so this is still weak scaling with no data-path collective
        n_global = args.steps * B * world
        shards = shard_by_cost([float(n_text_of(g)) for g in range(n_global)], world)
        mine = shards[rank]
        assert len(mine) == args.steps * B or zero_shot
        job = [make_utt(g) for g in mine]
        hand = Handoff(shards, B, dst=0, keep=stub)
        for i, wav, toks in pipe.synthesize_continuous(job, lm_slots=args.lm_slots, max_token_text_ratio=ratio, min_token_text_ratio=ratio,
                                                       acoustic_batch=args.acoustic_batch, acoustic_min_batch=args.acoustic_min_batch,
                                                       pace=[int(v) for v in args.lm_pace.split(',')] if args.lm_pace else None):
            hand.push(job[i].seed, wav)                # one step's worth (B utterances) per hand-off, inside the timed region
        got = hand.finish()
        assert rank != 0 or hand.n_received == n_global
        cont = dict(pipe.last_continuous)
    barrier()
    elapsed = time.time() - t0
    # Per-kernel durations: hipEvent brackets on the launch stream, every `prof_period`-th launch of each kernel class.
    # A hipGraph replay cannot be bracketed per kernel, so the brackets run over one more, identical step with the decode
    # graph disabled (same kernels, same launch geometry; the rocprofv3 --kernel-trace summary under profiles/ covers the
    # graph-replayed timed steps themselves and must agree).
    prof = {}
    if rank == 0 and not stub:
        lib.hvx_prof_enable(args.prof_period)
        step_prof_t0 = time.time()
        pipe.synthesize(utts, max_token_text_ratio=ratio, min_token_text_ratio=ratio)
        torch.cuda.synchronize()
        prof = read_prof(lib)
        lib.hvx_prof_enable(0)
    barrier()

    if cont is not None:
        tokens, audio, llm_s, flow_s, hift_s = cont['tokens'], cont['audio_seconds'], cont['llm_seconds'], cont['acoustic_seconds'], 0.0
    else:
        tokens = sum(s.tokens for s in stats)
        audio = sum(s.audio_seconds for s in stats)
        llm_s = sum(s.llm_seconds for s in stats)
        flow_s = sum(s.flow_seconds for s in stats)
        hift_s = sum(s.hift_seconds for s in stats)
        if not args.serial:                          # overlapped: only the sum of the two acoustic stages is a wall time
            flow_s, hift_s = flow_s + hift_s, 0.0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed, float(tokens), audio, llm_s, flow_s, hift_s], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        tokens, audio = float(tsum[1]), float(tsum[2])
        llm_s, flow_s, hift_s = float(tmax[3]), float(tmax[4]), float(tmax[5])
    if rank != 0:
        return
    assert world > 1 or len(got) == B or cont is not None
    in_flight = {'continuous': args.lm_slots + args.acoustic_batch, 'chains': B * (max(args.lm_chains, args.acoustic_chains) + 1), 'serial': B}[args.mode]

    # The dominant "kernel" is the decode step: one hipGraph replay of forward + sampler + advance (~160 launches of 5-9 us, which a
    # per-launch event bracket would distort), timed over the timed region itself by hipEvents around every block of 8 replays on
    # the decode stream.  Its algorithmic bytes are every LM weight once plus the cached K/V rows (HvxLLM.decode_step_bytes).
    def step_roofline(sts):
        sts = [x.llm for x in sts if x.llm.get('decode_steps_timed')]
        if not sts:
            return None
        n = sum(x['decode_steps_timed'] for x in sts)
        us = sum(x['decode_step_us'] * x['decode_steps_timed'] for x in sts) / n
        by = sum(x['decode_step_bytes'] * x['decode_steps_timed'] for x in sts) / n
        ach = by / us / 1e3
        return dict(achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4), avg_launch_us=round(us, 1), algorithmic_bytes_per_launch=round(by),
                    launches=n)
    class _S:                                        # (engine statistics in the shape step_roofline reads)
        def __init__(self, d):
            self.llm = d
    rl_timed = step_roofline(stats if cont is None else [_S(cont['llm'])])
    rl_alone = step_roofline([_S(lm_alone)] if lm_alone else ([serial] if serial is not None else []))
    grid_seqs = cont['llm'].get('mean_active_sequences', B) if cont is not None else B
    est = sorted(((p['est_total_ms'], n) for n, p in prof.items() if n not in ('llm_decode_gemm', 'llm_attention', 'ras_sampler')), reverse=True)
    line = {
        'metric': 'speech-tokens/sec + RTF, HydraVox-CV3 head_num=%d, %s' % (K, '%d-char batch' % chars if not zero_shot else 'zero-shot mixed-length batch'),
        'value': round(tokens / elapsed, 2), 'unit': 'speech-tokens/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 2),
        # ranks that took part in the collective group of THIS run (torch.distributed over RCCL: backend "nccl"), so that a scaling record can be checked for N ranks
        'rccl_ranks': (torch.distributed.get_world_size() if world > 1 else 1), 'collective_backend': ((torch.distributed.get_backend() + ('' if stub else ' (= RCCL on ROCm)')) if world > 1 else None),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if args.llm_dtype == 'bf16' else 'f32 (LM) + bf16 (flow)', 'data': 'synthetic',
        'config': {'workload': ('HydraVox-CV3%s inference_head_num=%d, batch=%dx%d-char utterances per GPU (' + ('zero-shot: text lengths U{64..512} per utterance, ' if zero_shot else '') + '%d text -> %d speech tokens -> %d mel frames '
                               'each' + (' AT THE MAXIMUM LENGTH' if zero_shot else '') + '), %s / hift split-bf16 (fp32 operands as (hi, lo) bf16 pairs, 3 MFMAs per step: ~16 mantissa bits; `exact_vocoder` = the exact fp32 form), '
                               'llm->flow->hift end to end, seeded N(0,0.02) weights')
                               % (' [STUB PIPELINE ON CPU / GLOO - RANK LOGIC ONLY, NOT A BENCHMARK]' if stub else ' [TINY DIMS - NOT A BENCHMARK]' if args.tiny else '', K, B, chars, chars, n_spk, 2 * n_spk,
                                  'llm+flow bf16' if args.llm_dtype == 'bf16' else 'llm fp32 (speech-token ids bit-exact against the reference) + flow bf16'),
                   'baseline_config': {'tts': 'configs[1]', 'stress': 'configs[2]', 'zero_shot': 'configs[3]: text lengths U{64..512} per utterance (the chars figure above is the maximum), '
                                       'prompt = 75 speech tokens + 150 mel frames + 20 prompt-text tokens'}[args.config],
                   'global_batch': B * world, 'parallelism': 'utterance-dp%d' % world, 'lib_options': args.lib_opt or None,
                   'schedule': {'continuous': 'continuous batching: one decode grid of %d slots, utterances of later steps join as earlier ones finish; '
                                              'flow + vocoder of finished utterances run beside it' % args.lm_slots,
                                'chains': '%d independent decode chains of one step each + %d acoustic chain(s)' % (args.lm_chains, args.acoustic_chains),
                                'serial': 'stages back to back'}[args.mode],
                   'utterances_in_flight_per_gpu': in_flight,
                   'cu_partition': ('decode engine on %d CUs, acoustic stage on the other %d' % (args.lm_cus, lib.hvx_device_ok() - args.lm_cus)) if args.lm_cus > 0 and args.mode == 'continuous' else ('decode engine confined to %d CUs, acoustic stage on all' % args.lm_cus_only) if args.lm_cus_only > 0 and args.mode == 'continuous' else 'none (both stages share all CUs)',
                   'flow_arithmetic': ('bf16 attention operands; fp16 residual stream%s%s' % (
                       ', fp16 block Linear operands' if getattr(pipe.flow, 'f16_linears', False) else ', bf16 block Linear operands',
                       ', fp32 time MLP / adaLN modulation / input + output projection (the reference deploys this decoder in fp16: this mode sits at its fp16 run\'s distance from fp32, tests/test_gpu_cv3d.py)'
                       if getattr(pipe.flow, 'f32_small', False) else ', bf16 small Linears')) if getattr(pipe.flow, 'half_stream', False) else 'bf16 operands, fp32 residual stream',
                   'sampling': {'top_p': 0.9, 'top_k': 10, 'win_size': 32, 'tau_r': 0.2}},
        'rtf': round(elapsed / audio, 6) if audio else None,
        'llm_tokens_per_s': round(tokens / llm_s, 2) if llm_s else None,
        'stage_seconds_per_step': ({'llm': round(llm_s / args.steps, 4), 'flow': round(flow_s / args.steps, 4), 'hift': round(hift_s / args.steps, 4)} if args.serial else
                                   {'llm': round(llm_s / args.steps, 4), 'flow+hift': round(flow_s / args.steps, 4),
                                    'overlap': ('llm = wall time of the decode engine / steps, flow+hift = busy time of the acoustic stage / steps; they run beside each other'
                                                if args.mode == 'continuous' else
                                                'flow+hift of step i runs beside the llm of the next %d step(s); llm = mean decode wall time of a step while %d decode at once' % (args.lm_chains, args.lm_chains))}),
        'stage_seconds_serial': None if serial is None else {'llm': round(serial.llm_seconds, 4), 'flow': round(serial.flow_seconds, 4), 'hift': round(serial.hift_seconds, 4)},
        'audio_seconds_per_step': round(audio / args.steps, 2),
        'setup_seconds': round(t_build, 1),
    }
    if cont is not None:
        line['llm_engine'] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in cont['llm'].items()
                              if k in ('steps', 'requests', 'prefill_and_setup_seconds', 'device_idle_ms_between_blocks', 'mean_active_sequences', 'mean_ctx', 'seconds')}
    if rl_timed:
        line['roofline'] = dict(kernel='llm_decode_step (hipGraph: %d-layer backbone + %d MTP heads + sampler + advance, one launch = one step of a grid of %.1f live sequences x %d heads)'
                                       % (cfg.llm.layers, K, grid_seqs, K), bound='hbm', achieved=rl_timed['achieved'], peak=HBM_PEAK_GBS, unit='GB/s',
                                frac=rl_timed['frac'], traffic=pmc_traffic('llm_decode_step'), traffic_counted_on=pmc_traffic('llm_decode_step', with_grid=True)[1],
                                avg_launch_us=rl_timed['avg_launch_us'],
                                algorithmic_bytes_per_launch=rl_timed['algorithmic_bytes_per_launch'], launches_per_timed_region=rl_timed['launches'],
                                measured='hipEvents around every 8 replays on the decode stream, all timed steps' +
                                         ('' if args.serial else '; the flow decoder + vocoder of finished utterances share the GPU meanwhile' if args.mode == 'continuous' else
                                          '; the flow decoder + vocoder of an earlier step and the decode chain of %d other step(s) share the GPU meanwhile' % (args.lm_chains - 1)))
        # what the LM stage moves over the WHOLE timed region (every decode launch of the job) against the HBM peak: the pipeline-wide figure
        line['roofline']['pipeline_wide'] = dict(achieved=round(rl_timed['algorithmic_bytes_per_launch'] * rl_timed['launches'] / elapsed / 1e9, 1), unit='GB/s',
                                                 frac=round(rl_timed['algorithmic_bytes_per_launch'] * rl_timed['launches'] / elapsed / 1e9 / HBM_PEAK_GBS, 4),
                                                 note='algorithmic bytes of all decode launches / wall time of the timed region (the acoustic stage owns most of the GPU time)')
        if args.mode == 'chains' and args.lm_chains > 1:
            # launches of different steps overlap in time: per launch `achieved` is bytes / its own duration; the chains together move this much
            line['roofline']['concurrent_decode_chains'] = args.lm_chains
            line['roofline']['achieved_all_chains'] = round(rl_timed['achieved'] * args.lm_chains, 1)
        if rl_alone and not args.serial:
            line['roofline']['alone'] = dict(achieved=rl_alone['achieved'], frac=rl_alone['frac'], avg_launch_us=rl_alone['avg_launch_us'],
                                             algorithmic_bytes_per_launch=rl_alone['algorithmic_bytes_per_launch'],
                                             measured='the same grid of %d sequences with nothing else on the GPU (untimed run before the timed region)' % args.lm_slots
                                                      if lm_alone else 'same brackets in the warm-up step (stages back to back, nothing else on the GPU)')
    if est:
        for n in prof:
            prof[n]['steps'], prof[n]['period'] = args.steps, args.prof_period
        line['roofline_other'] = [roofline_of(n, prof[n]) for _, n in est if prof[n]['work_per_launch'] > 0]
        line['kernel_time_share_ms'] = {n: round(t, 1) for t, n in est}
    if strict is not None:
        line['strict_batch%d' % B] = strict
    extras = world == 1 and not args.no_extras and args.config == 'tts' and args.mode == 'continuous' and args.llm_dtype == 'bf16'
    if extras:
        try:
            line['single_request'] = single_request(pipe, cfg, ratio)
        except Exception as e:
            line['single_request'] = {'error': repr(e)}
        try:
            line['head_sweep'] = head_sweep(pipe, args, make_utt, ratio, K, B)
        except Exception as e:
            line['head_sweep'] = {'error': repr(e)}
        try:
            # the decode step of a ONE-sequence grid (every single request; configs[1] read literally sits at 8 x 2 rows on the same launch floor): a roofline entry of its own
            sr = line['single_request']['head_num_1']
            by = pipe.llm.decode_step_bytes(1, 1, 2 + 64 + 176)                       # mean context of a 64-char request: prefix 66 + half of its 352 tokens
            line.setdefault('roofline_other', []).append(dict(
                kernel='llm_decode_small_grid (hipGraph step of ONE sequence x 1 head: 6 dependent launches x 24 layers + head + sampler)', bound='hbm',
                achieved=round(by / sr['decode_step_us'] / 1e3, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(by / sr['decode_step_us'] / 1e3 / HBM_PEAK_GBS, 4), traffic=None,
                avg_launch_us=sr['decode_step_us'], algorithmic_bytes_per_launch=round(by), launches_per_timed_region=0,
                measured='hipEvents around blocks of decode-graph replays inside `single_request` (untimed extra): launch-latency-bound, DESIGN.md §8'))
        except Exception as e:
            line.setdefault('roofline_other', []).append({'kernel': 'llm_decode_small_grid', 'error': repr(e)})
    del pipe
    if stub:
        line['stub'] = {'received_on_rank0': hand.n_received if cont is not None else len(got), 'handoff_rounds': hand.rounds if cont is not None else None,
                        'shard_sizes': [len(sh) for sh in shards] if cont is not None else None,
                        'shard_text': [sum(n_text_of(g) for g in sh) for sh in shards] if cont is not None else None,
                        'checksum': float(sum(float(w.double().sum()) for w in got.values())) if got else 0.0}
        print(json.dumps(line))
        return
    torch.cuda.empty_cache()
    if world == 1 and not args.no_fp32_mode and not args.tiny:
        # the parity-exact mode (fp32 LM + fp32 flow: speech-token ids bit-exact against the reference, mel / waveform within 1e-3): the
        # same batch, strict schedule, one un-timed warm-up on a single utterance and one measured step
        try:
            pipe32 = HvxPipeline(cfg, llm_dtype=torch.float32, flow_dtype=torch.float32, max_batch=B, max_ctx=max_ctx, max_t=2 * (n_spk + P_SPK) + 64,
                                 seed=1986, init='normal02', sampling=sampling, inference_head_num=K)
            pipe32.acoustic_batch = max(1, args.acoustic_batch)
            pipe32.synthesize(utts[:1], max_token_text_ratio=ratio, min_token_text_ratio=ratio)
            torch.cuda.synchronize()
            t32 = time.time()
            _, st32 = pipe32.synthesize(utts, max_token_text_ratio=ratio, min_token_text_ratio=ratio)
            torch.cuda.synchronize()
            t32 = time.time() - t32
            line['fp32_mode'] = {'value': round(st32.tokens / t32, 1), 'unit': 'speech-tokens/s', 'rtf': round(t32 / st32.audio_seconds, 6), 'latency_s': round(t32, 3),
                                 'dtype': 'f32 (LM and flow on the exact fp32 MFMA forms; vocoder as in the headline)',
                                 'schedule': 'one batch of %d, lock-step decode, stages back to back: compare with strict_batch%d' % (B, B),
                                 'stage_seconds': {'llm': round(st32.llm_seconds, 4), 'flow': round(st32.flow_seconds, 4), 'hift': round(st32.hift_seconds, 4)}}
            del pipe32
            torch.cuda.empty_cache()
        except Exception as e:
            line['fp32_mode'] = {'value': None, 'error': repr(e)}
    if world == 1 and not args.no_fp32_mode and not args.tiny and args.mode == 'continuous' and args.llm_dtype == 'bf16' and args.config == 'tts':
        # the mode that meets north_star's parity clause as written — speech-token ids BIT-EXACT against the reference (LM on the exact fp32 forms) with
        # the flow decoder at the reference's own fp16 precision and the vocoder as in the headline — through the SAME continuous engine and the same
        # number of steps (not part of the line's `value`; the clock below is its own)
        try:
            pipex = HvxPipeline(cfg, llm_dtype=torch.float32, flow_dtype=torch.bfloat16, max_batch=B, max_ctx=max_ctx, max_t=2 * (n_spk + P_SPK) + 64,
                                seed=1986, init='normal02', sampling=sampling, inference_head_num=K)
            pipex.acoustic_batch = max(1, args.acoustic_batch)
            pipex.synthesize(utts[:1], max_token_text_ratio=ratio, min_token_text_ratio=ratio)
            n_x = min(args.steps, 12)                 # (a secondary figure: 96 utterances through the 64-slot grid)
            jobx = [make_utt(g) for g in range(n_x * B)]
            torch.cuda.synchronize()
            tx = time.time()
            tokx = 0
            for i, wav, toks in pipex.synthesize_continuous(jobx, lm_slots=args.lm_slots, max_token_text_ratio=ratio, min_token_text_ratio=ratio,
                                                            acoustic_batch=args.acoustic_batch, acoustic_min_batch=args.acoustic_min_batch):
                tokx += len(toks)
            torch.cuda.synchronize()
            tx = time.time() - tx
            cx = dict(pipex.last_continuous)
            if extras:
                # north_star / SURVEY.md §7: "ids bit-exact in an fp32 GPU mode ... reported as match-rate in bf16 mode": the bf16 LM of the headline
                # teacher-forced on the fp32 LM's streams of the bench batch (same history, same noise position per decision)
                try:
                    from flowmirror_hydravox_amd import weights as W
                    from flowmirror_hydravox_amd.llm import HvxLLM, decision_agreement
                    t_a = time.time()
                    llm_b = HvxLLM(cfg.llm, W.make_llm_state(cfg.llm, seed=1986, init='normal02'), dtype=torch.bfloat16, max_batch=B, max_ctx=max_ctx, sampling=sampling,
                                   inference_head_num=K)
                    reqs = [dict(text=u.text, seed=u.seed, max_token_text_ratio=ratio, min_token_text_ratio=ratio) for u in utts]
                    ag = decision_agreement(pipex.llm, llm_b, reqs)
                    line['bf16_id_agreement'] = dict(ag, seconds=round(time.time() - t_a, 1),
                                                     what='teacher-forced per-decision agreement of the bf16 LM (the headline\'s) with the fp32 LM (ids bit-exact against the reference) on the '
                                                          '%d utterances of one bench step: every draw of the bf16 LM is made from ITS log-probs with the fp32 stream\'s history, repetition '
                                                          'window and noise position; `agreement` = equal draws / draws, `steps_all_equal` = steps whose %d draws are all equal' % (B, K))
                    del llm_b
                except Exception as e:
                    line['bf16_id_agreement'] = {'error': repr(e)}
            line['ids_exact_mode'] = {'value': round(tokx / tx, 1), 'unit': 'speech-tokens/s', 'rtf': round(tx / cx['audio_seconds'], 6), 'steps': n_x, 'ms_per_step': round(1e3 * tx / n_x, 2),
                                      'dtype': 'f32 LM (speech-token ids bit-exact against the reference: tests/test_gpu_cv3w.py, test_gpu_cv3d.py) + the flow decoder and '
                                               'vocoder of the headline (mel within the reference\'s own fp16 distance of fp32)',
                                      'schedule': 'the continuous engine of the headline, %d slots, %d utterances' % (args.lm_slots, n_x * B),
                                      'decode_step_us': round(cx['llm'].get('decode_step_us', 0.0), 1)}
            del pipex
            torch.cuda.empty_cache()
        except Exception as e:
            line['ids_exact_mode'] = {'value': None, 'error': repr(e)}
    if world == 1 and not args.no_fp32_mode and not args.tiny and args.mode == 'continuous' and args.llm_dtype == 'bf16' and args.config == 'tts' and not args.hift_exact:
        # the headline's job with the vocoder on the EXACT fp32 MFMA forms (hvx_hift_config.exact_fp32 = the reference's own fp32 vocoder arithmetic, 2.8e-5 of the
        # reference's waveform at 5632 frames against 1.5e-4 for the split-bf16 default: tests/test_gpu_refpin.py), through the same engine, its own clock
        try:
            pipev = HvxPipeline(cfg, llm_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, max_batch=B, max_ctx=max_ctx, max_t=2 * (n_spk + P_SPK) + 64,
                                seed=1986, init='normal02', sampling=sampling, inference_head_num=K, hift_exact_fp32=True)
            pipev.acoustic_batch = max(1, args.acoustic_batch)
            _, stv = pipev.synthesize(utts, max_token_text_ratio=ratio, min_token_text_ratio=ratio)        # warm-up + the un-overlapped stage times
            n_v = min(args.steps, 8)
            jobv = [make_utt(g) for g in range(n_v * B)]
            torch.cuda.synchronize()
            tv = time.time()
            tokv = 0
            for i, wav, toks in pipev.synthesize_continuous(jobv, lm_slots=args.lm_slots, max_token_text_ratio=ratio, min_token_text_ratio=ratio,
                                                            acoustic_batch=args.acoustic_batch, acoustic_min_batch=args.acoustic_min_batch):
                tokv += len(toks)
            torch.cuda.synchronize()
            tv = time.time() - tv
            line['exact_vocoder'] = {'value': round(tokv / tv, 1), 'unit': 'speech-tokens/s', 'steps': n_v, 'ms_per_step': round(1e3 * tv / n_v, 2),
                                     'hift_seconds': round(stv.hift_seconds, 4), 'hift_seconds_split_bf16': None if serial is None else round(serial.hift_seconds, 4),
                                     'what': 'the headline job with HvxHift(exact_fp32=True): every vocoder convolution on v_mfma_f32_16x16x4_f32 (the reference\'s fp32 arithmetic); '
                                             'hift_seconds = the vocoder stage of one batch of %d utterances, stages back to back' % B}
            del pipev
            torch.cuda.empty_cache()
        except Exception as e:
            line['exact_vocoder'] = {'value': None, 'error': repr(e)}
    if world == 1 and not args.tiny and args.config == 'tts' and args.mode == 'continuous' and args.llm_dtype == 'bf16':
        # north_star's parity clause (ids bit-exact; mel / waveform 1e-3) against what each measured mode delivers, in ONE place.  The distances are what
        # the GPU tests assert against the REFERENCE's own outputs at these shapes (tests/test_gpu_refpin.py, test_gpu_cv3d.py), not measured in this run.
        def _v(k):
            return (line.get(k) or {}).get('value')
        line['parity_modes'] = {
            'headline (bf16 LM, reference-fp16-precision flow, split-bf16 vocoder)': {
                'speech_tokens_per_s': line['value'], 'ids_vs_reference': 'not bit-exact: teacher-forced per-decision agreement %s' % (
                    (line.get('bf16_id_agreement') or {}).get('agreement')), 'mel_vs_reference': 1.5e-3, 'vocoder_stage_vs_reference': 1.5e-4},
            'ids_exact_mode (fp32 LM, same flow / vocoder)': {
                'speech_tokens_per_s': _v('ids_exact_mode'), 'ids_vs_reference': 'bit-exact (the 64-slot x 2-head grid of this mode beside 63 live sequences at contexts > 1024, and '
                'batch-invariant over a whole 1408-step stream: tests/test_gpu_refpin.py)', 'mel_vs_reference': 1.5e-3, 'vocoder_stage_vs_reference': 1.5e-4},
            'exact_vocoder (headline + exact fp32 vocoder)': {'speech_tokens_per_s': _v('exact_vocoder'), 'vocoder_stage_vs_reference': 2.8e-5},
            'fp32_mode (fp32 LM + fp32 flow, strict batch of 8)': {
                'speech_tokens_per_s': _v('fp32_mode'), 'ids_vs_reference': 'bit-exact', 'mel_vs_reference': 1.1e-6, 'vocoder_stage_vs_reference': 1.5e-4},
            'note': 'the 19-20 k tokens/s of `value` and "ids bit-exact" never hold in the same run; end-to-end waveforms are held to the reference\'s own conditioning band '
                    '(tests/golden/hift_full_cond.npz: the reference against itself under 1-ulp f0 perturbations), stage-wise numbers above'}
    if world == 1 and not args.tiny and not stub and args.config == 'tts' and args.config_steps > 0 and not args.no_extras:
        # the other BASELINE configs on this GPU, behind the headline: one fresh process each (their pipelines differ: 4 heads x 32 slots, prompts, no LM), compact summaries
        import subprocess
        line['configs'] = {}
        for name, extra in (('stress', []), ('zero_shot', []), ('acoustic', [])):
            try:
                cmd = [sys.executable, os.path.abspath(__file__), '--config', name, '--steps', str(args.config_steps), '--warmup', '1', '--no-cpu-baseline', '--no-fp32-mode',
                       '--no-extras', '--config-steps', '0'] + extra
                t_c = time.time()
                out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
                js = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
                d = json.loads(js[-1])
                line['configs'][name] = {'baseline_config': {'stress': 'configs[2]', 'zero_shot': 'configs[3] (one GPU of its 8)', 'acoustic': 'configs[4] (one GPU of its 8)'}[name],
                                         'metric': d.get('metric'), 'value': d.get('value'), 'unit': d.get('unit'), 'steps': d.get('steps'), 'ms_per_step': d.get('ms_per_step'),
                                         'rtf': d.get('rtf'), 'workload': (d.get('config') or {}).get('workload'), 'seconds': round(time.time() - t_c, 1),
                                         'roofline_frac': (d.get('roofline') or {}).get('frac')}
            except Exception as e:
                line['configs'][name] = {'value': None, 'error': repr(e)}
    if world == 1 and not args.no_cpu_baseline:
        try:
            # separate process with a hard wall-clock limit: the GPU line must be printed whatever the host cores do
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--chars', str(chars), '--heads', str(K)] + (['--tiny'] if args.tiny else [])
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=int(os.environ.get('HVX_CPU_BASELINE_TIMEOUT', '240')),
                                 env=dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES=''))
            line['cpu_baseline'] = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:                    # the bench line must still be printed
            line['cpu_baseline'] = {'value': None, 'unit': 'speech-tokens/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': 'failed: %r' % (e,)}
    print(json.dumps(line))


if __name__ == '__main__':
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()       # (a rank that leaves with the group alive aborts in the backend's watchdog thread)
