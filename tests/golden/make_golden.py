#!/usr/bin/env python3
"""Mint golden vectors by running the REFERENCE implementation (build container only).

    python tests/golden/make_golden.py          # writes tests/golden/*.npz

Imports the reference's own modules from /root/reference (through tests/golden/_ref_shim.py),
instantiates them at toy dimensions, loads seeded synthetic checkpoints produced by
flowmirror_hydravox_amd.weights (asserting that our checkpoint spec equals the reference
modules' state_dict keys/shapes) and records inputs + reference outputs.  The fixtures hold data
only (inputs, seeds, expected outputs); weights are re-generated from the recorded seed and guarded
by a checksum.  Nothing under /root/reference is read by the tests themselves.
"""
import os
import sys
import hashlib
from functools import partial

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, ROOT]

import _ref_shim  # noqa: E402

DictConfig = _ref_shim.install()

from flowmirror_hydravox_amd.config import tiny_config, cv3w_config  # noqa: E402
from flowmirror_hydravox_amd import weights as W  # noqa: E402
from oracle import sampler_ref, llm_ref, flow_ref, hift_ref  # noqa: E402

torch.set_grad_enabled(False)


def state_checksum(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().contiguous().numpy().tobytes())
    return h.hexdigest()


def assert_spec(module, spec, what):
    ref = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    mine = {k: tuple(s) for k, s, _ in spec}
    assert ref == mine, (what, set(ref) ^ set(mine), [(k, ref[k], mine[k]) for k in ref if k in mine and ref[k] != mine[k]])
    print('[spec] %s: %d keys identical to the reference state_dict' % (what, len(ref)))


# ------------------------------------------------------------------------------------------------
# sampler
# ------------------------------------------------------------------------------------------------
def gen_sampler():
    from cosyvoice.utils.common import ras_sampling
    from cosyvoice.llm.llm_multi_head_v3 import CosyVoice3LM

    class _Stub:                       # only what sampling_ids touches
        pass

    g = torch.Generator()
    g.manual_seed(20260929)
    cases = []

    def make_case(V, Vs, kind, i):
        top_k = [10, 25, 5, 1, 50][i % 5]
        top_p = [0.9, 0.8, 0.5, 0.99, 1.0][(i // 5) % 5]
        win = [32, 24, 10, 4][(i // 3) % 4]
        tau = [0.2, 0.1, 0.5, 0.05][(i // 7) % 4]
        x = torch.randn(V, generator=g)
        if kind == 'peaked':           # low entropy -> the repetition window triggers the fallback
            x = x * 0.3
            hot = torch.randint(0, Vs, (3,), generator=g)
            x[hot] += torch.tensor([9.0, 8.0, 7.5])
        elif kind == 'eos':            # most mass on stop ids -> EOS rejection loop
            x = x * 0.5
            x[Vs:] += (4.0 if i % 3 == 0 else 1.0)
        elif kind == 'ties':           # exact ties: stable sort must keep the lower index first
            x = torch.round(x * 2) / 2
        elif kind == 'flat':
            x = x * 0.01
        logp = x.log_softmax(dim=0)
        hlen = int(torch.randint(0, 80, (1,), generator=g))
        if kind == 'peaked':
            top = int(logp.argmax())
            hist = [top if torch.rand(1, generator=g).item() < 0.6 else int(torch.randint(0, Vs, (1,), generator=g)) for _ in range(hlen)]
        else:
            hist = torch.randint(0, Vs, (hlen,), generator=g).tolist()
        ignore_eos = bool(i % 2 == 0) if kind != 'eos' else True
        return dict(logp=logp.numpy(), hist=hist, top_k=top_k, top_p=top_p, win=win, tau=tau,
                    ignore_eos=ignore_eos, seed=1000 + len(cases), Vs=Vs)

    kinds = ['plain', 'peaked', 'eos', 'ties', 'flat']
    for i in range(240):
        cases.append(make_case(296, 96, kinds[i % 5], i))
    for i in range(40):
        cases.append(make_case(6761, 6561, kinds[i % 5], i))

    n_fallback = n_retry = 0
    for c in cases:
        stub = _Stub()
        stub.speech_token_size = c['Vs']
        stub.sampling = partial(ras_sampling, top_p=c['top_p'], top_k=c['top_k'], win_size=c['win'], tau_r=c['tau'])
        torch.manual_seed(c['seed'])
        logp_t = torch.from_numpy(c['logp'])
        try:
            ref_id = CosyVoice3LM.sampling_ids(stub, logp_t, list(c['hist']), 25, ignore_eos=c['ignore_eos'])
        except RuntimeError:           # 'sampling reaches max_trials 100 and still get eos' (llm_multi_head_v3.py:164-165)
            ref_id = -1
        probe = torch.empty(1).exponential_(1.0).item()          # next value of the global stream
        ns = sampler_ref.NoiseStream(seed=c['seed'])
        try:
            ora_id = sampler_ref.sampling_ids(c['logp'], list(c['hist']), ns, c['Vs'], c['ignore_eos'], top_p=c['top_p'],
                                              top_k=c['top_k'], win_size=c['win'], tau_r=c['tau'])
        except RuntimeError:
            ora_id = -1
        assert int(ref_id) == ora_id, (c['seed'], ref_id, ora_id)
        assert abs(float(ns.peek(ns.cursor, 1)[0]) - probe) == 0.0, 'noise consumption differs from the reference'
        c['id'] = int(ref_id)
        c['consumed'] = ns.cursor
        n_fallback += ns.cursor > 296 if c['Vs'] == 96 else ns.cursor > 6761
        n_retry += 1 if ns.cursor > c['top_k'] + len(c['logp']) else 0
    n_err = sum(1 for c in cases if c['id'] < 0)
    print('[sampler] %d cases, %d hit the RAS fallback, %d needed EOS retries, %d exhausted max_trials; oracle == reference id-for-id'
          % (len(cases), n_fallback, n_retry, n_err))

    def pack(sub):
        H = max(len(c['hist']) for c in sub)
        hist = -np.ones((len(sub), max(H, 1)), dtype=np.int32)
        for r, c in enumerate(sub):
            hist[r, :len(c['hist'])] = c['hist']
        return dict(
            logp=np.stack([c['logp'] for c in sub]).astype(np.float32), hist=hist,
            hist_len=np.array([len(c['hist']) for c in sub], dtype=np.int32),
            top_k=np.array([c['top_k'] for c in sub], dtype=np.int32), top_p=np.array([c['top_p'] for c in sub], dtype=np.float64),
            win=np.array([c['win'] for c in sub], dtype=np.int32), tau=np.array([c['tau'] for c in sub], dtype=np.float64),
            ignore_eos=np.array([c['ignore_eos'] for c in sub], dtype=np.int32), seed=np.array([c['seed'] for c in sub], dtype=np.int64),
            Vs=np.array([c['Vs'] for c in sub], dtype=np.int32), id=np.array([c['id'] for c in sub], dtype=np.int32),
            consumed=np.array([c['consumed'] for c in sub], dtype=np.int64))

    small = pack([c for c in cases if c['Vs'] == 96])
    big = pack([c for c in cases if c['Vs'] == 6561])
    np.savez_compressed(os.path.join(HERE, 'sampler_small.npz'), **small)
    np.savez_compressed(os.path.join(HERE, 'sampler_big.npz'), **big)


def gen_sampler_many():
    """1000 further sampler cases (760 at V = 296, 240 at V = 6761: plain / peaked / EOS-heavy / exact ties / flat, RAS fallbacks, EOS retries,
    max_trials exhaustion).  The INPUTS are not stored: tests/golden/sampler_cases.py regenerates case i from its own seed; the fixture holds
    the reference's id, the number of noise values it consumed and a checksum of the regenerated log-probabilities."""
    import hashlib
    from cosyvoice.utils.common import ras_sampling
    from cosyvoice.llm.llm_multi_head_v3 import CosyVoice3LM
    from sampler_cases import make_case, N_CASES

    class _Stub:
        pass
    ids, consumed, shas = [], [], []
    n_fallback = n_retry = n_err = 0
    for i in range(N_CASES):
        c = make_case(i)
        stub = _Stub()
        stub.speech_token_size = c['Vs']
        stub.sampling = partial(ras_sampling, top_p=c['top_p'], top_k=c['top_k'], win_size=c['win'], tau_r=c['tau'])
        torch.manual_seed(c['seed'])
        try:
            ref_id = int(CosyVoice3LM.sampling_ids(stub, torch.from_numpy(c['logp']), list(c['hist']), 25, ignore_eos=c['ignore_eos']))
        except RuntimeError:
            ref_id = -1
        probe = torch.empty(1).exponential_(1.0).item()
        ns = sampler_ref.NoiseStream(seed=c['seed'])
        try:
            ora_id = sampler_ref.sampling_ids(c['logp'], list(c['hist']), ns, c['Vs'], c['ignore_eos'], top_p=c['top_p'], top_k=c['top_k'],
                                              win_size=c['win'], tau_r=c['tau'])
        except RuntimeError:
            ora_id = -1
        assert ref_id == ora_id, (i, ref_id, ora_id)
        assert abs(float(ns.peek(ns.cursor, 1)[0]) - probe) == 0.0, 'noise consumption differs from the reference'
        V = len(c['logp'])
        n_fallback += ns.cursor > V
        n_retry += ns.cursor > c['top_k'] + V
        n_err += ref_id < 0
        ids.append(ref_id)
        consumed.append(ns.cursor)
        shas.append(int.from_bytes(hashlib.sha256(c['logp'].tobytes()).digest()[:8], 'little', signed=True))
    print('[sampler-many] %d cases (%d at V = 6761), %d hit the RAS fallback, %d needed EOS retries, %d exhausted max_trials; oracle == reference id-for-id'
          % (N_CASES, sum(1 for i in range(N_CASES) if make_case(i, header_only=True)['Vs'] == 6561), n_fallback, n_retry, n_err))
    np.savez_compressed(os.path.join(HERE, 'sampler_many.npz'), id=np.array(ids, dtype=np.int32), consumed=np.array(consumed, dtype=np.int64),
                        logp_sha=np.array(shas, dtype=np.int64))


# ------------------------------------------------------------------------------------------------
# LLM
# ------------------------------------------------------------------------------------------------
def build_ref_llm(cfg, sd, sampling):
    from transformers import Qwen2ForCausalLM
    from transformers.models.qwen2.configuration_qwen2 import Qwen2Config
    from cosyvoice.llm.llm_multi_head_v3 import CosyVoice3LM, Qwen2Encoder
    from cosyvoice.utils.common import ras_sampling
    qc = Qwen2Config(vocab_size=cfg.text_vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter,
                     num_hidden_layers=cfg.layers, num_attention_heads=cfg.q_heads, num_key_value_heads=cfg.kv_heads,
                     rope_theta=cfg.rope_theta, rms_norm_eps=cfg.rms_eps, tie_word_embeddings=False,
                     max_position_embeddings=4096)
    enc = Qwen2Encoder.__new__(Qwen2Encoder)
    torch.nn.Module.__init__(enc)
    enc.model = Qwen2ForCausalLM(qc)
    lm = CosyVoice3LM(cfg.hidden, cfg.hidden, cfg.speech_tokens, enc, partial(ras_sampling, **sampling),
                      head_num=cfg.head_num, inference_head_num=2, mtp_head_num=cfg.mtp_heads).eval()
    assert_spec(lm, W.llm_spec(cfg, with_lm_head=True), 'llm.pt')
    lm.load_state_dict(sd)
    return lm


def gen_llm():
    cfg = tiny_config().llm
    seed_w = 7
    sd = W.make_llm_state(cfg, seed=seed_w, init='fan_in', with_lm_head=True)
    out = dict(weight_seed=np.int64(seed_w), weight_sha=np.array(state_checksum(sd)))
    runs = [
        dict(K=1, seed=101, n_text=10, n_ptext=0, n_pspeech=0, sampling=dict(top_p=0.8, top_k=25, win_size=10, tau_r=0.1), maxr=4, minr=2),
        dict(K=2, seed=102, n_text=12, n_ptext=0, n_pspeech=0, sampling=dict(top_p=0.9, top_k=10, win_size=24, tau_r=0.2), maxr=4, minr=2),
        dict(K=3, seed=103, n_text=9, n_ptext=4, n_pspeech=7, sampling=dict(top_p=0.9, top_k=10, win_size=32, tau_r=0.2), maxr=5, minr=3),
        dict(K=5, seed=104, n_text=8, n_ptext=0, n_pspeech=5, sampling=dict(top_p=0.95, top_k=5, win_size=8, tau_r=0.1), maxr=6, minr=3),
        dict(K=0, seed=105, n_text=6, n_ptext=0, n_pspeech=0, sampling=dict(top_p=0.8, top_k=25, win_size=10, tau_r=0.1), maxr=3, minr=2),
    ]
    for r, run in enumerate(runs):
        lm = build_ref_llm(cfg, sd, run['sampling']) if r == 0 else lm
        lm.sampling = partial(lm.sampling.func, **run['sampling'])
        lm.inference_head_num = run['K']
        g = torch.Generator()
        g.manual_seed(run['seed'])
        text = torch.randint(0, cfg.text_vocab, (1, run['n_text']), dtype=torch.int32, generator=g)
        ptext = torch.randint(0, cfg.text_vocab, (1, run['n_ptext']), dtype=torch.int32, generator=g)
        pspeech = torch.randint(0, cfg.speech_tokens, (1, run['n_pspeech']), dtype=torch.int32, generator=g)
        torch.manual_seed(run['seed'])
        toks = list(lm.inference(
            text=text, text_len=torch.tensor([run['n_text']], dtype=torch.int32), prompt_text=ptext,
            prompt_text_len=torch.tensor([run['n_ptext']], dtype=torch.int32),
            prompt_speech_token=pspeech if run['n_pspeech'] else None,
            prompt_speech_token_len=torch.tensor([run['n_pspeech']], dtype=torch.int32),
            embedding=torch.zeros(0, 192), max_token_text_ratio=run['maxr'], min_token_text_ratio=run['minr']))
        # numeric pin of the first step: reference hidden + per-head log-probs on the initial prefix
        lm_input = llm_ref.build_prefix(sd, cfg, text[0], ptext[0], pspeech[0])[None]
        y, _ = lm.llm.forward_one_step(lm_input, masks=torch.tril(torch.ones(1, lm_input.shape[1], lm_input.shape[1])).bool(), cache=None)
        last = y[:, -1:, :]
        K = llm_ref.effective_heads(cfg, run['K'])
        logps = torch.stack([lm.llm_decoder(lm.mtp_block[j](last)[0][:, -1]).log_softmax(dim=-1)[0] for j in range(K)])
        # oracle must agree token-for-token, in both cache modes
        for use_cache in (False, True):
            ns = sampler_ref.NoiseStream(seed=run['seed'])
            otoks = list(llm_ref.llm_inference(sd, cfg, text[0], ns, prompt_text=ptext[0], prompt_speech_token=pspeech[0],
                                               inference_head_num=run['K'], sampling=run['sampling'],
                                               max_token_text_ratio=run['maxr'], min_token_text_ratio=run['minr'],
                                               use_kv_cache=use_cache))
            assert otoks == [int(t) for t in toks], (run, otoks, toks)
        oy = llm_ref.backbone(lm_input[0], sd, cfg)
        assert (oy - y[0]).abs().max() < 2e-5, (oy - y[0]).abs().max()
        print('[llm] run %d K=%d: %d tokens, oracle == reference (cached and uncached); hidden diff %.2e'
              % (r, run['K'], len(toks), (oy - y[0]).abs().max()))
        p = 'r%d_' % r
        out.update({p + 'K': np.int32(run['K']), p + 'seed': np.int64(run['seed']), p + 'text': text[0].numpy(),
                    p + 'ptext': ptext[0].numpy(), p + 'pspeech': pspeech[0].numpy(),
                    p + 'sampling': np.array([run['sampling']['top_p'], run['sampling']['top_k'], run['sampling']['win_size'], run['sampling']['tau_r']], dtype=np.float64),
                    p + 'ratios': np.array([run['maxr'], run['minr']], dtype=np.float64),
                    p + 'tokens': np.array(toks, dtype=np.int32), p + 'y_last': last[0, 0].numpy(), p + 'logps': logps.numpy()})
    out['n_runs'] = np.int32(len(runs))
    np.savez_compressed(os.path.join(HERE, 'llm_tiny.npz'), **out)


def gen_llm_stress():
    """BASELINE configs[2] "multi-head accept-rate stress": the reference LM on the accept-stress checkpoint (W.accept_stress_llm_state),
    K = 4 and K = 2, win_size 32 / tau_r 0.2; records the token streams and how often ras_sampling fell back to random_sampling."""
    import cosyvoice.utils.common as common
    cfg = tiny_config().llm
    seed_w = 7
    sd = W.accept_stress_llm_state(W.make_llm_state(cfg, seed=seed_w, init='fan_in', with_lm_head=True))
    out = dict(weight_seed=np.int64(seed_w), weight_sha=np.array(state_checksum(sd)))
    sampling = dict(top_p=0.9, top_k=10, win_size=32, tau_r=0.2)
    lm = build_ref_llm(cfg, sd, sampling)
    calls = dict(nucleus=0, random=0)
    orig_n, orig_r = common.nucleus_sampling, common.random_sampling

    def count_n(*a, **k):
        calls['nucleus'] += 1
        return orig_n(*a, **k)

    def count_r(*a, **k):
        calls['random'] += 1
        return orig_r(*a, **k)

    common.nucleus_sampling, common.random_sampling = count_n, count_r
    try:
        for r, (K, seed, n_text, n_ps) in enumerate([(4, 201, 12, 0), (2, 202, 10, 6), (4, 203, 9, 3)]):
            lm.inference_head_num = K
            calls.update(nucleus=0, random=0)
            g = torch.Generator()
            g.manual_seed(seed)
            text = torch.randint(0, cfg.text_vocab, (1, n_text), dtype=torch.int32, generator=g)
            pspeech = torch.randint(0, cfg.speech_tokens, (1, n_ps), dtype=torch.int32, generator=g)
            torch.manual_seed(seed)
            toks = list(lm.inference(text=text, text_len=torch.tensor([n_text], dtype=torch.int32), prompt_text=torch.zeros(1, 0, dtype=torch.int32),
                                     prompt_text_len=torch.tensor([0], dtype=torch.int32), prompt_speech_token=pspeech if n_ps else None,
                                     prompt_speech_token_len=torch.tensor([n_ps], dtype=torch.int32), embedding=torch.zeros(0, 192),
                                     max_token_text_ratio=8, min_token_text_ratio=8))
            for use_cache in (False, True):
                otoks = list(llm_ref.llm_inference(sd, cfg, text[0], sampler_ref.NoiseStream(seed=seed), prompt_speech_token=pspeech[0],
                                                   inference_head_num=K, sampling=sampling, max_token_text_ratio=8, min_token_text_ratio=8,
                                                   use_kv_cache=use_cache))
                assert otoks == [int(t) for t in toks], (r, otoks, toks)
            print('[llm-stress] run %d K=%d: %d tokens, %d sampler calls of which %d fell back to the full softmax (%.0f %%); oracle == reference'
                  % (r, K, len(toks), calls['nucleus'], calls['random'], 100.0 * calls['random'] / calls['nucleus']))
            p = 'r%d_' % r
            out.update({p + 'K': np.int32(K), p + 'seed': np.int64(seed), p + 'text': text[0].numpy(), p + 'pspeech': pspeech[0].numpy(),
                        p + 'tokens': np.array(toks, dtype=np.int32), p + 'calls': np.int32(calls['nucleus']), p + 'fallbacks': np.int32(calls['random'])})
    finally:
        common.nucleus_sampling, common.random_sampling = orig_n, orig_r
    out['sampling'] = np.array([sampling['top_p'], sampling['top_k'], sampling['win_size'], sampling['tau_r']], dtype=np.float64)
    out['n_runs'] = np.int32(3)
    np.savez_compressed(os.path.join(HERE, 'llm_stress_tiny.npz'), **out)


def gen_llm_cv3w(cfg=None, runs=None, pins=(1, 12), fname='llm_cv3w.npz', tag='llm-cv3w', literal_runs=(0,)):
    """The reference LM at the HydraVox-CV3 WIDTHS (hidden 896, 14:2 heads, inter 4864, vocab 6761, 5 MTP heads of 22016; 2 layers):
    first-step hidden / log-probs of all 5 heads, and token streams for K in {1, 2, 4, 5} — a set of 8 utterances at K = 2 (the batch
    the benchmark decodes) whose contexts start below and cross the 256- and 512-key attention splits."""
    import time
    cfg = cfg or cv3w_config().llm
    seed_w = 1986
    sd = W.make_llm_state(cfg, seed=seed_w, init='fan_in', with_lm_head=True)
    out = dict(weight_seed=np.int64(seed_w), weight_sha=np.array(state_checksum(sd)))
    sampling = dict(top_p=0.9, top_k=10, win_size=32, tau_r=0.2)
    lm = build_ref_llm(cfg, sd, sampling)
    # (K, n_text, n_prompt_text, n_prompt_speech, max ratio, min ratio)
    runs = runs or [(2, 14, 0, 0, 3, 2), (2, 20, 6, 150, 3, 2), (2, 24, 0, 215, 3, 2), (2, 30, 8, 200, 2, 2), (2, 16, 0, 460, 4, 3), (2, 9, 0, 30, 5, 2),
                    (2, 40, 10, 180, 2, 1), (2, 12, 0, 236, 4, 2),
                    (1, 16, 0, 230, 3, 2), (1, 10, 4, 0, 4, 2), (4, 24, 0, 210, 4, 3), (4, 11, 5, 40, 5, 3), (5, 20, 0, 225, 4, 3)]
    for r, (K, n_text, n_pt, n_ps, maxr, minr) in enumerate(runs):
        t0 = time.time()
        lm.inference_head_num = K
        seed = 700 + r
        g = torch.Generator()
        g.manual_seed(seed)
        text = torch.randint(0, cfg.text_vocab, (1, n_text), dtype=torch.int32, generator=g)
        ptext = torch.randint(0, cfg.text_vocab, (1, n_pt), dtype=torch.int32, generator=g)
        pspeech = torch.randint(0, cfg.speech_tokens, (1, n_ps), dtype=torch.int32, generator=g)
        torch.manual_seed(seed)
        toks = list(lm.inference(
            text=text, text_len=torch.tensor([n_text], dtype=torch.int32), prompt_text=ptext, prompt_text_len=torch.tensor([n_pt], dtype=torch.int32),
            prompt_speech_token=pspeech if n_ps else None, prompt_speech_token_len=torch.tensor([n_ps], dtype=torch.int32),
            embedding=torch.zeros(0, 192), max_token_text_ratio=maxr, min_token_text_ratio=minr))
        ns = sampler_ref.NoiseStream(seed=seed)
        otoks = list(llm_ref.llm_inference(sd, cfg, text[0], ns, prompt_text=ptext[0], prompt_speech_token=pspeech[0], inference_head_num=K,
                                           sampling=sampling, max_token_text_ratio=maxr, min_token_text_ratio=minr, use_kv_cache=True))
        assert otoks == [int(t) for t in toks], (r, otoks[:8], toks[:8])
        if r in literal_runs:                           # the literal (uncached) form once
            o2 = list(llm_ref.llm_inference(sd, cfg, text[0], sampler_ref.NoiseStream(seed=seed), prompt_text=ptext[0], prompt_speech_token=pspeech[0],
                                            inference_head_num=K, sampling=sampling, max_token_text_ratio=maxr, min_token_text_ratio=minr))
            assert o2 == otoks
        n_ctx0 = 2 + n_text + n_pt + n_ps
        print('[%s] run %d K=%d: prefix %d rows, %d tokens (context %d -> %d), oracle == reference; %.1f s'
              % (tag, r, K, n_ctx0, len(toks), n_ctx0, n_ctx0 + len(toks), time.time() - t0))
        p = 'r%d_' % r
        out.update({p + 'K': np.int32(K), p + 'seed': np.int64(seed), p + 'text': text[0].numpy(), p + 'ptext': ptext[0].numpy(),
                    p + 'pspeech': pspeech[0].numpy(), p + 'ratios': np.array([maxr, minr], dtype=np.float64), p + 'tokens': np.array(toks, dtype=np.int32)})
        if r in pins:                                   # numeric pin of the first step: all 5 heads
            lm_input = llm_ref.build_prefix(sd, cfg, text[0], ptext[0], pspeech[0])[None]
            L = lm_input.shape[1]
            y, _ = lm.llm.forward_one_step(lm_input, masks=torch.tril(torch.ones(1, L, L)).bool(), cache=None)
            last = y[:, -1:, :]
            logps = torch.stack([lm.llm_decoder(lm.mtp_block[j](last)[0][:, -1]).log_softmax(dim=-1)[0] for j in range(cfg.head_num)])
            oy = llm_ref.backbone(lm_input[0], sd, cfg)
            ol = torch.stack(llm_ref.head_logps(oy[-1], sd, cfg, cfg.head_num))
            d = [(oy - y[0]).abs().max().item(), (ol - logps).abs().max().item()]
            assert d[0] < 2e-4 and d[1] < 2e-3, d
            print('[%s] run %d first step: oracle-reference max abs diff hidden %.1e logp %.1e (|y| max %.2f)' % (tag, r, d[0], d[1], y.abs().max()))
            out.update({p + 'y_last': last[0, 0].numpy(), p + 'logps': logps.numpy()})
    out['sampling'] = np.array([sampling['top_p'], sampling['top_k'], sampling['win_size'], sampling['tau_r']], dtype=np.float64)
    out['n_runs'] = np.int32(len(runs))
    np.savez_compressed(os.path.join(HERE, fname), **out)


def gen_llm_cv3d():
    """The reference LM at FULL DEPTH (24 layers, CV3 widths; config.cv3d_config): K in {1, 2, 4}, <= 64 generated tokens per run, three K = 2
    runs for a batched decode, and one run whose 1020-row prefix grows across context 1024 (the reference recomputes the whole prefix per step:
    ~0.75 TFLOP per step there, so that run is short).  First-step hidden / log-probs of all 5 heads on a short and on the long prefix."""
    from flowmirror_hydravox_amd.config import cv3d_config
    runs = [(1, 10, 0, 0, 3, 2), (2, 16, 4, 120, 3, 2), (2, 12, 0, 40, 4, 3), (2, 20, 0, 200, 2, 2), (4, 12, 0, 60, 4, 3), (4, 16, 0, 250, 3, 2),
            (2, 8, 0, 1010, 3, 2)]
    gen_llm_cv3w(cfg=cv3d_config().llm, runs=runs, pins=(1, 6), fname='llm_cv3d.npz', tag='llm-cv3d', literal_runs=())


def gen_llm_bf16():
    """The reference LM run the way the reference deploys it — `llm.eval().cuda().to(torch.bfloat16)` (infer_speech_model.py:102), here on the CPU —
    next to its fp32 run: first-step hidden state and the log-probs of the heads on pin run 1 of llm_cv3d.npz (24 layers, 142-row prefix).  How far
    the reference's OWN production arithmetic sits from fp32; the product's bf16 mode is held to a multiple of that (tests/test_gpu_cv3d.py)."""
    import copy
    from flowmirror_hydravox_amd.config import cv3d_config
    cfg = cv3d_config().llm
    sd = W.make_llm_state(cfg, seed=1986, init='fan_in', with_lm_head=True)
    lm = build_ref_llm(cfg, sd, dict(top_p=0.9, top_k=10, win_size=32, tau_r=0.2))
    g0 = np.load(os.path.join(HERE, 'llm_cv3d.npz'))
    assert str(g0['weight_sha']) == state_checksum(sd)
    p = 'r1_'
    text, ptext, pspeech = (torch.from_numpy(g0[p + k]) for k in ('text', 'ptext', 'pspeech'))
    lm_input = llm_ref.build_prefix(sd, cfg, text, ptext, pspeech)[None]
    L = lm_input.shape[1]
    masks = torch.tril(torch.ones(1, L, L)).bool()
    out = {}
    with torch.inference_mode():
        y, _ = lm.llm.forward_one_step(lm_input, masks=masks, cache=None)
        last = y[:, -1:, :]
        logps = torch.stack([lm.llm_decoder(lm.mtp_block[j](last)[0][:, -1]).log_softmax(dim=-1)[0] for j in range(cfg.head_num)])
        assert (last[0, 0] - torch.from_numpy(g0[p + 'y_last'])).abs().max().item() < 1e-5
        lb = copy.deepcopy(lm).to(torch.bfloat16)
        yb, _ = lb.llm.forward_one_step(lm_input.to(torch.bfloat16), masks=masks, cache=None)
        lastb = yb[:, -1:, :]
        logpsb = torch.stack([lb.llm_decoder(lb.mtp_block[j](lastb)[0][:, -1]).float().log_softmax(dim=-1)[0] for j in range(cfg.head_num)])
    dy = (lastb.float() - last).abs().max().item() / last.abs().max().item()
    dl = (logpsb - logps).abs().max().item()
    print('[llm-bf16] 24 layers, %d-row prefix: reference bf16 vs reference fp32: hidden %.2e of its scale, log-probs %.2e (max abs)' % (L, dy, dl))
    out.update(prefix_rows=np.int32(L), y_last_f32=last[0, 0].numpy(), y_last_bf16=lastb[0, 0].float().numpy(), logps_f32=logps.numpy(), logps_bf16=logpsb.numpy(),
               hidden_bf16_vs_f32=np.float64(dy), logp_bf16_vs_f32=np.float64(dl), weight_sha=np.array(state_checksum(sd)))
    np.savez_compressed(os.path.join(HERE, 'llm_bf16.npz'), **out)


# ------------------------------------------------------------------------------------------------
# flow
# ------------------------------------------------------------------------------------------------
def gen_flow():
    from cosyvoice.flow.flow import CausalMaskedDiffWithDiT
    from cosyvoice.flow.flow_matching import CausalConditionalCFM
    from cosyvoice.flow.DiT.dit import DiT
    from cosyvoice.transformer.upsample_encoder import PreLookaheadLayer
    c = tiny_config().flow
    dit = DiT(dim=c.dim, depth=c.depth, heads=c.heads, dim_head=c.head_dim, ff_mult=c.ff_mult, mel_dim=c.mel, mu_dim=c.mel,
              spk_dim=c.mel, out_channels=c.mel, static_chunk_size=50)
    cfm = CausalConditionalCFM(in_channels=240, cfm_params=DictConfig(sigma_min=1e-6, solver='euler', t_scheduler='cosine',
                                                                     training_cfg_rate=0.2, inference_cfg_rate=c.cfg_rate, reg_loss_type='l1'),
                               n_spks=1, spk_emb_dim=80, estimator=dit)
    pla = PreLookaheadLayer(in_channels=80, channels=c.pre_lookahead_channels, pre_lookahead_len=c.pre_lookahead_len)
    flow = CausalMaskedDiffWithDiT(input_size=80, output_size=80, spk_embed_dim=192, vocab_size=c.vocab, token_mel_ratio=2,
                                   pre_lookahead_len=3, pre_lookahead_layer=pla, decoder=cfm).eval()
    assert_spec(flow, W.flow_spec(c), 'flow.pt')
    seed_w = 11
    sd = W.make_flow_state(c, seed=seed_w, init='fan_in')
    flow.load_state_dict(sd)
    assert torch.equal(cfm.rand_noise, flow_ref.cfm_noise(c)), 'fixed CFM noise differs'
    out = dict(weight_seed=np.int64(seed_w), weight_sha=np.array(state_checksum(sd)), noise_head=cfm.rand_noise[0, :2, :8].numpy())
    g = torch.Generator()
    g.manual_seed(5)
    for r, (N, Np) in enumerate([(20, 6), (33, 0)]):
        token = torch.randint(0, c.vocab, (1, N), generator=g)
        ptoken = torch.randint(0, c.vocab, (1, Np), generator=g)
        pfeat = torch.randn(1, 2 * Np, 80, generator=g)
        emb = torch.randn(1, 192, generator=g)
        e = flow.spk_embed_affine_layer(F.normalize(emb, dim=1))
        tk = torch.cat([ptoken, token], dim=1)
        h0 = flow.input_embedding(tk)
        hp = flow.pre_lookahead_layer(h0)
        h = hp.repeat_interleave(2, dim=1)
        T = h.shape[1]
        cond = torch.zeros(1, T, 80)
        cond[:, :2 * Np] = pfeat
        feat, _ = flow.decoder(mu=h.transpose(1, 2).contiguous(), mask=torch.ones(1, 1, T), spks=e, cond=cond.transpose(1, 2),
                               n_timesteps=10, streaming=False)
        feat = feat[:, :, 2 * Np:]
        x_in = torch.randn(2, 80, T, generator=g)
        mu_in = torch.randn(2, 80, T, generator=g)
        t_in = torch.tensor([0.3, 0.3])
        sp_in = torch.randn(2, 80, generator=g)
        c_in = torch.randn(2, 80, T, generator=g)
        est = dit(x_in, torch.ones(2, 1, T), mu_in, t_in, sp_in, c_in)
        o_feat = flow_ref.flow_inference(token, emb, sd, c, prompt_token=ptoken if Np else None, prompt_feat=pfeat if Np else None)
        o_est = flow_ref.dit_forward(x_in, torch.ones(2, 1, T), mu_in, t_in, sp_in, c_in, sd, c)
        o_pla = flow_ref.pre_lookahead(h0, sd, c)
        d = [(o_pla - hp).abs().max().item(), (o_est - est).abs().max().item(), (o_feat - feat).abs().max().item()]
        assert max(d) < 1e-4, d
        print('[flow] run %d N=%d Np=%d: oracle-reference max abs diff pla %.1e est %.1e mel %.1e' % (r, N, Np, *d))
        p = 'r%d_' % r
        out.update({p + 'token': token.numpy(), p + 'ptoken': ptoken.numpy(), p + 'pfeat': pfeat.numpy(), p + 'emb': emb.numpy(),
                    p + 'h0': h0.numpy(), p + 'pla': hp.numpy(), p + 'mel': feat.numpy(),
                    p + 'est_x': x_in.numpy(), p + 'est_mu': mu_in.numpy(), p + 'est_t': t_in.numpy(), p + 'est_spk': sp_in.numpy(),
                    p + 'est_cond': c_in.numpy(), p + 'est_out': est.numpy()})
    out['n_runs'] = np.int32(2)
    np.savez_compressed(os.path.join(HERE, 'flow_tiny.npz'), **out)


def cv3w_flow_inputs(seed, T, lens):
    """seeded estimator inputs (regenerated by the tests from the seed; the fixture stores their checksum and the reference output)"""
    g = torch.Generator()
    g.manual_seed(seed)
    x, mu, cond = (torch.randn(2, 80, T, generator=g) for _ in range(3))
    spk = torch.randn(2, 80, generator=g)
    mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).float()[:, None, :]
    return x, mask, mu, spk, cond


def gen_flow_cv3w(c=None, cases=None, N=130, Np=45, fname='flow_cv3w.npz', tag_='flow-cv3w'):
    """The reference flow decoder at the HydraVox-CV3 WIDTHS (DiT 1024 x 16 heads x ff 2048, conv groups 16, pre-lookahead 1024; 2 blocks):
    estimator at T = 2176 (padded second row), estimator with the chunk mask, pre-lookahead and a whole flow.inference with a prompt."""
    from cosyvoice.flow.flow import CausalMaskedDiffWithDiT
    from cosyvoice.flow.flow_matching import CausalConditionalCFM
    from cosyvoice.flow.DiT.dit import DiT
    from cosyvoice.transformer.upsample_encoder import PreLookaheadLayer
    c = c or cv3w_config().flow
    dit = DiT(dim=c.dim, depth=c.depth, heads=c.heads, dim_head=c.head_dim, ff_mult=c.ff_mult, mel_dim=c.mel, mu_dim=c.mel,
              spk_dim=c.mel, out_channels=c.mel, static_chunk_size=c.static_chunk_size)
    cfm = CausalConditionalCFM(in_channels=240, cfm_params=DictConfig(sigma_min=1e-6, solver='euler', t_scheduler='cosine',
                                                                     training_cfg_rate=0.2, inference_cfg_rate=c.cfg_rate, reg_loss_type='l1'),
                               n_spks=1, spk_emb_dim=80, estimator=dit)
    pla = PreLookaheadLayer(in_channels=80, channels=c.pre_lookahead_channels, pre_lookahead_len=c.pre_lookahead_len)
    flow = CausalMaskedDiffWithDiT(input_size=80, output_size=80, spk_embed_dim=192, vocab_size=c.vocab, token_mel_ratio=2,
                                   pre_lookahead_len=3, pre_lookahead_layer=pla, decoder=cfm).eval()
    assert_spec(flow, W.flow_spec(c), 'flow.pt (cv3 widths)')
    seed_w = 1987
    sd = W.make_flow_state(c, seed=seed_w, init='fan_in')
    flow.load_state_dict(sd)
    out = dict(weight_seed=np.int64(seed_w), weight_sha=np.array(state_checksum(sd)))
    # ---- estimator, long padded batch ----------------------------------------------------------------------------------------
    for tag, seed, T, lens, streaming in (cases or (('e0', 41, 2176, [2176, 1900], False), ('e1', 42, 330, [330, 275], True))):
        x, mask, mu, spk, cond = cv3w_flow_inputs(seed, T, lens)
        t = torch.tensor([0.3, 0.3]) if tag == 'e0' else torch.tensor([0.7, 0.15])
        est = dit(x, mask, mu, t, spk, cond, streaming=streaming)
        o_est = flow_ref.dit_forward(x, mask, mu, t, spk, cond, sd, c, streaming=streaming)
        d = ((o_est - est) * mask).abs().max().item()
        assert d < 1e-3, d
        print('[' + tag_ + '] estimator %s T=%d lens=%s streaming=%s: oracle-reference max abs diff %.1e (out absmax %.2f)' % (tag, T, lens, streaming, d, (est * mask).abs().max()))
        out.update({tag + '_seed': np.int64(seed), tag + '_T': np.int32(T), tag + '_lens': np.array(lens, dtype=np.int32), tag + '_t': t.numpy(),
                    tag + '_streaming': np.int32(streaming), tag + '_in_sha': np.array(state_checksum(dict(x=x, mu=mu, spk=spk, cond=cond))),
                    tag + '_out': (est * mask).numpy()})
    # ---- pre-lookahead + whole inference with a prompt -------------------------------------------------------------------------
    g = torch.Generator()
    g.manual_seed(43)
    token = torch.randint(0, c.vocab, (1, N), generator=g)
    ptoken = torch.randint(0, c.vocab, (1, Np), generator=g)
    pfeat = torch.randn(1, 2 * Np, 80, generator=g)
    emb = torch.randn(1, 192, generator=g)
    e = flow.spk_embed_affine_layer(F.normalize(emb, dim=1))
    h0 = flow.input_embedding(torch.cat([ptoken, token], dim=1))
    hp = flow.pre_lookahead_layer(h0)
    h = hp.repeat_interleave(2, dim=1)
    T = h.shape[1]
    cond = torch.zeros(1, T, 80)
    cond[:, :2 * Np] = pfeat
    feat, _ = flow.decoder(mu=h.transpose(1, 2).contiguous(), mask=torch.ones(1, 1, T), spks=e, cond=cond.transpose(1, 2), n_timesteps=10, streaming=False)
    feat = feat[:, :, 2 * Np:]
    o_pla = flow_ref.pre_lookahead(h0, sd, c)
    o_feat = flow_ref.flow_inference(token, emb, sd, c, prompt_token=ptoken, prompt_feat=pfeat)
    d = [(o_pla - hp).abs().max().item(), (o_feat - feat).abs().max().item()]
    assert max(d) < 2e-3, d
    print('[' + tag_ + '] inference N=%d prompt=%d: oracle-reference max abs diff pla %.1e mel %.1e (mel absmax %.2f)' % (N, Np, d[0], d[1], feat.abs().max()))
    out.update(token=token.numpy(), ptoken=ptoken.numpy(), pfeat=pfeat.numpy(), emb=emb.numpy(), h0=h0.numpy(), pla=hp.numpy(), mel=feat.numpy())
    np.savez_compressed(os.path.join(HERE, fname), **out)


def gen_flow_cv3d():
    """The reference flow decoder at FULL DEPTH (22 DiT blocks, config.cv3d_config): estimator at T = 256 with a padded second row (200 frames)
    and at T = 192 with the chunk mask, and a whole 10-step CFG solve of 104 + 24 prompt tokens (T = 256)."""
    from flowmirror_hydravox_amd.config import cv3d_config
    gen_flow_cv3w(c=cv3d_config().flow, cases=(('e0', 51, 256, [256, 200], False), ('e1', 52, 192, [192, 150], True)), N=104, Np=24,
                  fname='flow_cv3d.npz', tag_='flow-cv3d')


def gen_flow_half():
    """The reference DiT run the way the reference deploys it — `flow.eval().cuda().half()` (infer_speech_model.py:103), here on the CPU — next to its
    fp32 run, at the CV3 widths (2 blocks, T = 330, chunk mask) and at full depth (22 blocks, T = 192): how far the reference's OWN production
    arithmetic sits from fp32.  The product's bf16 mode is held to a multiple of that distance (tests/test_gpu_cv3d.py)."""
    import copy
    from cosyvoice.flow.DiT.dit import DiT
    from flowmirror_hydravox_amd.config import cv3d_config
    out = {}
    for name, c, seed, T, lens, t in (('w', cv3w_config().flow, 42, 330, [330, 275], [0.7, 0.15]), ('d', cv3d_config().flow, 52, 192, [192, 150], [0.7, 0.15])):
        dit = DiT(dim=c.dim, depth=c.depth, heads=c.heads, dim_head=c.head_dim, ff_mult=c.ff_mult, mel_dim=c.mel, mu_dim=c.mel,
                  spk_dim=c.mel, out_channels=c.mel, static_chunk_size=c.static_chunk_size).eval()
        sd = W.make_flow_state(c, seed=1987, init='fan_in')
        dit.load_state_dict({k[len('decoder.estimator.'):]: v for k, v in sd.items() if k.startswith('decoder.estimator.')})
        x, mask, mu, spk, cond = cv3w_flow_inputs(seed, T, lens)
        tt = torch.tensor(t)
        with torch.inference_mode():
            est = dit(x, mask, mu, tt, spk, cond, streaming=True) * mask
            dh = copy.deepcopy(dit).half()
            est_h = (dh(x.half(), mask, mu.half(), tt.half(), spk.half(), cond.half(), streaming=True).float() * mask)
        scale = est.abs().max().item()
        d = (est_h - est).abs().max().item() / scale
        print('[flow-half] %s (%d blocks, T = %d): reference fp16 vs reference fp32: %.2e of the output scale (absmax %.2f)' % (name, c.depth, T, d, scale))
        out.update({name + '_seed': np.int64(seed), name + '_T': np.int32(T), name + '_lens': np.array(lens, dtype=np.int32), name + '_t': tt.numpy(),
                    name + '_out_f32': est.numpy(), name + '_out_f16': est_h.numpy(), name + '_half_vs_f32': np.float64(d),
                    name + '_in_sha': np.array(state_checksum(dict(x=x, mu=mu, spk=spk, cond=cond))), name + '_weight_sha': np.array(state_checksum(sd))})
    np.savez_compressed(os.path.join(HERE, 'flow_half.npz'), **out)


def gen_hift_cv3w():
    """The reference HiFT vocoder at full width (base 512, F0 predictor 512): stage outputs for T in {8, 50, 160}."""
    from cosyvoice.hifigan.generator import CausalHiFTGenerator
    from cosyvoice.hifigan.f0_predictor import CausalConvRNNF0Predictor
    c = cv3w_config().hift
    f0p = CausalConvRNNF0Predictor(num_class=1, in_channels=80, cond_channels=c.f0_channels)
    gen = CausalHiFTGenerator(
        in_channels=80, base_channels=c.base_channels, nb_harmonics=c.nb_harmonics, sampling_rate=c.sampling_rate,
        nsf_alpha=c.nsf_alpha, nsf_sigma=c.nsf_sigma, nsf_voiced_threshold=c.nsf_voiced_threshold,
        upsample_rates=c.upsample_rates, upsample_kernel_sizes=c.upsample_kernel_sizes,
        istft_params={'n_fft': c.n_fft, 'hop_len': c.hop}, resblock_kernel_sizes=c.resblock_kernel_sizes,
        resblock_dilation_sizes=c.resblock_dilations, source_resblock_kernel_sizes=c.source_resblock_kernel_sizes,
        source_resblock_dilation_sizes=c.source_resblock_dilations, lrelu_slope=c.lrelu_slope, audio_limit=c.audio_limit,
        conv_pre_look_right=c.conv_pre_look_right, f0_predictor=f0p).eval()
    assert_spec(gen, W.hift_spec(c), 'hift.pt (full width)')
    seed_w, seed_t = 1988, 9
    sd = W.make_hift_state(c, seed=seed_w, init='fan_in')
    gen.load_state_dict(sd)
    tables = hift_ref.make_tables(c, seed=seed_t)
    gen.m_source.l_sin_gen.rand_ini = tables['rand_ini']
    gen.m_source.l_sin_gen.sine_waves = tables['sine_waves']
    gen.m_source.uv = tables['uv']
    out = dict(weight_seed=np.int64(seed_w), table_seed=np.int64(seed_t), weight_sha=np.array(state_checksum(sd)))
    g = torch.Generator()
    g.manual_seed(4)
    for r, T in enumerate([8, 50, 160]):
        mel = torch.randn(1, 80, T, generator=g)
        with torch.inference_mode():
            f0 = gen.f0_predictor(mel)
            wav, s = gen.inference(speech_feat=mel)
        o_f0 = hift_ref.f0_predictor(mel, sd)
        o_wav_s = hift_ref.decode(mel, s, sd, c)
        d = [(o_f0 - f0).abs().max().item(), (o_wav_s - wav).abs().max().item()]
        assert d[0] < 2e-3 and d[1] < 5e-4, d
        print('[hift-cv3w] T=%d: oracle-reference max abs diff f0 %.1e decode(ref source) %.1e (wav std %.3f, f0 max %.1f, voiced %.0f %%)'
              % (T, d[0], d[1], wav.std(), f0.max(), 100.0 * (f0 > 10).float().mean()))
        p = 'r%d_' % r
        out.update({p + 'mel': mel.numpy(), p + 'f0': f0.numpy(), p + 'source': s.numpy(), p + 'wav': wav.numpy()})
    out['n_runs'] = np.int32(3)
    np.savez_compressed(os.path.join(HERE, 'hift_cv3w.npz'), **out)


# ------------------------------------------------------------------------------------------------
# hift
# ------------------------------------------------------------------------------------------------
def gen_hift():
    from cosyvoice.hifigan.generator import CausalHiFTGenerator
    from cosyvoice.hifigan.f0_predictor import CausalConvRNNF0Predictor
    c = tiny_config().hift
    f0p = CausalConvRNNF0Predictor(num_class=1, in_channels=80, cond_channels=c.f0_channels)
    gen = CausalHiFTGenerator(
        in_channels=80, base_channels=c.base_channels, nb_harmonics=c.nb_harmonics, sampling_rate=c.sampling_rate,
        nsf_alpha=c.nsf_alpha, nsf_sigma=c.nsf_sigma, nsf_voiced_threshold=c.nsf_voiced_threshold,
        upsample_rates=c.upsample_rates, upsample_kernel_sizes=c.upsample_kernel_sizes,
        istft_params={'n_fft': c.n_fft, 'hop_len': c.hop}, resblock_kernel_sizes=c.resblock_kernel_sizes,
        resblock_dilation_sizes=c.resblock_dilations, source_resblock_kernel_sizes=c.source_resblock_kernel_sizes,
        source_resblock_dilation_sizes=c.source_resblock_dilations, lrelu_slope=c.lrelu_slope, audio_limit=c.audio_limit,
        conv_pre_look_right=c.conv_pre_look_right, f0_predictor=f0p).eval()
    assert_spec(gen, W.hift_spec(c), 'hift.pt')
    seed_w, seed_t = 3, 9
    sd = W.make_hift_state(c, seed=seed_w, init='fan_in')
    gen.load_state_dict(sd)
    tables = hift_ref.make_tables(c, seed=seed_t)
    gen.m_source.l_sin_gen.rand_ini = tables['rand_ini']
    gen.m_source.l_sin_gen.sine_waves = tables['sine_waves']
    gen.m_source.uv = tables['uv']
    out = dict(weight_seed=np.int64(seed_w), table_seed=np.int64(seed_t), weight_sha=np.array(state_checksum(sd)))
    g = torch.Generator()
    g.manual_seed(2)
    for r, T in enumerate([8, 24, 50]):
        mel = torch.randn(1, 80, T, generator=g)
        with torch.inference_mode():
            f0 = gen.f0_predictor(mel)
            wav, s = gen.inference(speech_feat=mel)
        o_f0 = hift_ref.f0_predictor(mel, sd)
        o_wav_s = hift_ref.decode(mel, s, sd, c)                       # decode on the reference's own source
        o_wav, o_s = hift_ref.hift_inference(mel, sd, c, tables)
        d = [(o_f0 - f0).abs().max().item(), (o_s - s).abs().max().item(), (o_wav_s - wav).abs().max().item(), (o_wav - wav).abs().max().item()]
        assert d[0] < 1e-3 and d[1] < 1e-3 and d[2] < 2e-4, d
        print('[hift] T=%d: oracle-reference max abs diff f0 %.1e source %.1e decode(ref source) %.1e end-to-end %.1e (wav std %.3f)'
              % (T, *d, wav.std()))
        p = 'r%d_' % r
        out.update({p + 'mel': mel.numpy(), p + 'f0': f0.numpy(), p + 'source': s.numpy(), p + 'wav': wav.numpy()})
    out['n_runs'] = np.int32(3)
    np.savez_compressed(os.path.join(HERE, 'hift_tiny.npz'), **out)


# ------------------------------------------------------------------------------------------------
# Matcha-TTS family
# ------------------------------------------------------------------------------------------------
def gen_matcha():
    from matcha.models.components.decoder import Decoder
    from matcha.models.components.flow_matching import CFM
    from cosyvoice.flow.decoder import ConditionalDecoder
    from matcha.hifigan.models import Generator
    from matcha.hifigan.denoiser import Denoiser
    from omegaconf import DictConfig
    from flowmirror_hydravox_amd.config import tiny_matcha_config, tiny_hifigan_config
    from oracle import matcha_ref
    out = {}
    g = torch.Generator()
    g.manual_seed(11)
    # ---- M1-M3: Matcha Decoder inside CFM (n_spks > 1: the speaker vector is packed after mu) -----------------------------
    c = tiny_matcha_config()
    dec_params = dict(channels=c.channels, dropout=0.0, attention_head_dim=c.head_dim, n_blocks=c.n_blocks, num_mid_blocks=c.num_mid_blocks,
                      num_heads=c.num_heads, act_fn='snakebeta')
    cfm = CFM(in_channels=2 * c.mel, out_channel=c.mel, cfm_params=DictConfig(solver='euler', sigma_min=1e-4), decoder_params=dec_params,
              n_spks=2, spk_emb_dim=c.spk_dim).eval()
    assert isinstance(cfm.estimator, Decoder)
    assert_spec(cfm.estimator, W.matcha_spec(c), 'matcha decoder')
    sd = W.make_matcha_state(c, seed=21, init='fan_in')
    cfm.estimator.load_state_dict(sd)
    out['m_weight_seed'] = np.int64(21)
    out['m_weight_sha'] = np.array(state_checksum(sd))
    for r, T in enumerate([16, 40]):                      # multiples of 2^(stages-1): Matcha pads its inputs to that (fix_len_compatibility)
        x, mu, spks = torch.randn(1, c.mel, T, generator=g), torch.randn(1, c.mel, T, generator=g), torch.randn(1, c.spk_dim, generator=g)
        mask = torch.ones(1, 1, T)
        t = torch.tensor([0.3 + 0.4 * r])
        with torch.inference_mode():
            y = cfm.estimator(x, mask, mu, t, spks)
            torch.manual_seed(100 + r)
            noise = torch.randn_like(mu)
            torch.manual_seed(100 + r)
            sample = cfm(mu, mask, c.n_timesteps, temperature=c.temperature, spks=spks)
        o_y = matcha_ref.decoder_forward(sd, c, x, mask, mu, t, spks)
        o_s = matcha_ref.solve_euler(sd, c, noise * c.temperature, mask, mu, c.n_timesteps, spks)
        d = [(o_y - y).abs().max().item(), (o_s - sample).abs().max().item()]
        assert d[0] < 1e-4 and d[1] < 1e-3, d
        print('[matcha] T=%d: oracle-reference max abs diff estimator %.1e, %d-step sample %.1e (out std %.3f)' % (T, d[0], c.n_timesteps, d[1], y.std()))
        p = 'm%d_' % r
        out.update({p + 'x': x.numpy(), p + 'mu': mu.numpy(), p + 'spks': spks.numpy(), p + 't': t.numpy(), p + 'y': y.numpy(),
                    p + 'noise': noise.numpy(), p + 'sample': sample.numpy()})
    # ---- M2 (CosyVoice variant): padded batch, cond input, odd length (skip trimming) --------------------------------------------
    cc = tiny_matcha_config(cv=True)
    dec = ConditionalDecoder(in_channels=cc.in_channels, out_channels=cc.mel, channels=cc.channels, dropout=0.0, attention_head_dim=cc.head_dim,
                             n_blocks=cc.n_blocks, num_mid_blocks=cc.num_mid_blocks, num_heads=cc.num_heads, act_fn='snakebeta').eval()
    assert_spec(dec, W.matcha_spec(cc), 'conditional decoder')
    sdc = W.make_matcha_state(cc, seed=22, init='fan_in')
    dec.load_state_dict(sdc)
    out['c_weight_seed'] = np.int64(22)
    out['c_weight_sha'] = np.array(state_checksum(sdc))
    T = 37
    lens = [37, 29]
    x, mu, cond = torch.randn(2, cc.mel, T, generator=g), torch.randn(2, cc.mel, T, generator=g), torch.randn(2, cc.mel, T, generator=g)
    spks = torch.randn(2, cc.spk_dim, generator=g)
    mask = torch.zeros(2, 1, T)
    for b, n in enumerate(lens):
        mask[b, :, :n] = 1.0
    t = torch.tensor([0.25, 0.75])
    with torch.inference_mode():
        y = dec(x, mask, mu, t, spks, cond)
    o_y = matcha_ref.decoder_forward(sdc, cc, x, mask, mu, t, spks, cond)
    d = (o_y - y).abs().max().item()
    assert d < 1e-4, d
    print('[matcha] conditional decoder (B=2, lens %s): oracle-reference max abs diff %.1e' % (lens, d))
    out.update({'c_x': x.numpy(), 'c_mu': mu.numpy(), 'c_cond': cond.numpy(), 'c_spks': spks.numpy(), 'c_mask': mask.numpy(), 'c_t': t.numpy(), 'c_y': y.numpy()})
    # ---- M4 / M5: HiFi-GAN v1 generator + denoiser ---------------------------------------------------------------------------------
    hc = tiny_hifigan_config()
    h = DictConfig(resblock='1', upsample_rates=list(hc.upsample_rates), upsample_kernel_sizes=list(hc.upsample_kernel_sizes),
                   upsample_initial_channel=hc.initial_channel, resblock_kernel_sizes=list(hc.resblock_kernel_sizes),
                   resblock_dilation_sizes=[list(d) for d in hc.resblock_dilations])
    gen = Generator(h).eval()
    assert_spec(gen, W.hifigan_spec(hc), 'hifigan generator')
    sdg = W.make_hifigan_state(hc, seed=23, init='fan_in')
    gen.load_state_dict(sdg)
    out['g_weight_seed'] = np.int64(23)
    out['g_weight_sha'] = np.array(state_checksum(sdg))
    mel = torch.randn(1, hc.mel, 30, generator=g)
    with torch.inference_mode():
        wav = gen(mel)
        den = Denoiser(gen, filter_length=hc.n_fft, n_overlap=hc.n_overlap, win_length=hc.n_fft, mode='zeros')
        clean = den(wav.squeeze(1), strength=0.05)
    o_wav = matcha_ref.generator_forward(sdg, hc, mel)
    o_bias = matcha_ref.denoiser_bias(sdg, hc)
    o_clean = matcha_ref.denoise(wav.squeeze(1), o_bias, hc, 0.05)
    d = [(o_wav - wav).abs().max().item(), (o_bias - den.bias_spec).abs().max().item(), (o_clean - clean).abs().max().item()]
    assert max(d) < 1e-4, d
    print('[matcha] hifigan: oracle-reference max abs diff wav %.1e bias %.1e denoised %.1e (wav std %.3f, bias max %.3f)'
          % (d[0], d[1], d[2], wav.std(), den.bias_spec.max()))
    out.update({'g_mel': mel.numpy(), 'g_wav': wav.numpy(), 'g_bias': den.bias_spec.numpy(), 'g_clean': clean.numpy(), 'g_strength': np.float32(0.05)})
    # ---- N2: prompt log-mel (matcha/utils/audio.py: mel_spectrogram) at the CosyVoice3 settings and at toy settings ------------------
    import importlib.util
    spec_ = importlib.util.spec_from_file_location('ref_matcha_audio', '/root/reference/matcha/utils/audio.py')
    audio = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(audio)
    from flowmirror_hydravox_amd.packing import mel_filterbank
    for tag, kw, L in (('cv3', dict(n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=8000), 24000),
                       ('toy', dict(n_fft=64, num_mels=8, sampling_rate=800, hop_size=32, win_size=64, fmin=0, fmax=400), 1000)):
        y = (torch.rand(2, L, generator=g) * 1.6 - 0.8)
        audio.mel_basis.clear()
        audio.hann_window.clear()
        ref = audio.mel_spectrogram(y, center=False, **kw)
        mb = mel_filterbank(kw['sampling_rate'], kw['n_fft'], kw['num_mels'], kw['fmin'], kw['fmax'])
        ora = matcha_ref.mel_spectrogram(y, mel_basis=mb, **kw)
        d = (ora - ref).abs().max().item()
        assert d < 1e-4, d
        print('[mel] %s: oracle-reference max abs diff %.1e, shape %s (filterbank = restated librosa table)' % (tag, d, tuple(ref.shape)))
        out.update({'mel_%s_y' % tag: y.numpy(), 'mel_%s_out' % tag: ref.numpy()})
    np.savez_compressed(os.path.join(HERE, 'matcha_tiny.npz'), **out)


# ------------------------------------------------------------------------------------------------
# streaming synthesis: static chunk mask in the DiT, finalize=False in flow and HiFT
# ------------------------------------------------------------------------------------------------
def gen_stream():
    import dataclasses
    from cosyvoice.flow.flow import CausalMaskedDiffWithDiT
    from cosyvoice.flow.flow_matching import CausalConditionalCFM
    from cosyvoice.flow.DiT.dit import DiT
    from cosyvoice.transformer.upsample_encoder import PreLookaheadLayer
    from cosyvoice.hifigan.generator import CausalHiFTGenerator
    from cosyvoice.hifigan.f0_predictor import CausalConvRNNF0Predictor
    CHUNK = 16
    c = dataclasses.replace(tiny_config().flow, static_chunk_size=CHUNK)
    dit = DiT(dim=c.dim, depth=c.depth, heads=c.heads, dim_head=c.head_dim, ff_mult=c.ff_mult, mel_dim=c.mel, mu_dim=c.mel,
              spk_dim=c.mel, out_channels=c.mel, static_chunk_size=CHUNK)
    cfm = CausalConditionalCFM(in_channels=240, cfm_params=DictConfig(sigma_min=1e-6, solver='euler', t_scheduler='cosine',
                                                                     training_cfg_rate=0.2, inference_cfg_rate=c.cfg_rate, reg_loss_type='l1'),
                               n_spks=1, spk_emb_dim=80, estimator=dit)
    pla = PreLookaheadLayer(in_channels=80, channels=c.pre_lookahead_channels, pre_lookahead_len=c.pre_lookahead_len)
    flow = CausalMaskedDiffWithDiT(input_size=80, output_size=80, spk_embed_dim=192, vocab_size=c.vocab, token_mel_ratio=2,
                                   pre_lookahead_len=3, pre_lookahead_layer=pla, decoder=cfm).eval()
    seed_w = 11
    sd = W.make_flow_state(c, seed=seed_w, init='fan_in')
    flow.load_state_dict(sd)
    out = dict(chunk=np.int32(CHUNK), flow_weight_seed=np.int64(seed_w), flow_weight_sha=np.array(state_checksum(sd)))
    g = torch.Generator()
    g.manual_seed(17)
    # ---- estimator with the chunk mask, padded batch -----------------------------------------------------------------------
    T, lens = 70, [70, 45]
    x_in = torch.randn(2, 80, T, generator=g)
    mu_in = torch.randn(2, 80, T, generator=g)
    t_in = torch.tensor([0.3, 0.7])
    sp_in = torch.randn(2, 80, generator=g)
    c_in = torch.randn(2, 80, T, generator=g)
    mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).float()[:, None, :]
    est = dit(x_in, mask, mu_in, t_in, sp_in, c_in, streaming=True)
    o_est = flow_ref.dit_forward(x_in, mask, mu_in, t_in, sp_in, c_in, sd, c, streaming=True)
    d = ((o_est - est) * mask).abs().max().item()
    assert d < 1e-4, d
    full = dit(x_in, mask, mu_in, t_in, sp_in, c_in, streaming=False)
    print('[stream] estimator chunk=%d lens=%s: oracle-reference max abs diff %.1e (streaming vs full mask differ by %.2f)'
          % (CHUNK, lens, d, ((full - est) * mask).abs().max().item()))
    out.update(est_x=x_in.numpy(), est_mu=mu_in.numpy(), est_t=t_in.numpy(), est_spk=sp_in.numpy(), est_cond=c_in.numpy(),
               est_mask=mask.numpy(), est_out=est.numpy())

    # ---- flow.inference(streaming=True) whole and in chunks (the check of the reference's own __main__, flow.py:436-459) ----
    def ref_flow(token, ptoken, pfeat, emb, finalize):
        # flow.py:388-430 in fp32 (the shipped method casts to bf16/fp16 and cannot run on CPU; SURVEY.md finding 7)
        e = flow.spk_embed_affine_layer(F.normalize(emb, dim=1))
        h0 = flow.input_embedding(torch.cat([ptoken, token], dim=1))
        h = flow.pre_lookahead_layer(h0) if finalize else flow.pre_lookahead_layer(h0[:, :-3], context=h0[:, -3:])
        h = h.repeat_interleave(2, dim=1)
        Tm = h.shape[1]
        cond = torch.zeros(1, Tm, 80)
        cond[:, :pfeat.shape[1]] = pfeat
        feat, _ = flow.decoder(mu=h.transpose(1, 2).contiguous(), mask=torch.ones(1, 1, Tm), spks=e, cond=cond.transpose(1, 2),
                               n_timesteps=10, streaming=True)
        return feat[:, :, pfeat.shape[1]:]

    N, Np, hop = 4 * (CHUNK // 2), CHUNK // 2, CHUNK // 2              # token hop = chunk / token_mel_ratio, prompt = one chunk
    token = torch.randint(0, c.vocab, (1, N), generator=g)
    ptoken = torch.randint(0, c.vocab, (1, Np), generator=g)
    pfeat = torch.randn(1, 2 * Np, 80, generator=g)
    emb = torch.randn(1, 192, generator=g)
    whole = ref_flow(token, ptoken, pfeat, emb, True)
    o_whole = flow_ref.flow_inference(token, emb, sd, c, prompt_token=ptoken, prompt_feat=pfeat, streaming=True)
    assert (o_whole - whole).abs().max().item() < 1e-4
    out.update(token=token.numpy(), ptoken=ptoken.numpy(), pfeat=pfeat.numpy(), emb=emb.numpy(), mel_whole=whole.numpy())
    worst = 0.0
    for k, i in enumerate(range(0, N, hop)):
        fin = i + hop + 3 >= N
        part = ref_flow(token[:, :i + hop + 3], ptoken, pfeat, emb, fin)
        o_part = flow_ref.flow_inference(token[:, :i + hop + 3], emb, sd, c, prompt_token=ptoken, prompt_feat=pfeat, streaming=True, finalize=fin)
        assert (o_part - part).abs().max().item() < 1e-4
        new = part[:, :, 2 * i:]
        worst = max(worst, (whole[:, :, 2 * i:2 * i + new.shape[2]] - new).abs().max().item())
        out['mel_chunk%d' % k] = part.numpy()
        out['mel_chunk%d_final' % k] = np.int32(fin)
    out['n_chunks'] = np.int32(k + 1)
    out['hop'] = np.int32(hop)
    print('[stream] flow N=%d prompt=%d hop=%d: %d chunks, chunked vs whole streaming mel max abs diff %.1e' % (N, Np, hop, k + 1, worst))

    # ---- HiFT finalize=False ------------------------------------------------------------------------------------------------
    hc = tiny_config().hift
    f0p = CausalConvRNNF0Predictor(num_class=1, in_channels=80, cond_channels=hc.f0_channels)
    gen = CausalHiFTGenerator(
        in_channels=80, base_channels=hc.base_channels, nb_harmonics=hc.nb_harmonics, sampling_rate=hc.sampling_rate,
        nsf_alpha=hc.nsf_alpha, nsf_sigma=hc.nsf_sigma, nsf_voiced_threshold=hc.nsf_voiced_threshold,
        upsample_rates=hc.upsample_rates, upsample_kernel_sizes=hc.upsample_kernel_sizes,
        istft_params={'n_fft': hc.n_fft, 'hop_len': hc.hop}, resblock_kernel_sizes=hc.resblock_kernel_sizes,
        resblock_dilation_sizes=hc.resblock_dilations, source_resblock_kernel_sizes=hc.source_resblock_kernel_sizes,
        source_resblock_dilation_sizes=hc.source_resblock_dilations, lrelu_slope=hc.lrelu_slope, audio_limit=hc.audio_limit,
        conv_pre_look_right=hc.conv_pre_look_right, f0_predictor=f0p).eval()
    seed_h, seed_t = 3, 9
    sdh = W.make_hift_state(hc, seed=seed_h, init='fan_in')
    gen.load_state_dict(sdh)
    tables = hift_ref.make_tables(hc, seed=seed_t)
    gen.m_source.l_sin_gen.rand_ini = tables['rand_ini']
    gen.m_source.l_sin_gen.sine_waves = tables['sine_waves']
    gen.m_source.uv = tables['uv']
    out.update(hift_weight_seed=np.int64(seed_h), hift_table_seed=np.int64(seed_t), hift_weight_sha=np.array(state_checksum(sdh)))
    Tm = 60
    mel = torch.randn(1, 80, Tm, generator=g)
    with torch.inference_mode():
        wav_whole, _ = gen.inference(speech_feat=mel)
    out.update(h_mel=mel.numpy(), h_wav_whole=wav_whole.numpy())
    worst, up = 0.0, hc.upsample_total
    for k, n in enumerate([12, 31, 47]):
        with torch.inference_mode():
            wav, s = gen.inference(speech_feat=mel[:, :, :n], finalize=False)
        o_wav, o_s = hift_ref.hift_inference(mel[:, :, :n], sdh, hc, tables, finalize=False)
        d = [(o_s - s).abs().max().item(), (o_wav - wav).abs().max().item()]
        assert wav.shape[1] == up * (n - 8) and s.shape[2] == up * (n - 3), (wav.shape, s.shape)
        assert d[0] < 1e-3, d
        worst = max(worst, (wav_whole[:, :wav.shape[1]] - wav).abs().max().item())
        print('[stream] hift finalize=False T=%d: oracle-reference max abs diff source %.1e wav %.1e' % (n, *d))
        out.update({'h%d_n' % k: np.int32(n), 'h%d_wav' % k: wav.numpy(), 'h%d_source' % k: s.numpy()})
    out['h_runs'] = np.int32(3)
    print('[stream] hift: chunk prefix vs whole-utterance wav max abs diff %.1e' % worst)
    np.savez_compressed(os.path.join(HERE, 'stream_tiny.npz'), **out)


# ------------------------------------------------------------------------------------------------
# checkpoint tooling: the MTP graft script, run as shipped
# ------------------------------------------------------------------------------------------------
def gen_graft():
    import subprocess
    import tempfile
    from flowmirror_hydravox_amd.checkpoint import graft_mtp_heads
    script = '/root/reference/scripts/post_process/add_mtp_weights_to_cosyvoice3lm_ckpt.py'
    out = {}
    cases = [dict(hidden=128, vocab=296, head_num=2, mtp_head_num=2, seed=7), dict(hidden=96, vocab=260, head_num=3, mtp_head_num=3, seed=1986)]
    for ci, cs in enumerate(cases):
        g = torch.Generator()
        g.manual_seed(100 + ci)
        sd = {'speech_embedding.weight': torch.randn(cs['vocab'], cs['hidden'], generator=g),
              'llm_decoder.weight': torch.randn(cs['vocab'], cs['hidden'], generator=g),
              'llm_decoder.bias': torch.randn(cs['vocab'], generator=g), 'step': torch.tensor(5)}
        if ci == 1:                                                      # an entry that already exists must survive untouched
            sd['mtp_block.0.input_layernorm.weight'] = torch.full((cs['hidden'],), 0.5)
        with tempfile.TemporaryDirectory() as d:
            torch.save(sd if ci == 0 else {'state_dict': sd, 'epoch': 3}, os.path.join(d, 'old.pt'))
            subprocess.run([sys.executable, script, '--src_ckpt', os.path.join(d, 'old.pt'), '--dst_ckpt', os.path.join(d, 'new.pt'),
                            '--head_num', str(cs['head_num']), '--mtp_head_num', str(cs['mtp_head_num']), '--seed', str(cs['seed'])],
                           check=True, capture_output=True)
            ref = torch.load(os.path.join(d, 'new.pt'))
        if ci == 1:
            assert ref['epoch'] == 3
            ref = ref['state_dict']
        mine, added = graft_mtp_heads(sd, head_num=cs['head_num'], mtp_head_num=cs['mtp_head_num'], seed=cs['seed'])
        assert set(mine) == set(ref) and all(torch.equal(mine[k], ref[k]) and mine[k].dtype == ref[k].dtype for k in ref)
        keys = sorted(ref)
        sha = [hashlib.sha256(ref[k].contiguous().view(torch.uint8).numpy().tobytes() if ref[k].dim() else ref[k].numpy().tobytes()).hexdigest()
               for k in keys]
        p = 'c%d_' % ci
        out.update({p + 'keys': np.array(keys), p + 'sha': np.array(sha), p + 'shapes': np.array([str(tuple(ref[k].shape)) for k in keys]),
                    p + 'dtypes': np.array([str(ref[k].dtype) for k in keys]), p + 'added': np.int32(added), p + 'input_seed': np.int32(100 + ci),
                    p + 'container': np.int32(ci == 1)})
        out.update({p + k: np.int32(v) for k, v in cs.items()})
        print('[graft] case %d: %d entries added by the script, ours bit-identical (%d keys)' % (ci, added, len(keys)))
    out['n_cases'] = np.int32(len(cases))
    np.savez_compressed(os.path.join(HERE, 'graft_tiny.npz'), **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['sampler', 'llm', 'flow', 'hift', 'matcha', 'stream', 'graft', 'llm_stress', 'llm_cv3w', 'flow_cv3w', 'hift_cv3w']
    for w in which:
        {'flow_half': gen_flow_half, 'llm_bf16': gen_llm_bf16, 'llm_cv3d': gen_llm_cv3d, 'flow_cv3d': gen_flow_cv3d, 'sampler_many': gen_sampler_many, 'llm_cv3w': gen_llm_cv3w, 'flow_cv3w': gen_flow_cv3w, 'hift_cv3w': gen_hift_cv3w, 'sampler': gen_sampler, 'llm': gen_llm, 'flow': gen_flow, 'hift': gen_hift, 'matcha': gen_matcha, 'stream': gen_stream, 'graft': gen_graft, 'llm_stress': gen_llm_stress}[w]()
    print('golden fixtures written to', HERE)
