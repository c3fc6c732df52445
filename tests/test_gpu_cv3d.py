"""Parity at FULL DEPTH — 24 LM layers, 22 DiT blocks, CV3 widths (config.cv3d_config) — against fixtures minted from the reference's own classes
at that depth (tests/golden/make_golden.py: gen_llm_cv3d, gen_flow_cv3d), plus the 1000-case sampler set (gen_sampler_many).

  * fp32 mode (north star contract): speech-token ids bit-exact at K in {1, 2, 4}, one by one and decoded together, incl. a 1020-row prefix
    that grows across context 1024 (layer strides of the KV cache, the 256-key splits at depth); first-step hidden / log-probs of all 5 heads;
    estimator and the 10-step solve within 1e-3 of the reference.
  * bf16 mode (production dtype): first step and estimator / mel against the bf16-faithful oracle at depth, bounds stated per test."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, state_checksum
from test_oracle_golden import cv3w_flow_inputs

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


@pytest.fixture(scope='module')
def cfg():
    from flowmirror_hydravox_amd.config import cv3d_config
    return cv3d_config()


# ---- sampler: 1000 reference cases ---------------------------------------------------------------------------------------------------------
def test_sampler_thousand_reference_cases_bit_exact():
    """ids and noise consumption of the HIP sampler == the reference's on all 1000 cases of sampler_many.npz (240 at V = 6761)"""
    from flowmirror_hydravox_amd import ops
    from oracle import sampler_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from sampler_cases import make_case, N_CASES
    g = load_golden('sampler_many.npz')
    cases = [make_case(i) for i in range(N_CASES)]
    keys = {}
    for i, c in enumerate(cases):
        assert int.from_bytes(hashlib.sha256(c['logp'].tobytes()).digest()[:8], 'little', signed=True) == int(g['logp_sha'][i]), i
        keys.setdefault((len(c['logp']), c['Vs'], c['top_k'], c['top_p'], c['win'], c['tau']), []).append(i)
    bad = []
    for (V, Vs, top_k, top_p, win, tau), idxs in keys.items():
        S = len(idxs)
        ncap = (max(int(g['consumed'][i]) for i in idxs) + 2 * V + 1023) // 1024 * 1024     # (what the reference consumed, plus one more draw of slack)
        H = max(1, max(len(cases[i]['hist']) for i in idxs))
        hist = torch.zeros(S, H, dtype=torch.int32)
        for j, i in enumerate(idxs):
            hist[j, :len(cases[i]['hist'])] = torch.tensor(cases[i]['hist'], dtype=torch.int32)
        hist_len = torch.tensor([len(cases[i]['hist']) for i in idxs], dtype=torch.int32)
        ign = torch.tensor([int(cases[i]['ignore_eos']) for i in idxs])
        min_len = torch.where(ign > 0, hist_len + 1, torch.zeros_like(hist_len)).to(torch.int32)
        logp = torch.stack([torch.from_numpy(cases[i]['logp']) for i in idxs]).view(S, 1, V)
        noise = torch.stack([torch.from_numpy(sampler_ref.NoiseStream(seed=cases[i]['seed']).take(ncap).copy()) for i in idxs])
        cur = torch.zeros(S, dtype=torch.int64).to(DEV)
        ids = ops.ras_sample(logp.to(DEV), hist.to(DEV), hist_len.to(DEV), min_len.to(DEV), noise.to(DEV), cur, speech_tokens=Vs, top_k=top_k,
                             top_p=top_p, win_size=win, rep_thresh=sampler_ref.rep_threshold(win, tau)).cpu().view(-1).tolist()
        cur = cur.cpu().tolist()
        for j, i in enumerate(idxs):
            want_id, want_c = int(g['id'][i]), int(g['consumed'][i])
            if ids[j] != want_id or (want_id >= 0 and cur[j] != want_c):
                bad.append((i, ids[j], want_id, cur[j], want_c))
    assert not bad, bad[:10]


# ---- LM, 24 layers ---------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def llm_setup(cfg):
    from flowmirror_hydravox_amd import weights as W
    g = load_golden('llm_cv3d.npz')
    sd = W.make_llm_state(cfg.llm, seed=int(g['weight_seed']), init='fan_in', with_lm_head=True)
    assert state_checksum(sd) == str(g['weight_sha'])
    top_p, top_k, win, tau = g['sampling']
    return g, sd, dict(top_p=float(top_p), top_k=int(top_k), win_size=int(win), tau_r=float(tau))


def _make_llm(cfg, sd, sampling, dtype, max_batch, max_ctx):
    from functools import partial
    from flowmirror_hydravox_amd.llm import HvxLLM
    from flowmirror_hydravox_amd.sampling import ras_sampling
    return HvxLLM(cfg.llm, sd, dtype=dtype, max_batch=max_batch, max_ctx=max_ctx, sampling=partial(ras_sampling, **sampling))


def _single(llm, g, r):
    p = 'r%d_' % r
    llm.inference_head_num = int(g[p + 'K'])
    text, ptext, ps = (torch.from_numpy(g[p + k])[None] for k in ('text', 'ptext', 'pspeech'))
    return list(llm.inference(text=text, text_len=torch.tensor([text.shape[1]], dtype=torch.int32), prompt_text=ptext,
                              prompt_text_len=torch.tensor([ptext.shape[1]], dtype=torch.int32), prompt_speech_token=ps if ps.shape[1] else None,
                              prompt_speech_token_len=torch.tensor([ps.shape[1]], dtype=torch.int32), embedding=torch.zeros(0, 192),
                              max_token_text_ratio=float(g[p + 'ratios'][0]), min_token_text_ratio=float(g[p + 'ratios'][1]), seed=int(g[p + 'seed'])))


def _batch(llm, g, runs):
    llm.inference_head_num = int(g['r%d_K' % runs[0]])
    return llm.generate_batch([torch.from_numpy(g['r%d_text' % r]) for r in runs], prompt_texts=[torch.from_numpy(g['r%d_ptext' % r]) for r in runs],
                              prompt_speech_tokens=[torch.from_numpy(g['r%d_pspeech' % r]) for r in runs], seeds=[int(g['r%d_seed' % r]) for r in runs],
                              max_token_text_ratio=[float(g['r%d_ratios' % r][0]) for r in runs],
                              min_token_text_ratio=[float(g['r%d_ratios' % r][1]) for r in runs])


def test_llm_24_layers_fp32_vs_reference(cfg, llm_setup):
    """fp32 mode at full depth: first-step hidden / log-probs of all 5 heads on a 142-row and a 1020-row prefix, and every token stream of the
    fixture bit-exact — one by one (K = 1, 2, 4; the last run starts at context 1020 and crosses 1024) and decoded together per K."""
    g, sd, sampling = llm_setup
    llm = _make_llm(cfg, sd, sampling, torch.float32, max_batch=4, max_ctx=1280)
    llm.inference_head_num = 5
    for r in (1, 6):
        p = 'r%d_' % r
        enc = llm._encode_prefix(torch.from_numpy(g[p + 'text']), torch.from_numpy(g[p + 'ptext']), torch.from_numpy(g[p + 'pspeech']))
        logp, y = llm.prefill_logp(enc)
        e = (_rel(y.cpu().numpy(), g[p + 'y_last']), np.abs(logp.cpu().numpy() - g[p + 'logps']).max())
        print('24 layers fp32, run %d (%d-row prefix): hidden rel %.1e, log-probs abs %.1e' % (r, len(enc), e[0], e[1]))
        assert e[0] < 5e-4 and e[1] < 2e-3, (r, e)
    n = int(g['n_runs'])
    for r in range(n):
        assert _single(llm, g, r) == g['r%d_tokens' % r].tolist(), r
    for K in (2, 4):
        runs = [r for r in range(n) if int(g['r%d_K' % r]) == K]
        assert len(runs) >= 2
        for r, toks in zip(runs, _batch(llm, g, runs)):
            assert toks == g['r%d_tokens' % r].tolist(), (K, r)
    assert max(2 + len(g['r%d_text' % r]) + len(g['r%d_ptext' % r]) + len(g['r%d_pspeech' % r]) + len(g['r%d_tokens' % r]) for r in range(n)) > 1024


def test_llm_24_layers_bf16_first_step_vs_bf16_oracle(cfg, llm_setup):
    """Production dtype at full depth against the bf16-faithful oracle (every operand rounded where the HIP path holds it in bf16): the floor
    of two faithful bf16 evaluations grows with depth (rounding-boundary flips per layer, tests/test_oracle_golden.py::test_cv3w_bf16_rounding_floor:
    5e-3 / 5e-2 through 2 layers); bounds here: hidden 3e-2 of its scale, log-probs 0.25 abs, the most likely token of every head unchanged
    or within that log-prob bound of the oracle's."""
    from oracle import llm_ref
    g, sd, sampling = llm_setup
    c = cfg.llm
    llm = _make_llm(cfg, sd, sampling, torch.bfloat16, max_batch=2, max_ctx=512)
    llm.inference_head_num = 5
    p = 'r1_'
    text, ptext, ps = (torch.from_numpy(g[p + k]) for k in ('text', 'ptext', 'pspeech'))
    logp, y = llm.prefill_logp(llm._encode_prefix(text, ptext, ps))
    yo = llm_ref.backbone(llm_ref.build_prefix(sd, c, text, ptext, ps, emu=True), sd, c, emu=True)[-1]
    lo = torch.stack(llm_ref.head_logps(yo, sd, c, c.head_num, emu=True))
    e = (_rel(y.cpu().numpy(), yo.numpy()), (logp.cpu() - lo).abs().max().item(), _rel(y.cpu().numpy(), g[p + 'y_last']),
         np.abs(logp.cpu().numpy() - g[p + 'logps']).max())
    print('24 layers bf16 first step: hidden %.2e / log-probs %.2e of the bf16-faithful oracle; %.2e / %.2e of the fp32 reference' % e)
    assert e[0] < 3e-2 and e[1] < 0.25, e
    # The reference deploys this LM in bf16 (`llm.eval().cuda().to(torch.bfloat16)`, infer_speech_model.py:102) with bf16 residual stream, norms and
    # softmax; its OWN bf16 run of this very prefix (tests/golden/llm_bf16.npz, minted on the CPU by make_golden.py: gen_llm_bf16) is 2.1e-2 / 0.15
    # from its fp32 run.  The product's bf16 mode (bf16 operands, fp32 residual stream / norms / softmax / accumulation) has to be CLOSER to fp32 than that.
    rb = load_golden('llm_bf16.npz')
    assert str(rb['weight_sha']) == str(g['weight_sha']) and np.abs(rb['y_last_f32'] - g[p + 'y_last']).max() < 1e-5
    ref_h, ref_l = float(rb['hidden_bf16_vs_f32']), float(rb['logp_bf16_vs_f32'])
    print('   the reference in bf16 (its deployed dtype) on the same prefix: hidden %.2e / log-probs %.2e of its fp32 run; product / reference = %.2f / %.2f'
          % (ref_h, ref_l, e[2] / ref_h, e[3] / ref_l))
    assert e[2] < ref_h and e[3] < ref_l, (e, ref_h, ref_l)
    for h in range(c.head_num):
        top = int(lo[h].argmax())
        assert float(logp[h].max().cpu()) - float(logp[h, top].cpu()) < 0.25, h


# ---- flow, 22 DiT blocks ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def flow_setup(cfg):
    from flowmirror_hydravox_amd import weights as W
    g = load_golden('flow_cv3d.npz')
    sd = W.make_flow_state(cfg.flow, seed=int(g['weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['weight_sha'])
    return g, sd


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_flow_22_blocks_vs_reference_and_bf16_oracle(cfg, flow_setup, dtype):
    """estimator at T = 256 with a padded second row and at T = 192 with the chunk mask, and the 10-step CFG solve with a prompt (10 x 22 block
    evaluations, the modulation cache at full depth).  fp32: within 1e-3 of the reference.  bf16: estimator within 3e-2 of the
    bf16-faithful oracle (chunk-masked case) and of the fp32 reference (padded case), 10-step mel within 2e-2 of the fp32 reference."""
    from flowmirror_hydravox_amd.flow import HvxFlow
    from oracle import flow_ref
    g, sd = flow_setup
    c = cfg.flow
    # (the historical bf16 mode: bf16 operands everywhere + fp16 stream; the default handle adds fp16 block Linears and fp32 small Linears, tested below)
    flow = HvxFlow(c, sd, dtype=dtype, max_t=512, f16_linears=False, f32_small=False)
    for tag in ('e0', 'e1'):
        T, lens, streaming = int(g[tag + '_T']), g[tag + '_lens'].tolist(), bool(g[tag + '_streaming'])
        x, mask, mu, spk, cond = cv3w_flow_inputs(int(g[tag + '_seed']), T, lens)
        assert state_checksum(dict(x=x, mu=mu, spk=spk, cond=cond)) == str(g[tag + '_in_sha'])
        t = torch.from_numpy(g[tag + '_t'])
        est = (flow.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu() * mask).numpy()
        e_ref = _rel(est, g[tag + '_out'])
        if dtype == torch.float32:
            assert e_ref < 1e-3, (tag, e_ref)
        elif tag == 'e1':                      # (the bf16-faithful oracle takes ~1 s per block on the host: the shorter case only)
            emu = (flow_ref.dit_forward(x, mask, mu, t, spk, cond, sd, c, streaming=streaming, emu=True, resid16=True) * mask).numpy()
            e_emu = _rel(est, emu)
            print('22 blocks, %s bf16 estimator: %.2e of the bf16-faithful oracle, %.2e of the fp32 reference' % (tag, e_emu, e_ref))
            assert e_emu < 3e-2, (tag, e_emu, e_ref)
            # The reference deploys this network in fp16 (`flow.eval().cuda().half()`, infer_speech_model.py:103): its OWN fp16 run of this very case
            # (tests/golden/flow_half.npz, minted on the CPU by make_golden.py: gen_flow_half) is 1.4e-3 of the output scale from its fp32 run.
            # bf16 operands carry 8x the rounding unit of fp16; the product's bf16 mode has to stay within 6x the reference's own distance.
            h = load_golden('flow_half.npz')
            assert str(h['d_in_sha']) == str(g[tag + '_in_sha']) and np.abs(h['d_out_f32'] - g[tag + '_out']).max() < 1e-5
            e_half = float(h['d_half_vs_f32'])
            print('   the reference in fp16 (its deployed dtype) on the same case: %.2e of the fp32 reference; product bf16 / reference fp16 = %.1f' % (e_half, e_ref / e_half))
            assert e_ref < 6.0 * e_half, (e_ref, e_half)
        else:
            print('22 blocks, %s bf16 estimator: %.2e of the fp32 reference' % (tag, e_ref))
            assert e_ref < 3e-2, (tag, e_ref)
    token, ptoken, pfeat, emb = (torch.from_numpy(g[k]) for k in ('token', 'ptoken', 'pfeat', 'emb'))
    mel, _ = flow.inference(token=token.to(DEV), token_len=torch.tensor([token.shape[1]], dtype=torch.int32), embedding=emb.to(DEV), finalize=True,
                            prompt_token=ptoken.to(DEV), prompt_token_len=torch.tensor([ptoken.shape[1]], dtype=torch.int32),
                            prompt_feat=pfeat.to(DEV), prompt_feat_len=torch.tensor([pfeat.shape[1]], dtype=torch.int32))
    assert tuple(mel.shape) == g['mel'].shape
    if dtype == torch.float32:
        assert _rel(mel.cpu().numpy(), g['mel']) < 1e-3, _rel(mel.cpu().numpy(), g['mel'])
    else:
        # (220 block evaluations of the bf16-faithful oracle would take minutes on the host: the 10-step mel is held to the reference's fp32 mel;
        # measured 3.3e-3, of which 1.8e-3 is the distance to the bf16-faithful oracle)
        e = _rel(mel.cpu().numpy(), g['mel'])
        print('22 blocks, bf16 10-step mel: %.2e of the fp32 reference' % e)
        assert e < 2e-2, e


def test_flow_22_blocks_fp16_linear_operands(cfg, flow_setup):
    """bf16 mode with the four Linears of every DiT block on IEEE fp16 operands (HvxFlow(f16_linears=True) -> hvx_flow_set_f16_linears; the attention and
    q / k / v stay bf16, the residual stream is fp16): the reference deploys this decoder in fp16, and with fp16 Linear operands the product sits at the
    block Linears stop contributing to the distance from fp32 — what remains (tools/dit_rounding_study.py) is the bf16 of the input / output projections
    and of the adaLN modulation Linears, which this option does not touch yet.  Chunk-masked estimator (T = 192: few tiles, the 256-tile kernel takes
    every fp16 Linear whatever the size) against the faithful oracle with the same rounding points and against the fp32 reference; padded estimator at
    T = 256; 10-step mel with a prompt."""
    from flowmirror_hydravox_amd.flow import HvxFlow
    from oracle import flow_ref
    g, sd = flow_setup
    c = cfg.flow
    flow = HvxFlow(c, sd, dtype=torch.bfloat16, max_t=512, f16_linears=True, f32_small=False)
    assert flow.f16_linears and flow.half_stream and not flow.f32_small
    h = load_golden('flow_half.npz')
    e_half = float(h['d_half_vs_f32'])
    for tag in ('e0', 'e1'):
        T, lens, streaming = int(g[tag + '_T']), g[tag + '_lens'].tolist(), bool(g[tag + '_streaming'])
        x, mask, mu, spk, cond = cv3w_flow_inputs(int(g[tag + '_seed']), T, lens)
        t = torch.from_numpy(g[tag + '_t'])
        est = (flow.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu() * mask).numpy()
        e_ref = _rel(est, g[tag + '_out'])
        if tag == 'e1':
            emu = (flow_ref.dit_forward(x, mask, mu, t, spk, cond, sd, c, streaming=streaming, emu=True, resid16=True, lin16=True) * mask).numpy()
            e_emu = _rel(est, emu)
            print('22 blocks, fp16 Linear operands, %s estimator: %.2e of the faithful oracle, %.2e of the fp32 reference (the reference in fp16: %.2e)' % (tag, e_emu, e_ref, e_half))
            assert e_emu < 1e-2 and e_ref < 5e-3, (e_emu, e_ref, e_half)
        else:
            print('22 blocks, fp16 Linear operands, %s estimator: %.2e of the fp32 reference' % (tag, e_ref))
            assert e_ref < 5e-3, e_ref
    token, ptoken, pfeat, emb = (torch.from_numpy(g[k]) for k in ('token', 'ptoken', 'pfeat', 'emb'))
    mel, _ = flow.inference(token=token.to(DEV), token_len=torch.tensor([token.shape[1]], dtype=torch.int32), embedding=emb.to(DEV), finalize=True,
                            prompt_token=ptoken.to(DEV), prompt_token_len=torch.tensor([ptoken.shape[1]], dtype=torch.int32),
                            prompt_feat=pfeat.to(DEV), prompt_feat_len=torch.tensor([pfeat.shape[1]], dtype=torch.int32))
    e = _rel(mel.cpu().numpy(), g['mel'])
    print('22 blocks, fp16 Linear operands, 10-step mel: %.2e of the fp32 reference' % e)
    assert e < 5e-3, e


def test_flow_22_blocks_reference_precision_mode(cfg, flow_setup):
    """bf16 mode + fp16 stream + fp16 block Linears + fp32 small Linears (HvxFlow(f16_linears=True, f32_small=True)): the configuration the host study
    (tools/dit_rounding_study.py) puts at the reference's OWN fp16 distance from fp32.  Estimators against the faithful oracle with the same rounding
    points and against the fp32 reference — held to 1.6x the distance of the reference's own fp16 run of the same case (tests/golden/flow_half.npz) — and
    the 10-step mel with a prompt."""
    from flowmirror_hydravox_amd.flow import HvxFlow
    from oracle import flow_ref
    g, sd = flow_setup
    c = cfg.flow
    flow = HvxFlow(c, sd, dtype=torch.bfloat16, max_t=512)                      # the default bf16 handle
    assert flow.f16_linears and flow.f32_small and flow.half_stream
    e_half = float(load_golden('flow_half.npz')['d_half_vs_f32'])
    for tag in ('e0', 'e1'):
        T, lens, streaming = int(g[tag + '_T']), g[tag + '_lens'].tolist(), bool(g[tag + '_streaming'])
        x, mask, mu, spk, cond = cv3w_flow_inputs(int(g[tag + '_seed']), T, lens)
        t = torch.from_numpy(g[tag + '_t'])
        est = (flow.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu() * mask).numpy()
        e_ref = _rel(est, g[tag + '_out'])
        if tag == 'e1':
            emu = (flow_ref.dit_forward(x, mask, mu, t, spk, cond, sd, c, streaming=streaming, emu=True, resid16=True, lin16=True, small32=True) * mask).numpy()
            e_emu = _rel(est, emu)
            print('22 blocks, reference-precision mode, %s estimator: %.2e of the faithful oracle, %.2e of the fp32 reference (the reference in fp16: %.2e)' % (tag, e_emu, e_ref, e_half))
            assert e_emu < 5e-3 and e_ref < 1.6 * e_half, (e_emu, e_ref, e_half)
        else:
            print('22 blocks, reference-precision mode, %s estimator: %.2e of the fp32 reference' % (tag, e_ref))
            assert e_ref < 3e-3, e_ref
    token, ptoken, pfeat, emb = (torch.from_numpy(g[k]) for k in ('token', 'ptoken', 'pfeat', 'emb'))
    mel, _ = flow.inference(token=token.to(DEV), token_len=torch.tensor([token.shape[1]], dtype=torch.int32), embedding=emb.to(DEV), finalize=True,
                            prompt_token=ptoken.to(DEV), prompt_token_len=torch.tensor([ptoken.shape[1]], dtype=torch.int32),
                            prompt_feat=pfeat.to(DEV), prompt_feat_len=torch.tensor([pfeat.shape[1]], dtype=torch.int32))
    e = _rel(mel.cpu().numpy(), g['mel'])
    print('22 blocks, reference-precision mode, 10-step mel: %.2e of the fp32 reference' % e)
    assert e < 2.5e-3, e

