#!/usr/bin/env python3
"""Mint golden vectors AT THE BENCHMARKED SHAPES by running the REFERENCE implementation (build container only; minutes of CPU per part).

    python tests/golden/make_golden_fullsize.py [flow_est] [flow_solve] [hift] [hift_cond] [llm] [single]

BASELINE.json configs[1] = 512-char utterances: 2816 speech tokens -> 5632 mel frames -> 2 703 360 samples, LM contexts 514 .. 3330 rows.  The small
fixtures of make_golden.py stop at T = 256 (22 blocks), 160 vocoder frames and a 1100-row LM context; the kernels the benchmark actually runs
(64-rows-per-wave DiT attention from 2048 rows, 256-tile Linears over 45056 rows, the vocoder's 675 841-row last stage, 7 decode-attention
splits at contexts > 3072) are pinned here to ONE run each of the reference's own modules at full depth and full length:

  flow_est   cosyvoice/flow/DiT/dit.py:145-176       DiT.forward, 22 blocks, T = 5632, B = 2 (second row padded to 5000 frames)   -> flow_full_est.npz
  flow_solve cosyvoice/flow/flow_matching.py:204-228  CausalConditionalCFM.forward, 10 Euler steps x CFG 2 over 2816 tokens         -> flow_full_solve.npz
  hift       cosyvoice/hifigan/generator.py:713-726   CausalHiFTGenerator.inference at 5632 frames                                   -> hift_full.npz
  llm        cosyvoice/llm/llm_multi_head_v3.py:248-260, 886-888  first-step hidden / 5-head log-probs on a 3300-row prefix + the
             reference's own (uncached) K = 2 generation from that context                                                            -> llm_full.npz
  single     BASELINE configs[0]: head_num = 1, one 64-char utterance (64 text -> 352 tokens -> 704 frames) at full CV3 depth:
             ids, mel, f0 and waveform of the reference's three stages run back to back                                               -> single_cv3.npz

The fixtures hold data only (seeds, checksums, expected outputs); weights and inputs are regenerated from the recorded seeds.
"""
import os
import sys
import time
import hashlib

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (installs the import shim, puts the repo root on sys.path)

from flowmirror_hydravox_amd.config import cv3_config  # noqa: E402
from flowmirror_hydravox_amd import weights as W  # noqa: E402
from oracle import sampler_ref, llm_ref, flow_ref, hift_ref  # noqa: E402

DictConfig = MG.DictConfig
torch.set_grad_enabled(False)


def sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def build_ref_flow(c):
    from cosyvoice.flow.flow import CausalMaskedDiffWithDiT
    from cosyvoice.flow.flow_matching import CausalConditionalCFM
    from cosyvoice.flow.DiT.dit import DiT
    from cosyvoice.transformer.upsample_encoder import PreLookaheadLayer
    dit = DiT(dim=c.dim, depth=c.depth, heads=c.heads, dim_head=c.head_dim, ff_mult=c.ff_mult, mel_dim=c.mel, mu_dim=c.mel,
              spk_dim=c.mel, out_channels=c.mel, static_chunk_size=c.static_chunk_size)
    cfm = CausalConditionalCFM(in_channels=240, cfm_params=DictConfig(sigma_min=1e-6, solver='euler', t_scheduler='cosine',
                                                                     training_cfg_rate=0.2, inference_cfg_rate=c.cfg_rate, reg_loss_type='l1'),
                               n_spks=1, spk_emb_dim=80, estimator=dit)
    pla = PreLookaheadLayer(in_channels=80, channels=c.pre_lookahead_channels, pre_lookahead_len=c.pre_lookahead_len)
    flow = CausalMaskedDiffWithDiT(input_size=80, output_size=80, spk_embed_dim=192, vocab_size=c.vocab, token_mel_ratio=2,
                                   pre_lookahead_len=3, pre_lookahead_layer=pla, decoder=cfm).eval()
    MG.assert_spec(flow, W.flow_spec(c), 'flow.pt (cv3)')
    sd = W.make_flow_state(c, seed=1987, init='fan_in')
    flow.load_state_dict(sd)
    return flow, dit, sd


def ref_flow_mel(flow, token, emb):
    """what CausalMaskedDiffWithDiT.inference computes (flow.py:367-430), driven below its dtype wrapper (SURVEY.md finding 7), no prompt"""
    e = flow.spk_embed_affine_layer(F.normalize(emb, dim=1))
    h = flow.pre_lookahead_layer(flow.input_embedding(token)).repeat_interleave(2, dim=1)
    T = h.shape[1]
    feat, _ = flow.decoder(mu=h.transpose(1, 2).contiguous(), mask=torch.ones(1, 1, T), spks=e, cond=torch.zeros(1, 80, T), n_timesteps=10, streaming=False)
    return feat


def gen_flow_est():
    c = cv3_config().flow
    flow, dit, sd = build_ref_flow(c)
    T, lens, seed = 5632, [5632, 5000], 61
    x, mask, mu, spk, cond = MG.cv3w_flow_inputs(seed, T, lens)
    t = torch.tensor([0.35, 0.35])
    t0 = time.time()
    est = dit(x, mask, mu, t, spk, cond, streaming=False) * mask
    print('[flow-full] reference DiT, 22 blocks, T = %d, lens %s: %.0f s; out absmax %.3f' % (T, lens, time.time() - t0, est.abs().max()))
    np.savez_compressed(os.path.join(HERE, 'flow_full_est.npz'), weight_seed=np.int64(1987), weight_sha=np.array(MG.state_checksum(sd)),
                        seed=np.int64(seed), T=np.int32(T), lens=np.array(lens, dtype=np.int32), t=t.numpy(),
                        in_sha=np.array(MG.state_checksum(dict(x=x, mu=mu, spk=spk, cond=cond))), out=est.numpy())


def gen_flow_solve():
    c = cv3_config().flow
    flow, dit, sd = build_ref_flow(c)
    g = torch.Generator().manual_seed(62)
    token = torch.randint(0, c.vocab, (1, 2816), generator=g)
    emb = torch.randn(1, 192, generator=g)
    t0 = time.time()
    mel = ref_flow_mel(flow, token, emb)
    print('[flow-full] reference 10-step CFG solve of 2816 tokens (T = %d): %.0f s; mel absmax %.3f' % (mel.shape[-1], time.time() - t0, mel.abs().max()))
    np.savez_compressed(os.path.join(HERE, 'flow_full_solve.npz'), weight_seed=np.int64(1987), weight_sha=np.array(MG.state_checksum(sd)),
                        token=token.numpy().astype(np.int32), emb=emb.numpy(), mel=mel.numpy())


def gen_flow_prompt():
    """BASELINE configs[3]'s flow call at full depth: CausalMaskedDiffWithDiT.inference with a 3 s prompt (75 prompt tokens + 150 prompt mel frames, flow.py:389-430:
    prompt tokens prepended, prompt mel as `cond`, the prompt frames cut from the result) and 1408 tokens -> flow_full_prompt.npz"""
    c = cv3_config().flow
    flow, dit, sd = build_ref_flow(c)
    g = torch.Generator().manual_seed(63)
    N, Np = 1408, 75
    token = torch.randint(0, c.vocab, (1, N), generator=g)
    ptoken = torch.randint(0, c.vocab, (1, Np), generator=g)
    pfeat = torch.randn(1, 2 * Np, 80, generator=g)
    emb = torch.randn(1, 192, generator=g)
    t0 = time.time()
    e = flow.spk_embed_affine_layer(F.normalize(emb, dim=1))
    h = flow.pre_lookahead_layer(flow.input_embedding(torch.cat([ptoken, token], dim=1))).repeat_interleave(2, dim=1)
    T = h.shape[1]
    cond = torch.zeros(1, T, 80)
    cond[:, :2 * Np] = pfeat
    feat, _ = flow.decoder(mu=h.transpose(1, 2).contiguous(), mask=torch.ones(1, 1, T), spks=e, cond=cond.transpose(1, 2), n_timesteps=10, streaming=False)
    feat = feat[:, :, 2 * Np:]
    print('[flow-full] reference solve with a %d-token / %d-frame prompt, %d tokens (T = %d): %.0f s; mel absmax %.3f' % (Np, 2 * Np, N, T, time.time() - t0, feat.abs().max()))
    o_feat = flow_ref.flow_inference(token, emb, sd, c, prompt_token=ptoken, prompt_feat=pfeat)
    d = (o_feat - feat).abs().max().item()
    assert d < 2e-3, d
    print('[flow-full] oracle-reference max abs diff %.1e' % d)
    np.savez_compressed(os.path.join(HERE, 'flow_full_prompt.npz'), weight_seed=np.int64(1987), weight_sha=np.array(MG.state_checksum(sd)),
                        token=token.numpy().astype(np.int32), ptoken=ptoken.numpy().astype(np.int32), pfeat=pfeat.numpy(), emb=emb.numpy(), mel=feat.numpy())


def build_ref_hift(c, seed_w=1988, seed_t=9):
    from cosyvoice.hifigan.generator import CausalHiFTGenerator
    from cosyvoice.hifigan.f0_predictor import CausalConvRNNF0Predictor
    f0p = CausalConvRNNF0Predictor(num_class=1, in_channels=80, cond_channels=c.f0_channels)
    gen = CausalHiFTGenerator(
        in_channels=80, base_channels=c.base_channels, nb_harmonics=c.nb_harmonics, sampling_rate=c.sampling_rate,
        nsf_alpha=c.nsf_alpha, nsf_sigma=c.nsf_sigma, nsf_voiced_threshold=c.nsf_voiced_threshold,
        upsample_rates=c.upsample_rates, upsample_kernel_sizes=c.upsample_kernel_sizes,
        istft_params={'n_fft': c.n_fft, 'hop_len': c.hop}, resblock_kernel_sizes=c.resblock_kernel_sizes,
        resblock_dilation_sizes=c.resblock_dilations, source_resblock_kernel_sizes=c.source_resblock_kernel_sizes,
        source_resblock_dilation_sizes=c.source_resblock_dilations, lrelu_slope=c.lrelu_slope, audio_limit=c.audio_limit,
        conv_pre_look_right=c.conv_pre_look_right, f0_predictor=f0p).eval()
    MG.assert_spec(gen, W.hift_spec(c), 'hift.pt (cv3)')
    sd = W.make_hift_state(c, seed=seed_w, init='fan_in')
    gen.load_state_dict(sd)
    tables = hift_ref.make_tables(c, seed=seed_t)
    gen.m_source.l_sin_gen.rand_ini = tables['rand_ini']
    gen.m_source.l_sin_gen.sine_waves = tables['sine_waves']
    gen.m_source.uv = tables['uv']
    return gen, sd, tables


def pack_wave(out, p, wav, s):
    """a 2.7 M-sample waveform as a fixture: every sample in fp16 (5e-4 absolute), every 16th + the first / last 32 k samples in fp32, checksums;
    the source (same length) as head / tail / stride samples + checksum: the tests recompute it from the stored f0 with the oracle"""
    w, s = wav.reshape(-1), s.reshape(-1)
    out.update({p + 'wav_f16': w.numpy().astype(np.float16), p + 'wav_s16': w[::16].numpy().copy(), p + 'wav_head': w[:32768].numpy().copy(),
                p + 'wav_tail': w[-32768:].numpy().copy(), p + 'wav_sha': np.array(sha(w)),
                p + 'src_s16': s[::16].numpy().copy(), p + 'src_head': s[:32768].numpy().copy(), p + 'src_tail': s[-32768:].numpy().copy(), p + 'src_sha': np.array(sha(s))})


def gen_hift():
    c = cv3_config().hift
    gen, sd, tables = build_ref_hift(c)
    T = 5632
    g = torch.Generator().manual_seed(11)
    mel = torch.randn(1, c.mel, T, generator=g)             # (N(0, 1) mel: 0.01 % of the samples clip; the N(-4, 1.5) mel of test_gpu_fullsize.py clips 40 % of them, which hides errors)
    t0 = time.time()
    with torch.inference_mode():
        f0 = gen.f0_predictor(mel)
        wav, s = gen.inference(speech_feat=mel)
    print('[hift-full] reference CausalHiFTGenerator.inference, T = %d: %.0f s; wav std %.3f, clipped %.2f %%, f0 max %.1f, voiced %.0f %%'
          % (T, time.time() - t0, wav.std(), 100.0 * (wav.abs() >= c.audio_limit).float().mean(), f0.max(), 100.0 * (f0 > 10).float().mean()))
    out = dict(weight_seed=np.int64(1988), table_seed=np.int64(9), weight_sha=np.array(MG.state_checksum(sd)), T=np.int32(T), mel_seed=np.int64(11),
               mel_sha=np.array(sha(mel)), f0=f0.numpy())
    pack_wave(out, '', wav, s)
    np.savez_compressed(os.path.join(HERE, 'hift_full.npz'), **out)
    return gen, sd, tables, mel, f0, wav, s


def gen_hift_cond(which='full'):
    """The REFERENCE's own conditioning number for the end-to-end vocoder output (generator.py:713-726; the ill-conditioned step is SineGen2's fp32
    `cumsum(rad) * 2 pi`, scaled by 480, :254-260): the reference's m_source + decode are run again on the SAME mel with its f0 perturbed by what another
    convolution summation order changes — uniform noise of +-5e-4 Hz (two draws), +-1 ulp, and the f0 predictor evaluated in fp64 — and the distance of
    each waveform to the unperturbed one is recorded sample-wise (first second / every second / whole utterance) and under the phase-insensitive
    distances of tests/wave_metrics.py.  A REAL error of known size (the mel perturbed by 1e-2 N(0,1); the source scaled by 1.01) is recorded under the
    same distances, to show what they do detect.  which = 'full': the 5632-frame mel of hift_full.npz -> hift_full_cond.npz; 'single': the 704-frame mel of
    single_cv3.npz (BASELINE configs[0]) -> single_cv3_cond.npz (scalars and per-second profiles only)."""
    import copy
    sys.path.insert(0, os.path.dirname(HERE))
    import wave_metrics as WM
    c = cv3_config().hift
    gen, sd, tables = build_ref_hift(c)
    if which == 'full':
        g0 = np.load(os.path.join(HERE, 'hift_full.npz'))
        T = int(g0['T'])
        mel = torch.randn(1, c.mel, T, generator=torch.Generator().manual_seed(int(g0['mel_seed'])))
        assert sha(mel) == str(g0['mel_sha'])
        out_name = 'hift_full_cond.npz'
    else:
        g0 = np.load(os.path.join(HERE, 'single_cv3.npz'))
        assert MG.state_checksum(sd) == str(g0['hift_sha'])
        mel = torch.from_numpy(g0['mel'])
        T = mel.shape[-1]
        out_name = 'single_cv3_cond.npz'

    def run(f0, mel_=None, gain=1.0):
        with torch.inference_mode():
            s = gen.f0_upsamp(f0[:, None]).transpose(1, 2)
            s, _, _ = gen.m_source(s)
            s = s.transpose(1, 2) * gain
            return gen.decode(x=mel if mel_ is None else mel_, s=s, finalize=True).reshape(-1).numpy()

    t0 = time.time()
    with torch.inference_mode():
        f0 = gen.f0_predictor(mel)
        f0_64 = copy.deepcopy(gen.f0_predictor).double()(mel.double()).float()
    assert np.array_equal(f0.numpy(), g0['f0'])
    base = run(f0)
    assert np.abs(base[::16] - g0['wav_s16']).max() == 0.0                      # the run of the fixture, reproduced bit for bit
    print('[hift-cond %s] baseline reproduced in %.0f s; f0(fp64 predictor) - f0(fp32): max %.2e Hz' % (which, time.time() - t0, (f0_64 - f0).abs().max()))
    gn = torch.Generator().manual_seed(4711)
    cases = {
        'noise_a': f0 + (torch.rand(f0.shape, generator=gn) * 2 - 1) * 5e-4,
        'noise_b': f0 + (torch.rand(f0.shape, generator=gn) * 2 - 1) * 5e-4,
        'ulp_up': torch.nextafter(f0, torch.full_like(f0, 1e9)),
        'ulp_down': torch.nextafter(f0, torch.full_like(f0, -1e9)),
        'f0_fp64': f0_64,
    }
    sec = 24000
    out = dict(T=np.int32(T), f0_fp64_minus_fp32=np.float64((f0_64 - f0).abs().max()), peak=np.float64(np.abs(base).max()),
               names=np.array(list(cases) + ['real_mel_1e-2', 'real_source_gain_1.01']))

    def record(name, w):
        d = np.abs(w - base)
        n = (d.size // sec) * sec
        prof = d[:n].reshape(-1, sec).max(axis=1)
        m = WM.all_metrics(w, base)
        out[name + '_first_second'] = np.float64(d[:sec].max())
        out[name + '_max'] = np.float64(d.max())
        out[name + '_l2'] = np.float64(np.linalg.norm(w - base) / np.linalg.norm(base))
        out[name + '_per_second'] = prof.astype(np.float32)
        for k, v in m.items():
            out[name + '_' + k] = np.float64(v)
        print('[hift-cond %s] %-22s sample-wise: first second %.2e, whole %.2e (L2 %.2e) | %s' %
              (which, name, d[:sec].max(), d.max(), out[name + '_l2'], ' '.join('%s %.2e' % kv for kv in m.items())))

    for name, f in cases.items():
        record(name, run(f))
    record('real_mel_1e-2', run(f0, mel_=mel + 1e-2 * torch.randn(mel.shape, generator=gn)))
    record('real_source_gain_1.01', run(f0, gain=1.01))
    np.savez_compressed(os.path.join(HERE, out_name), **out)


def gen_llm():
    c = cv3_config().llm
    sd = W.make_llm_state(c, seed=1986, init='fan_in', with_lm_head=True)
    sampling = dict(top_p=0.9, top_k=10, win_size=32, tau_r=0.2)
    lm = MG.build_ref_llm(c, sd, sampling)
    out = dict(weight_seed=np.int64(1986), weight_sha=np.array(MG.state_checksum(sd)),
               sampling=np.array([sampling['top_p'], sampling['top_k'], sampling['win_size'], sampling['tau_r']], dtype=np.float64))
    n_text, n_ps, K, seed = 512, 2786, 2, 801                                   # 2 + 512 + 2786 = 3300 prefix rows
    g = torch.Generator().manual_seed(seed)
    text = torch.randint(0, c.text_vocab, (1, n_text), dtype=torch.int32, generator=g)
    pspeech = torch.randint(0, c.speech_tokens, (1, n_ps), dtype=torch.int32, generator=g)
    ptext = torch.zeros(1, 0, dtype=torch.int32)
    # ---- first step: hidden state + log-probs of all 5 heads (llm_multi_head_v3.py:248-260, 886-888) ----
    t0 = time.time()
    lm_input = llm_ref.build_prefix(sd, c, text[0], ptext[0], pspeech[0])[None]
    L = lm_input.shape[1]
    assert L == 3300
    y, _ = lm.llm.forward_one_step(lm_input, masks=torch.tril(torch.ones(1, L, L)).bool(), cache=None)
    last = y[:, -1:, :]
    logps = torch.stack([lm.llm_decoder(lm.mtp_block[j](last)[0][:, -1]).log_softmax(dim=-1)[0] for j in range(c.head_num)])
    print('[llm-full] reference first step on a %d-row prefix: %.0f s (|y| max %.2f)' % (L, time.time() - t0, y.abs().max()))
    out.update(text=text[0].numpy(), pspeech=pspeech[0].numpy(), K=np.int32(K), seed=np.int64(seed), y_last=last[0, 0].numpy(), logps=logps.numpy(),
               y_rows=y[0, [0, 1, 255, 256, 511, 512, 1023, 1024, 2047, 2048, 3071, 3072, 3298]].numpy())
    # ---- the reference's own generation from that context (no KV cache: the whole prefix per step), K = 2, 12 steps ----
    t0 = time.time()
    lm.inference_head_num = K
    n_gen = 24
    torch.manual_seed(seed)
    toks = []
    for tok in lm.inference(text=text, text_len=torch.tensor([n_text], dtype=torch.int32), prompt_text=ptext, prompt_text_len=torch.tensor([0], dtype=torch.int32),
                            prompt_speech_token=pspeech, prompt_speech_token_len=torch.tensor([n_ps], dtype=torch.int32), embedding=torch.zeros(0, 192),
                            max_token_text_ratio=20, min_token_text_ratio=2):
        toks.append(int(tok))
        if len(toks) >= n_gen:
            break
    print('[llm-full] reference generation, K = %d, contexts %d .. %d: %d tokens in %.0f s' % (K, L, L + len(toks), len(toks), time.time() - t0))
    ns = sampler_ref.NoiseStream(seed=seed)
    otoks = []
    for tok in llm_ref.llm_inference(sd, c, text[0], ns, prompt_text=ptext[0], prompt_speech_token=pspeech[0], inference_head_num=K, sampling=sampling,
                                     max_token_text_ratio=20, min_token_text_ratio=2, use_kv_cache=True):
        otoks.append(int(tok))
        if len(otoks) >= n_gen:
            break
    assert otoks == toks, (otoks, toks)
    print('[llm-full] oracle (KV-cached) == reference on those %d tokens' % len(toks))
    out.update(tokens=np.array(toks, dtype=np.int32))
    np.savez_compressed(os.path.join(HERE, 'llm_full.npz'), **out)


def gen_single():
    """BASELINE configs[0]: inference_head_num = 1, ONE 64-char utterance — 64 text ids -> 352 speech tokens (min = max ratio 5.5) ->
    704 mel frames -> 337 920 samples — through the reference's llm.inference (uncached, as shipped), flow decoder and vocoder at full CV3 depth."""
    cfg = cv3_config()
    c = cfg.llm
    sd = W.make_llm_state(c, seed=1986, init='fan_in', with_lm_head=True)
    sampling = dict(top_p=0.9, top_k=10, win_size=32, tau_r=0.2)              # router.py:22-44 defaults
    lm = MG.build_ref_llm(c, sd, sampling)
    lm.inference_head_num = 1
    seed, n_text = 901, 64
    g = torch.Generator().manual_seed(seed)
    text = torch.randint(0, c.text_vocab, (1, n_text), dtype=torch.int32, generator=g)
    emb = torch.randn(1, 192, generator=g)
    t0 = time.time()
    torch.manual_seed(seed)
    toks = [int(t) for t in lm.inference(
        text=text, text_len=torch.tensor([n_text], dtype=torch.int32), prompt_text=torch.zeros(1, 0, dtype=torch.int32), prompt_text_len=torch.tensor([0], dtype=torch.int32),
        prompt_speech_token=None, prompt_speech_token_len=torch.tensor([0], dtype=torch.int32), embedding=torch.zeros(0, 192),
        max_token_text_ratio=5.5, min_token_text_ratio=5.5)]
    t_llm = time.time() - t0
    print('[single] reference llm.inference (no KV cache), K = 1: %d tokens in %.0f s' % (len(toks), t_llm))
    otoks = list(llm_ref.llm_inference(sd, c, text[0], sampler_ref.NoiseStream(seed=seed), inference_head_num=1, sampling=sampling,
                                       max_token_text_ratio=5.5, min_token_text_ratio=5.5, use_kv_cache=True))
    assert otoks == toks
    del lm
    flow, dit, fsd = build_ref_flow(cfg.flow)
    t0 = time.time()
    mel = ref_flow_mel(flow, torch.tensor(toks)[None], emb)
    t_flow = time.time() - t0
    print('[single] reference flow, %d frames: %.0f s' % (mel.shape[-1], t_flow))
    del flow, dit
    gen, hsd, tables = build_ref_hift(cfg.hift)
    t0 = time.time()
    with torch.inference_mode():
        f0 = gen.f0_predictor(mel)
        wav, s = gen.inference(speech_feat=mel)
    t_hift = time.time() - t0
    print('[single] reference vocoder: %.0f s; %d samples' % (t_hift, wav.numel()))
    out = dict(llm_sha=np.array(MG.state_checksum(sd)), flow_sha=np.array(MG.state_checksum(fsd)), hift_sha=np.array(MG.state_checksum(hsd)),
               seed=np.int64(seed), text=text[0].numpy(), emb=emb.numpy(), tokens=np.array(toks, dtype=np.int32), mel=mel.numpy(), f0=f0.numpy(),
               sampling=np.array([sampling['top_p'], sampling['top_k'], sampling['win_size'], sampling['tau_r']], dtype=np.float64),
               ref_seconds=np.array([t_llm, t_flow, t_hift]), ref_threads=np.int32(torch.get_num_threads()))
    pack_wave(out, '', wav, s)
    np.savez_compressed(os.path.join(HERE, 'single_cv3.npz'), **out)


def gen_denoiser_normal():
    """matcha/hifigan/denoiser.py:20-21: Denoiser(mode="normal") — the probe mel is torch.randn((1, 80, 88)) from the global generator — on the tiny HiFi-GAN
    of matcha_tiny.npz -> denoiser_normal.npz (seed, bias spectrum, denoised audio)"""
    from matcha.hifigan.models import Generator
    from matcha.hifigan.denoiser import Denoiser
    from flowmirror_hydravox_amd.config import tiny_hifigan_config
    from oracle import matcha_ref
    hc = tiny_hifigan_config()
    h = DictConfig(resblock='1', upsample_rates=list(hc.upsample_rates), upsample_kernel_sizes=list(hc.upsample_kernel_sizes),
                   upsample_initial_channel=hc.initial_channel, resblock_kernel_sizes=list(hc.resblock_kernel_sizes),
                   resblock_dilation_sizes=[list(d) for d in hc.resblock_dilations])
    gen = Generator(h).eval()
    sdg = W.make_hifigan_state(hc, seed=23, init='fan_in')
    gen.load_state_dict(sdg)
    g = torch.Generator().manual_seed(77)
    mel = torch.randn(1, hc.mel, 30, generator=g)
    seed = 4242
    with torch.inference_mode():
        wav = gen(mel)
        torch.manual_seed(seed)
        den = Denoiser(gen, filter_length=hc.n_fft, n_overlap=hc.n_overlap, win_length=hc.n_fft, mode='normal')
        clean = den(wav.squeeze(1), strength=0.05)
    torch.manual_seed(seed)
    o_bias = matcha_ref.denoiser_bias(sdg, hc, mode='normal')
    o_clean = matcha_ref.denoise(wav.squeeze(1), o_bias, hc, 0.05)
    d = [(o_bias - den.bias_spec).abs().max().item(), (o_clean - clean).abs().max().item()]
    assert max(d) < 1e-4, d
    print('[denoiser-normal] oracle-reference max abs diff bias %.1e denoised %.1e (bias max %.3f)' % (d[0], d[1], den.bias_spec.max()))
    np.savez_compressed(os.path.join(HERE, 'denoiser_normal.npz'), weight_seed=np.int64(23), weight_sha=np.array(MG.state_checksum(sdg)), seed=np.int64(seed),
                        wav=wav.numpy(), bias=den.bias_spec.numpy(), clean=clean.numpy(), strength=np.float32(0.05))


if __name__ == '__main__':
    which = sys.argv[1:] or ['flow_est', 'flow_solve', 'hift', 'llm', 'single', 'denoiser_normal', 'flow_prompt']
    for w in which:
        t0 = time.time()
        {'flow_est': gen_flow_est, 'flow_solve': gen_flow_solve, 'hift': gen_hift, 'llm': gen_llm, 'single': gen_single, 'denoiser_normal': gen_denoiser_normal, 'flow_prompt': gen_flow_prompt, 'hift_cond': gen_hift_cond, 'single_cond': lambda: gen_hift_cond('single')}[w]()
        print('== %s done in %.0f s' % (w, time.time() - t0))
