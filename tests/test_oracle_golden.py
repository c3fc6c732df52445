"""The CPU oracle against the golden vectors minted from the reference (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, state_checksum
from flowmirror_hydravox_amd import weights as W
from oracle import sampler_ref, llm_ref, flow_ref, hift_ref


@pytest.mark.parametrize('fname', ['sampler_small.npz', 'sampler_big.npz'])
def test_sampler_ids_and_noise_consumption(fname):
    g = load_golden(fname)
    n = len(g['id'])
    for i in range(n):
        ns = sampler_ref.NoiseStream(seed=int(g['seed'][i]))
        hist = g['hist'][i, :g['hist_len'][i]].tolist()
        try:
            got = sampler_ref.sampling_ids(g['logp'][i], hist, ns, int(g['Vs'][i]), bool(g['ignore_eos'][i]),
                                           top_p=float(g['top_p'][i]), top_k=int(g['top_k'][i]),
                                           win_size=int(g['win'][i]), tau_r=float(g['tau'][i]))
        except RuntimeError:
            got = -1
        assert got == int(g['id'][i]), i
        assert ns.cursor == int(g['consumed'][i]), i


def test_noise_stream_is_chunk_invariant():
    a = sampler_ref.NoiseStream(seed=3, chunk=7)
    b = sampler_ref.NoiseStream(seed=3, chunk=4096)
    xs = np.concatenate([a.take(n) for n in (3, 10, 6761, 1, 25)])
    ys = b.take(len(xs))
    assert np.array_equal(xs, ys)


@pytest.fixture(scope='module')
def llm_golden(tiny_cfg):
    g = load_golden('llm_tiny.npz')
    sd = W.make_llm_state(tiny_cfg.llm, seed=int(g['weight_seed']), init='fan_in', with_lm_head=True)
    assert state_checksum(sd) == str(g['weight_sha']), 'synthetic weights differ from the ones the fixture was minted with'
    return g, sd


@pytest.mark.parametrize('use_cache', [False, True])
def test_llm_token_streams(tiny_cfg, llm_golden, use_cache):
    g, sd = llm_golden
    cfg = tiny_cfg.llm
    for r in range(int(g['n_runs'])):
        p = 'r%d_' % r
        top_p, top_k, win, tau = g[p + 'sampling']
        ns = sampler_ref.NoiseStream(seed=int(g[p + 'seed']))
        toks = list(llm_ref.llm_inference(
            sd, cfg, torch.from_numpy(g[p + 'text']), ns, prompt_text=torch.from_numpy(g[p + 'ptext']),
            prompt_speech_token=torch.from_numpy(g[p + 'pspeech']), inference_head_num=int(g[p + 'K']),
            sampling=dict(top_p=float(top_p), top_k=int(top_k), win_size=int(win), tau_r=float(tau)),
            max_token_text_ratio=float(g[p + 'ratios'][0]), min_token_text_ratio=float(g[p + 'ratios'][1]),
            use_kv_cache=use_cache))
        assert toks == g[p + 'tokens'].tolist(), r
        assert all(t < cfg.speech_tokens for t in toks)


def test_llm_accept_stress_token_streams(tiny_cfg):
    """BASELINE configs[2] "accept-rate stress": low-entropy checkpoint on which the reference's ras_sampling fell back to the full
    softmax on 28-70 % of its calls (tests/golden/make_golden.py::gen_llm_stress); the oracle emits the reference's ids and takes the
    fallback branch exactly as often."""
    g = load_golden('llm_stress_tiny.npz')
    cfg = tiny_cfg.llm
    sd = W.accept_stress_llm_state(W.make_llm_state(cfg, seed=int(g['weight_seed']), init='fan_in', with_lm_head=True))
    assert state_checksum(sd) == str(g['weight_sha'])
    top_p, top_k, win, tau = g['sampling']
    calls = [0, 0]
    orig = sampler_ref.ras_sample_once

    def counted(*a, **k):
        top, info = orig(*a, **k)
        calls[0] += 1
        calls[1] += int(info['fallback'])
        return top, info

    sampler_ref.ras_sample_once = counted
    try:
        for r in range(int(g['n_runs'])):
            p = 'r%d_' % r
            calls[:] = [0, 0]
            toks = list(llm_ref.llm_inference(sd, cfg, torch.from_numpy(g[p + 'text']), sampler_ref.NoiseStream(seed=int(g[p + 'seed'])),
                                              prompt_speech_token=torch.from_numpy(g[p + 'pspeech']), inference_head_num=int(g[p + 'K']),
                                              sampling=dict(top_p=float(top_p), top_k=int(top_k), win_size=int(win), tau_r=float(tau)),
                                              max_token_text_ratio=8, min_token_text_ratio=8, use_kv_cache=True))
            assert toks == g[p + 'tokens'].tolist(), r
            assert calls == [int(g[p + 'calls']), int(g[p + 'fallbacks'])], (r, calls)
            assert calls[1] / calls[0] > 0.25
    finally:
        sampler_ref.ras_sample_once = orig


def test_llm_first_step_numerics(tiny_cfg, llm_golden):
    g, sd = llm_golden
    cfg = tiny_cfg.llm
    for r in range(int(g['n_runs'])):
        p = 'r%d_' % r
        x = llm_ref.build_prefix(sd, cfg, torch.from_numpy(g[p + 'text']), torch.from_numpy(g[p + 'ptext']), torch.from_numpy(g[p + 'pspeech']))
        y = llm_ref.backbone(x, sd, cfg)
        assert np.allclose(y[-1].numpy(), g[p + 'y_last'], atol=2e-5)
        K = llm_ref.effective_heads(cfg, int(g[p + 'K']))
        lps = torch.stack(llm_ref.head_logps(torch.from_numpy(g[p + 'y_last']), sd, cfg, K))
        assert np.allclose(lps.numpy(), g[p + 'logps'], atol=5e-5)


def test_flow_pieces_and_end_to_end(tiny_cfg):
    g = load_golden('flow_tiny.npz')
    c = tiny_cfg.flow
    sd = W.make_flow_state(c, seed=int(g['weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['weight_sha'])
    assert np.array_equal(flow_ref.cfm_noise(c)[0, :2, :8].numpy(), g['noise_head'])
    for r in range(int(g['n_runs'])):
        p = 'r%d_' % r
        T = g[p + 'est_x'].shape[-1]
        pla = flow_ref.pre_lookahead(torch.from_numpy(g[p + 'h0']), sd, c)
        assert np.allclose(pla.numpy(), g[p + 'pla'], atol=1e-5)
        est = flow_ref.dit_forward(torch.from_numpy(g[p + 'est_x']), torch.ones(2, 1, T), torch.from_numpy(g[p + 'est_mu']),
                                   torch.from_numpy(g[p + 'est_t']), torch.from_numpy(g[p + 'est_spk']),
                                   torch.from_numpy(g[p + 'est_cond']), sd, c)
        assert np.allclose(est.numpy(), g[p + 'est_out'], atol=1e-4)
        ptoken = torch.from_numpy(g[p + 'ptoken'])
        has_p = ptoken.shape[1] > 0
        mel = flow_ref.flow_inference(torch.from_numpy(g[p + 'token']), torch.from_numpy(g[p + 'emb']), sd, c,
                                      prompt_token=ptoken if has_p else None,
                                      prompt_feat=torch.from_numpy(g[p + 'pfeat']) if has_p else None)
        assert mel.shape == g[p + 'mel'].shape
        assert np.allclose(mel.numpy(), g[p + 'mel'], atol=2e-4)


def test_cfm_schedule_and_cfg_combine():
    # t-grid and CFG arithmetic with a stub linear estimator (flow_matching.py:71-124, 225-227)
    ts = flow_ref.cosine_t_span(10)
    assert ts[0] == 0 and abs(float(ts[-1]) - 1.0) < 1e-6 and torch.all(ts[1:] > ts[:-1])

    def est(x, m, mu, t, s, c):
        return 0.5 * x + mu + t[:, None, None]
    x0 = torch.ones(1, 80, 4)
    mu = 2 * torch.ones(1, 80, 4)
    out, traj = flow_ref.solve_euler(x0, ts, mu, torch.ones(1, 1, 4), torch.zeros(1, 80), torch.zeros(1, 80, 4), est, 0.7)
    x = x0.clone()
    t = ts[0]
    for k in range(10):
        dt = ts[k + 1] - t
        d = 1.7 * (0.5 * x + mu + t) - 0.7 * (0.5 * x + 0 + t)
        x = x + dt * d
        t = t + dt
    assert torch.allclose(out, x, atol=1e-6)


def test_hift_stages(tiny_cfg):
    g = load_golden('hift_tiny.npz')
    c = tiny_cfg.hift
    sd = W.make_hift_state(c, seed=int(g['weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['weight_sha'])
    tables = hift_ref.make_tables(c, seed=int(g['table_seed']))
    for r in range(int(g['n_runs'])):
        p = 'r%d_' % r
        mel = torch.from_numpy(g[p + 'mel'])
        f0 = hift_ref.f0_predictor(mel, sd)
        assert np.allclose(f0.numpy(), g[p + 'f0'], atol=1e-3)
        wav_s = hift_ref.decode(mel, torch.from_numpy(g[p + 'source']), sd, c)          # decode on the reference's source
        assert np.allclose(wav_s.numpy(), g[p + 'wav'], atol=2e-4)
        wav, s = hift_ref.hift_inference(mel, sd, c, tables)
        assert wav.shape == (1, 480 * mel.shape[-1])
        assert np.allclose(s.numpy(), g[p + 'source'], atol=1e-3)
        # end to end the F0 -> phase accumulation amplifies fp32 rounding: looser bound
        assert np.abs(wav.numpy() - g[p + 'wav']).max() < 5e-3


# ------------------------------------------------------------------------------------------------------------------------
# Matcha-TTS family: oracle vs the vectors minted from the reference's own modules
# ------------------------------------------------------------------------------------------------------------------------
def test_matcha_decoder_cfm_hifigan_denoiser():
    from flowmirror_hydravox_amd.config import tiny_matcha_config, tiny_hifigan_config
    from oracle import matcha_ref
    g = load_golden('matcha_tiny.npz')
    c = tiny_matcha_config()
    sd = W.make_matcha_state(c, seed=int(g['m_weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['m_weight_sha'])
    for r in range(2):
        p = 'm%d_' % r
        x, mu, spks, t = (torch.from_numpy(g[p + k]) for k in ('x', 'mu', 'spks', 't'))
        mask = torch.ones(1, 1, x.shape[-1])
        y = matcha_ref.decoder_forward(sd, c, x, mask, mu, t, spks)
        assert (y - torch.from_numpy(g[p + 'y'])).abs().max() < 1e-5
        s = matcha_ref.solve_euler(sd, c, torch.from_numpy(g[p + 'noise']) * c.temperature, mask, mu, c.n_timesteps, spks)
        assert (s - torch.from_numpy(g[p + 'sample'])).abs().max() < 1e-4
    cc = tiny_matcha_config(cv=True)
    sdc = W.make_matcha_state(cc, seed=int(g['c_weight_seed']), init='fan_in')
    assert state_checksum(sdc) == str(g['c_weight_sha'])
    y = matcha_ref.decoder_forward(sdc, cc, *(torch.from_numpy(g['c_' + k]) for k in ('x', 'mask', 'mu', 't', 'spks', 'cond')))
    assert (y - torch.from_numpy(g['c_y'])).abs().max() < 1e-5
    hc = tiny_hifigan_config()
    sdg = W.make_hifigan_state(hc, seed=int(g['g_weight_seed']), init='fan_in')
    assert state_checksum(sdg) == str(g['g_weight_sha'])
    wav = matcha_ref.generator_forward(sdg, hc, torch.from_numpy(g['g_mel']))
    assert (wav - torch.from_numpy(g['g_wav'])).abs().max() < 1e-5
    bias = matcha_ref.denoiser_bias(sdg, hc)
    assert (bias - torch.from_numpy(g['g_bias'])).abs().max() < 1e-5
    clean = matcha_ref.denoise(torch.from_numpy(g['g_wav']).squeeze(1), bias, hc, float(g['g_strength']))
    assert (clean - torch.from_numpy(g['g_clean'])).abs().max() < 1e-5


def test_matcha_euler_schedule_matches_the_reference_loop():
    from flowmirror_hydravox_amd.matcha import matcha_euler_schedule
    from oracle import matcha_ref
    for n in (1, 2, 10, 32):
        assert matcha_euler_schedule(n) == matcha_ref.euler_schedule(n)
        ts, dts = matcha_euler_schedule(n)
        assert len(ts) == n and abs(ts[-1] + dts[-1] - 1.0) < 1e-6


def test_convtranspose_phases_and_stft_bases_against_torch():
    from flowmirror_hydravox_amd.packing import convtranspose_phases, stft_bases
    import torch.nn.functional as F
    gq = torch.Generator().manual_seed(5)
    for (cin, cout, k, s) in [(6, 4, 4, 2), (5, 3, 16, 8), (3, 7, 8, 4)]:
        pd = (k - s) // 2
        w = torch.randn(cin, cout, k, generator=gq)
        x = torch.randn(1, cin, 9, generator=gq)
        ref = F.conv_transpose1d(x, w, stride=s, padding=pd)[0]             # (cout, 9*s)
        ph = convtranspose_phases(w, s, pd).view(s, cout, k // s, -1)[..., :cin]
        taps = k // s
        for p in range(s):
            c_p = (p + pd) // s
            xp = F.pad(x[0], (taps - 1 - c_p, c_p + taps))                 # input index t + c_p - (taps-1) + tau
            y = torch.stack([sum(ph[p, :, tau] @ xp[:, t + tau] for tau in range(taps)) for t in range(9)], 1)
            assert (y - ref[:, p::s]).abs().max() < 1e-5
    ana, syn, wsq = stft_bases(64)
    x = torch.randn(64, generator=gq)
    spec = torch.fft.rfft(x * torch.hann_window(64))
    mine = ana @ x
    assert (mine[:33] - spec.real).abs().max() < 1e-4 and (mine[33:] - spec.imag).abs().max() < 1e-4
    back = syn[:, :66] @ mine
    assert (back - torch.fft.irfft(spec, 64) * torch.hann_window(64)).abs().max() < 1e-4


def test_mel_spectrogram_oracle_vs_reference_vectors():
    from flowmirror_hydravox_amd.packing import mel_filterbank
    from oracle import matcha_ref
    g = load_golden('matcha_tiny.npz')
    for tag, kw in (('cv3', dict(n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=8000)),
                    ('toy', dict(n_fft=64, num_mels=8, sampling_rate=800, hop_size=32, win_size=64, fmin=0, fmax=400))):
        mb = mel_filterbank(kw['sampling_rate'], kw['n_fft'], kw['num_mels'], kw['fmin'], kw['fmax'])
        # shape and partition-of-energy sanity of the restated librosa table: every filter has support, Slaney area normalisation
        assert mb.shape == (kw['num_mels'], kw['n_fft'] // 2 + 1) and bool((mb.sum(1) > 0).all()) and float(mb.min()) >= 0.0
        out = matcha_ref.mel_spectrogram(torch.from_numpy(g['mel_%s_y' % tag]), mel_basis=mb, **kw)
        assert (out - torch.from_numpy(g['mel_%s_out' % tag])).abs().max() < 1e-5


def test_slaney_mel_table_is_pinned_and_the_product_table_equals_it():
    """N2 / matcha/utils/audio.py:53, cosyvoice/cli/frontend.py:95 (whisper's mel_filters.npz = librosa.filters.mel(sr=16000, n_fft=400, n_mels=80 | 128)):
    librosa is absent, so (1) the oracle's scalar float64 statement of the Slaney table is held to the LITERAL values librosa prints in its docstrings
    (hz_to_mel, mel_to_hz, mel_frequencies(n_mels=40), filters.mel(sr=22050, n_fft=2048)[0, 1]) at their printed precision, and to the table's defining
    properties; (2) the product's vectorised packing.mel_filterbank — separate code — must equal it to 1e-7 of the table's scale at every geometry the path uses."""
    from flowmirror_hydravox_amd.packing import mel_filterbank
    from oracle import frontend_ref as R
    pub = R.LIBROSA_PUBLISHED
    for f, m in pub['hz_to_mel'].items():
        assert abs(R.slaney_hz_to_mel(f) - m) < 5e-3, (f, R.slaney_hz_to_mel(f))                      # printed to 2 decimals
    for m, f in pub['mel_to_hz'].items():
        assert abs(R.slaney_mel_to_hz(m) - f) < 5e-4, (m, R.slaney_mel_to_hz(m))                      # printed to 3 decimals
    mf = R.slaney_mel_frequencies(40, 0.0, 11025.0)
    assert len(mf) == 40 and max(abs(a - b) for a, b in zip(mf, pub['mel_frequencies_40'])) < 6e-4      # printed to 3 decimals
    t = R.slaney_mel_table(22050, 2048, 128)
    assert t.shape == (128, 1025) and abs(float(t[0, 1]) - pub['filters_mel_22050_2048_row0_col1']) < 5e-4 and float(t[0, 0]) == 0.0
    for (sr, n_fft, n_mels, fmin, fmax) in ((24000, 1920, 80, 0.0, None), (24000, 1920, 80, 0.0, 8000.0), (22050, 1024, 80, 0.0, 8000.0), (22050, 2048, 128, 0.0, None),
                                            (16000, 400, 80, 0.0, None), (16000, 400, 128, 0.0, None), (800, 64, 8, 0.0, 400.0)):
        ref = R.slaney_mel_table(sr, n_fft, n_mels, fmin, fmax)
        # defining properties: non-negative, zero outside (edge[i], edge[i+2]), and Slaney's normalisation: the continuous triangle has unit area, so the sampled
        # one sums to ~ 1 / bin width wherever a filter spans several bins
        edges = R.slaney_mel_frequencies(n_mels + 2, fmin, sr / 2.0 if fmax is None else fmax)
        freqs = torch.arange(n_fft // 2 + 1, dtype=torch.float64) * sr / n_fft
        assert float(ref.min()) >= 0.0
        for i in (0, n_mels // 2, n_mels - 1):
            outside = (freqs <= edges[i]) | (freqs >= edges[i + 2])
            assert float(ref[i][outside].abs().max()) == 0.0
            if edges[i + 2] - edges[i] > 8 * sr / n_fft:
                assert abs(float(ref[i].sum()) * sr / n_fft - 1.0) < 0.05, (sr, n_fft, i, float(ref[i].sum()) * sr / n_fft)
        mine = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
        assert mine.dtype == torch.float32 and mine.shape == ref.shape
        d = float((mine.double() - ref).abs().max() / ref.max())
        assert d < 1e-7, (sr, n_fft, n_mels, fmin, fmax, d)
        # (3) an independent implementation of the same table, where this image has one: transformers.audio_utils.mel_filter_bank(norm / mel_scale "slaney")
        try:
            from transformers.audio_utils import mel_filter_bank
        except Exception:
            continue
        ind = mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin, max_frequency=sr / 2.0 if fmax is None else fmax,
                              sampling_rate=sr, norm='slaney', mel_scale='slaney').T
        assert np.abs(ind - ref.numpy()).max() / float(ref.max()) < 1e-9, (sr, n_fft, n_mels)


# ------------------------------------------------------------------------------------------------------------------------
# streaming synthesis: static chunk mask, finalize=False in flow and HiFT
# ------------------------------------------------------------------------------------------------------------------------
def test_streaming_oracle_vs_reference_vectors(tiny_cfg):
    import dataclasses
    g = load_golden('stream_tiny.npz')
    c = dataclasses.replace(tiny_cfg.flow, static_chunk_size=int(g['chunk']))
    sd = W.make_flow_state(c, seed=int(g['flow_weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['flow_weight_sha'])
    t = lambda k: torch.from_numpy(g[k])
    mask = t('est_mask')
    est = flow_ref.dit_forward(t('est_x'), mask, t('est_mu'), t('est_t'), t('est_spk'), t('est_cond'), sd, c, streaming=True)
    assert np.allclose((est * mask).numpy(), g['est_out'] * g['est_mask'], atol=1e-4)
    full = flow_ref.dit_forward(t('est_x'), mask, t('est_mu'), t('est_t'), t('est_spk'), t('est_cond'), sd, c, streaming=False)
    assert np.abs((full * mask).numpy() - g['est_out'] * g['est_mask']).max() > 1e-2          # the chunk mask is not a no-op here
    # chunked flow: every non-final chunk of the reference equals the oracle's, and is a prefix of the whole streaming pass
    token, hop = t('token'), int(g['hop'])
    kw = dict(prompt_token=t('ptoken'), prompt_feat=t('pfeat'), streaming=True)
    whole = flow_ref.flow_inference(token, t('emb'), sd, c, **kw)
    assert np.allclose(whole.numpy(), g['mel_whole'], atol=2e-4)
    for k in range(int(g['n_chunks'])):
        fin = bool(g['mel_chunk%d_final' % k])
        part = flow_ref.flow_inference(token[:, :k * hop + hop + 3], t('emb'), sd, c, finalize=fin, **kw)
        assert part.shape == g['mel_chunk%d' % k].shape == (1, 80, 2 * (k * hop + hop) if not fin else 2 * token.shape[1])
        assert np.allclose(part.numpy(), g['mel_chunk%d' % k], atol=2e-4)
        assert np.allclose(part.numpy(), g['mel_whole'][:, :, :part.shape[2]], atol=2e-4)      # the check of flow.py:436-459
    # HiFT finalize=False
    hc = tiny_cfg.hift
    sdh = W.make_hift_state(hc, seed=int(g['hift_weight_seed']), init='fan_in')
    assert state_checksum(sdh) == str(g['hift_weight_sha'])
    tables = hift_ref.make_tables(hc, seed=int(g['hift_table_seed']))
    mel = t('h_mel')
    for k in range(int(g['h_runs'])):
        n = int(g['h%d_n' % k])
        wav, s = hift_ref.hift_inference(mel[:, :, :n], sdh, hc, tables, finalize=False)
        assert wav.shape == (1, 480 * (n - 8)) and s.shape == (1, 1, 480 * (n - 3))
        assert np.allclose(s.numpy(), g['h%d_source' % k], atol=1e-3)
        assert np.abs(wav.numpy() - g['h%d_wav' % k]).max() < 5e-3
        assert np.abs(g['h%d_wav' % k] - g['h_wav_whole'][:, :wav.shape[1]]).max() < 1e-3      # the check of generator.py:739-747


# ------------------------------------------------------------------------------------------------------------------------------
# HydraVox-CV3 WIDTHS (config.cv3w_config: every per-layer shape of the benchmarked model, 2 layers / 2 DiT blocks)
# ------------------------------------------------------------------------------------------------------------------------------
def cv3w_flow_inputs(seed, T, lens):
    """the seeded estimator inputs of tests/golden/make_golden.py::cv3w_flow_inputs (fixtures hold their checksum and the reference's output)"""
    g = torch.Generator()
    g.manual_seed(seed)
    x, mu, cond = (torch.randn(2, 80, T, generator=g) for _ in range(3))
    spk = torch.randn(2, 80, generator=g)
    mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).float()[:, None, :]
    return x, mask, mu, spk, cond


@pytest.fixture(scope='module')
def cv3w_cfg():
    from flowmirror_hydravox_amd.config import cv3w_config
    return cv3w_config()


def test_cv3w_llm_oracle_vs_reference_vectors(cv3w_cfg):
    g = load_golden('llm_cv3w.npz')
    cfg = cv3w_cfg.llm
    sd = W.make_llm_state(cfg, seed=int(g['weight_seed']), init='fan_in', with_lm_head=True)
    assert state_checksum(sd) == str(g['weight_sha'])
    top_p, top_k, win, tau = g['sampling']
    sampling = dict(top_p=float(top_p), top_k=int(top_k), win_size=int(win), tau_r=float(tau))
    for r in range(int(g['n_runs'])):
        p = 'r%d_' % r
        toks = list(llm_ref.llm_inference(sd, cfg, torch.from_numpy(g[p + 'text']), sampler_ref.NoiseStream(seed=int(g[p + 'seed'])),
                                          prompt_text=torch.from_numpy(g[p + 'ptext']), prompt_speech_token=torch.from_numpy(g[p + 'pspeech']),
                                          inference_head_num=int(g[p + 'K']), sampling=sampling, max_token_text_ratio=float(g[p + 'ratios'][0]),
                                          min_token_text_ratio=float(g[p + 'ratios'][1]), use_kv_cache=True))
        assert toks == g[p + 'tokens'].tolist(), r
        if p + 'y_last' in g:
            x = llm_ref.build_prefix(sd, cfg, torch.from_numpy(g[p + 'text']), torch.from_numpy(g[p + 'ptext']), torch.from_numpy(g[p + 'pspeech']))
            y = llm_ref.backbone(x, sd, cfg)
            assert np.abs(y[-1].numpy() - g[p + 'y_last']).max() < 5e-5
            lp = torch.stack(llm_ref.head_logps(y[-1], sd, cfg, cfg.head_num)).numpy()
            assert np.abs(lp - g[p + 'logps']).max() < 5e-4
            # the bf16-faithful mode is the same function with rounded operands: close to, and different from, the fp32 result
            yb = llm_ref.backbone(llm_ref.bf16r(x), sd, cfg, emu=True)
            rel = (yb[-1] - y[-1]).abs().max().item() / y[-1].abs().max().item()
            assert 1e-4 < rel < 3e-2, rel


def test_cv3w_bf16_rounding_floor(cv3w_cfg):
    """How far can two FAITHFUL bf16 evaluations drift apart?  The bf16-faithful oracle is run twice on the reference's own token streams:
    once as is, once with every GEMM's columns permuted (emu='perm': identical bf16 operands, another fp32 summation order).  An fp32
    intermediate that lands on the other side of a bf16 rounding boundary changes one operand element by one bf16 ulp; through 2 layers and
    the 22016-wide MTP heads this yields ~5e-3 of the hidden scale, ~5e-2 in the log-probs, and sampling decisions that agree on ~93 % of
    the draws — the floor the bf16 HIP path is held to in tests/test_gpu_cv3w.py (it cannot be asked to beat it)."""
    g = load_golden('llm_cv3w.npz')
    c = cv3w_cfg.llm
    sd = W.make_llm_state(c, seed=int(g['weight_seed']), init='fan_in', with_lm_head=True)
    top_p, top_k, win, tau = g['sampling']
    sampling = dict(top_p=float(top_p), top_k=int(top_k), win_size=int(win), tau_r=float(tau))
    draws = same = 0
    worst = [0.0, 0.0]
    for r in (0, 11):
        p = 'r%d_' % r
        K = int(g[p + 'K'])
        text, ptext, ps = (torch.from_numpy(g[p + k]) for k in ('text', 'ptext', 'pspeech'))
        maxr, minr = float(g[p + 'ratios'][0]), float(g[p + 'ratios'][1])
        trace = []
        ora = list(llm_ref.llm_inference(sd, c, text, sampler_ref.NoiseStream(seed=int(g[p + 'seed'])), prompt_text=ptext, prompt_speech_token=ps,
                                         inference_head_num=K, sampling=sampling, max_token_text_ratio=maxr, min_token_text_ratio=minr,
                                         use_kv_cache=True, emu=True, trace=trace))
        hist, min_len = [], int(len(text) * minr)
        for st in trace:
            def draw(logps):
                ns = sampler_ref.NoiseStream(seed=int(g[p + 'seed']))
                ns.cursor = st['cursor']
                return sampler_ref.sample_step(logps, hist, ns, c.speech_tokens, min_len, sampling)
            want = draw([lp.numpy() for lp in st['logps']])
            x = llm_ref.build_prefix(sd, c, text, ptext, torch.cat([ps, torch.tensor(hist, dtype=torch.int32)]), emu=True)
            y = llm_ref.backbone(x, sd, c, emu='perm')[-1]
            lps = llm_ref.head_logps(y, sd, c, K, emu='perm')
            got = draw([lp.numpy() for lp in lps])
            worst[0] = max(worst[0], ((y - st['y_last']).abs().max() / st['y_last'].abs().max()).item())
            worst[1] = max(worst[1], max((a - b).abs().max().item() for a, b in zip(lps, st['logps'])))
            same += sum(int(a == b) for a, b in zip(got, want))
            draws += len(want)
            hist += [t for t in want if t < c.speech_tokens][:max(0, len(ora) - len(hist))]
        assert hist == ora
    print('two faithful bf16 evaluations: %d / %d draws identical, hidden rel %.1e, logp abs %.1e' % (same, draws, worst[0], worst[1]))
    assert 1e-4 < worst[0] < 2e-2 and 1e-3 < worst[1] < 0.15         # a real floor, of the size the GPU bounds assume
    assert 0.8 * draws <= same < draws                                # decisions do flip at that floor


def test_cv3w_flow_oracle_vs_reference_vectors(cv3w_cfg):
    g = load_golden('flow_cv3w.npz')
    c = cv3w_cfg.flow
    sd = W.make_flow_state(c, seed=int(g['weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['weight_sha'])
    tag = 'e1'                                                                  # (e0, T = 2176, is checked on the GPU box; here the short one)
    x, mask, mu, spk, cond = cv3w_flow_inputs(int(g[tag + '_seed']), int(g[tag + '_T']), g[tag + '_lens'].tolist())
    assert state_checksum(dict(x=x, mu=mu, spk=spk, cond=cond)) == str(g[tag + '_in_sha'])
    est = flow_ref.dit_forward(x, mask, mu, torch.from_numpy(g[tag + '_t']), spk, cond, sd, c, streaming=bool(g[tag + '_streaming']))
    assert np.abs((est * mask).numpy() - g[tag + '_out']).max() < 2e-4
    emu = flow_ref.dit_forward(x, mask, mu, torch.from_numpy(g[tag + '_t']), spk, cond, sd, c, streaming=bool(g[tag + '_streaming']), emu=True)
    rel = np.abs((emu * mask).numpy() - g[tag + '_out']).max() / np.abs(g[tag + '_out']).max()
    assert 1e-4 < rel < 5e-2, rel
    # the product's bf16 mode additionally keeps the residual stream in fp16 (one rounding per residual add, as the reference's .half() run does)
    emu16 = flow_ref.dit_forward(x, mask, mu, torch.from_numpy(g[tag + '_t']), spk, cond, sd, c, streaming=bool(g[tag + '_streaming']), emu=True, resid16=True)
    rel16 = np.abs((emu16 * mask).numpy() - g[tag + '_out']).max() / np.abs(g[tag + '_out']).max()
    d16 = np.abs((emu16 - emu).numpy()).max() / np.abs(g[tag + '_out']).max()
    assert 0 < d16 < 2e-2 and rel16 < 5e-2, (rel16, d16)
    # the reference's own fp16 run of this case (flow_half.npz: `flow.half()`, its deployed dtype) sits between fp32 and the bf16 emulation
    h = load_golden('flow_half.npz')
    assert str(h['w_in_sha']) == str(g[tag + '_in_sha']) and str(h['w_weight_sha']) == str(g['weight_sha'])
    assert np.abs(h['w_out_f32'] - g[tag + '_out']).max() < 1e-5
    half = np.abs(h['w_out_f16'] - g[tag + '_out']).max() / np.abs(g[tag + '_out']).max()
    assert abs(half - float(h['w_half_vs_f32'])) < 1e-6 and 1e-4 < half < rel16, (half, rel16)
    pla = flow_ref.pre_lookahead(torch.from_numpy(g['h0']), sd, c)
    assert np.abs(pla.numpy() - g['pla']).max() < 1e-4
    mel = flow_ref.flow_inference(torch.from_numpy(g['token']), torch.from_numpy(g['emb']), sd, c, prompt_token=torch.from_numpy(g['ptoken']),
                                  prompt_feat=torch.from_numpy(g['pfeat']))
    assert np.abs(mel.numpy() - g['mel']).max() < 5e-4


def test_cv3w_hift_oracle_vs_reference_vectors(cv3w_cfg):
    g = load_golden('hift_cv3w.npz')
    c = cv3w_cfg.hift
    sd = W.make_hift_state(c, seed=int(g['weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['weight_sha'])
    for r in range(int(g['n_runs'])):
        p = 'r%d_' % r
        mel = torch.from_numpy(g[p + 'mel'])
        assert np.abs(hift_ref.f0_predictor(mel, sd).numpy() - g[p + 'f0']).max() < 2e-3
        wav = hift_ref.decode(mel, torch.from_numpy(g[p + 'source']), sd, c)
        assert np.abs(wav.numpy() - g[p + 'wav']).max() < 5e-4


# ---- round 3: 1000 more sampler cases, and the models at FULL DEPTH (24 LM layers / 22 DiT blocks), all minted from the reference -----------
def test_sampler_thousand_cases_ids_and_noise_consumption():
    """tests/golden/sampler_many.npz: inputs regenerated from per-case seeds (tests/golden/sampler_cases.py, checksummed), ids and noise
    consumption == what the reference's sampling_ids returned for them (760 cases at V = 296, 240 at V = 6761)."""
    import hashlib
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from sampler_cases import make_case, N_CASES
    g = load_golden('sampler_many.npz')
    assert len(g['id']) == N_CASES >= 1000
    n_big = 0
    for i in range(N_CASES):
        c = make_case(i)
        assert int.from_bytes(hashlib.sha256(c['logp'].tobytes()).digest()[:8], 'little', signed=True) == int(g['logp_sha'][i]), i
        n_big += c['Vs'] == 6561
        ns = sampler_ref.NoiseStream(seed=c['seed'])
        try:
            got = sampler_ref.sampling_ids(c['logp'], list(c['hist']), ns, c['Vs'], c['ignore_eos'], top_p=c['top_p'], top_k=c['top_k'],
                                           win_size=c['win'], tau_r=c['tau'])
        except RuntimeError:
            got = -1
        assert got == int(g['id'][i]), i
        assert ns.cursor == int(g['consumed'][i]), i
    assert n_big >= 200


def test_cv3d_flow_oracle_vs_reference_vectors():
    """22 DiT blocks (config.cv3d_config): the chunk-masked estimator at T = 192 and the whole 10-step solve == the reference's outputs"""
    from flowmirror_hydravox_amd.config import cv3d_config
    g = load_golden('flow_cv3d.npz')
    c = cv3d_config().flow
    assert c.depth == 22
    sd = W.make_flow_state(c, seed=int(g['weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['weight_sha'])
    tag = 'e1'
    x, mask, mu, spk, cond = cv3w_flow_inputs(int(g[tag + '_seed']), int(g[tag + '_T']), g[tag + '_lens'].tolist())
    assert state_checksum(dict(x=x, mu=mu, spk=spk, cond=cond)) == str(g[tag + '_in_sha'])
    est = flow_ref.dit_forward(x, mask, mu, torch.from_numpy(g[tag + '_t']), spk, cond, sd, c, streaming=bool(g[tag + '_streaming']))
    assert np.abs((est * mask).numpy() - g[tag + '_out']).max() < 1e-3
    mel = flow_ref.flow_inference(torch.from_numpy(g['token']), torch.from_numpy(g['emb']), sd, c, prompt_token=torch.from_numpy(g['ptoken']),
                                  prompt_feat=torch.from_numpy(g['pfeat']))
    assert np.abs(mel.numpy() - g['mel']).max() < 2e-3


def test_cv3d_llm_oracle_vs_reference_vectors():
    """24 LM layers: the K = 1 and the first K = 2 / K = 4 token streams and the first-step pin of run 1 == the reference's (the remaining runs,
    incl. the one across context 1024, are replayed on the GPU box only: the CPU oracle needs minutes for them)"""
    from flowmirror_hydravox_amd.config import cv3d_config
    g = load_golden('llm_cv3d.npz')
    cfg = cv3d_config().llm
    assert cfg.layers == 24
    sd = W.make_llm_state(cfg, seed=int(g['weight_seed']), init='fan_in', with_lm_head=True)
    assert state_checksum(sd) == str(g['weight_sha'])
    top_p, top_k, win, tau = g['sampling']
    sampling = dict(top_p=float(top_p), top_k=int(top_k), win_size=int(win), tau_r=float(tau))
    for r in (0, 2, 4):
        p = 'r%d_' % r
        toks = list(llm_ref.llm_inference(sd, cfg, torch.from_numpy(g[p + 'text']), sampler_ref.NoiseStream(seed=int(g[p + 'seed'])),
                                          prompt_text=torch.from_numpy(g[p + 'ptext']), prompt_speech_token=torch.from_numpy(g[p + 'pspeech']),
                                          inference_head_num=int(g[p + 'K']), sampling=sampling, max_token_text_ratio=float(g[p + 'ratios'][0]),
                                          min_token_text_ratio=float(g[p + 'ratios'][1]), use_kv_cache=True))
        assert toks == g[p + 'tokens'].tolist(), r
    p = 'r1_'
    x = llm_ref.build_prefix(sd, cfg, torch.from_numpy(g[p + 'text']), torch.from_numpy(g[p + 'ptext']), torch.from_numpy(g[p + 'pspeech']))
    y = llm_ref.backbone(x, sd, cfg)
    assert np.abs(y[-1].numpy() - g[p + 'y_last']).max() < 2e-4
    lp = torch.stack(llm_ref.head_logps(y[-1], sd, cfg, cfg.head_num)).numpy()
    assert np.abs(lp - g[p + 'logps']).max() < 2e-3


def test_reference_deployed_dtype_fixtures_are_consistent():
    """llm_bf16.npz / flow_half.npz hold the reference run in the dtypes it deploys (llm bf16, flow fp16; make_golden.py: gen_llm_bf16, gen_flow_half)
    beside its fp32 run of the SAME case as the full-depth fixtures: same weights, same fp32 outputs, and the recorded distances are those of the arrays."""
    lb, ld = load_golden('llm_bf16.npz'), load_golden('llm_cv3d.npz')
    assert str(lb['weight_sha']) == str(ld['weight_sha'])
    assert np.abs(lb['y_last_f32'] - ld['r1_y_last']).max() < 1e-5 and np.abs(lb['logps_f32'] - ld['r1_logps']).max() < 1e-4
    dy = np.abs(lb['y_last_bf16'] - lb['y_last_f32']).max() / np.abs(lb['y_last_f32']).max()
    dl = np.abs(lb['logps_bf16'] - lb['logps_f32']).max()
    assert abs(dy - float(lb['hidden_bf16_vs_f32'])) < 1e-6 and abs(dl - float(lb['logp_bf16_vs_f32'])) < 1e-5
    assert 5e-3 < dy < 5e-2 and 2e-2 < dl < 0.5            # bf16 through 24 layers: percent-level, as the reference itself runs
    fh, fd = load_golden('flow_half.npz'), load_golden('flow_cv3d.npz')
    assert str(fh['d_in_sha']) == str(fd['e1_in_sha']) and np.abs(fh['d_out_f32'] - fd['e1_out']).max() < 1e-5
    dh = np.abs(fh['d_out_f16'] - fh['d_out_f32']).max() / np.abs(fh['d_out_f32']).max()
    assert abs(dh - float(fh['d_half_vs_f32'])) < 1e-6 and 2e-4 < dh < 5e-3


def test_onnx_container_round_trip_and_oracle_operators_vs_torch():
    """(1) the ONNX container reader / writer of the product (protobuf wire format, no `onnx` package) round-trips the two synthetic
    frontend graphs — node list, attributes of every type, initializers bit for bit; (2) the numpy oracle of the operator set (oracle/onnx_ref.py: published
    ONNX semantics; onnxruntime and the real graphs are absent -> parity unpinned) agrees with torch's CPU functional forms of the same operators where
    torch has one (Conv 1-D / 2-D with stride, dilation, padding and groups; AveragePool with ceil_mode; LayerNorm; BatchNorm; Softmax; GELU)."""
    import sys
    import os
    import torch.nn.functional as F
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import onnx_synth
    from flowmirror_hydravox_amd import onnx_graph as og
    from oracle import onnx_ref
    for mk in (onnx_synth.campplus_like, onnx_synth.tokenizer_like):
        g = mk()
        g2 = og.load_onnx(og.save_onnx(g))
        assert g2.inputs == g.inputs and g2.outputs == g.outputs and g2.opset == g.opset and len(g2.nodes) == len(g.nodes)
        assert set(g2.initializers) == set(g.initializers)
        for k, a in g.initializers.items():
            assert a.dtype == g2.initializers[k].dtype and np.array_equal(a, g2.initializers[k])
        for a, b in zip(g.nodes, g2.nodes):
            assert (a.op, a.inputs, a.outputs, sorted(a.attrs)) == (b.op, b.inputs, b.outputs, sorted(b.attrs))
            for k, val in a.attrs.items():
                assert (abs(val - b.attrs[k]) < 1e-7) if isinstance(val, float) else (val == b.attrs[k]), (a.op, k)
    N = og.Node
    rng = np.random.default_rng(3)

    def one(op, ins, **attrs):
        return onnx_ref._node(N(op, ['i%d' % i for i in range(len(ins))], ['o'], attrs), ins, 17)

    x = rng.standard_normal((2, 6, 37)).astype(np.float32)
    w = rng.standard_normal((8, 3, 5)).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    got = one('Conv', [x, w, b], kernel_shape=[5], strides=[2], dilations=[3], pads=[4, 7], group=2)
    want = F.conv1d(F.pad(torch.from_numpy(x), (4, 7)), torch.from_numpy(w), torch.from_numpy(b), stride=2, dilation=3, groups=2).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    x4 = rng.standard_normal((1, 3, 11, 9)).astype(np.float32)
    w4 = rng.standard_normal((4, 3, 3, 3)).astype(np.float32)
    got = one('Conv', [x4, w4], kernel_shape=[3, 3], strides=[2, 1], pads=[1, 1, 1, 1])
    np.testing.assert_allclose(got, F.conv2d(torch.from_numpy(x4), torch.from_numpy(w4), stride=(2, 1), padding=1).numpy(), rtol=1e-5, atol=1e-5)
    for T, k, s, p, cm, cip in ((37, 10, 10, 0, 1, 0), (40, 10, 10, 0, 1, 0), (23, 4, 3, 1, 1, 1), (23, 4, 3, 1, 0, 0), (9, 5, 5, 2, 1, 0)):
        xa = rng.standard_normal((2, 3, T)).astype(np.float32)
        got = one('AveragePool', [xa], kernel_shape=[k], strides=[s], pads=[p, p], ceil_mode=cm, count_include_pad=cip)
        want = F.avg_pool1d(torch.from_numpy(xa), k, s, p, ceil_mode=bool(cm), count_include_pad=bool(cip)).numpy()
        assert got.shape == want.shape, (T, k, s, p, cm, got.shape, want.shape)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    xs = rng.standard_normal((2, 5, 16)).astype(np.float32)
    g_, b_ = rng.standard_normal(16).astype(np.float32), rng.standard_normal(16).astype(np.float32)
    np.testing.assert_allclose(one('LayerNormalization', [xs, g_, b_], axis=-1, epsilon=1e-5),
                               F.layer_norm(torch.from_numpy(xs), (16,), torch.from_numpy(g_), torch.from_numpy(b_)).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(one('Softmax', [xs], axis=1), torch.softmax(torch.from_numpy(xs), 1).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(one('Gelu', [xs]), F.gelu(torch.from_numpy(xs)).numpy(), rtol=1e-5, atol=1e-6)
    p = [rng.standard_normal(5).astype(np.float32) for _ in range(3)] + [(0.5 + rng.random(5)).astype(np.float32)]
    np.testing.assert_allclose(one('BatchNormalization', [xs] + p, epsilon=1e-5),
                               F.batch_norm(torch.from_numpy(xs), torch.from_numpy(p[2]), torch.from_numpy(p[3]), torch.from_numpy(p[0]), torch.from_numpy(p[1]), False, 0.0, 1e-5).numpy(),
                               rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(one('Gemm', [xs[0], g_.reshape(1, 16).repeat(3, 0), b_[:3]], transB=1, alpha=0.5, beta=2.0),
                               0.5 * xs[0] @ g_.reshape(1, 16).repeat(3, 0).T + 2.0 * b_[:3], rtol=1e-5, atol=1e-5)
    assert np.array_equal(one('Round', [np.asarray([0.5, 1.5, 2.5, -0.5, -1.5], np.float32)]), np.asarray([0, 2, 2, -0, -2], np.float32))
    assert np.array_equal(one('Slice', [np.arange(10), np.asarray([8]), np.asarray([-11]), np.asarray([0]), np.asarray([-3])]), np.asarray([8, 5, 2]))


# ------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[0] and the benchmarked shapes (tests/golden/make_golden_fullsize.py)
# ------------------------------------------------------------------------------------------------------------------------------
def test_configs0_single_utterance_oracle_vs_reference_end_to_end():
    """BASELINE configs[0] on the CPU: head_num 1, ONE 64-char utterance at full CV3 depth — the oracle's three stages back to back (KV-cached LM:
    the reference's own uncached loop took minutes when the fixture was minted and was asserted equal there) against the reference's ids, mel, f0,
    source and waveform.  This is the configuration whose "CPU path" the north star names as the parity target."""
    import torch.nn.functional as F
    from flowmirror_hydravox_amd.config import cv3_config
    g = load_golden('single_cv3.npz')
    cfg = cv3_config()
    top_p, top_k, win, tau = g['sampling']
    sampling = dict(top_p=float(top_p), top_k=int(top_k), win_size=int(win), tau_r=float(tau))
    sd = W.make_llm_state(cfg.llm, seed=1986, init='fan_in', with_lm_head=True)
    assert state_checksum(sd) == str(g['llm_sha'])
    text = torch.from_numpy(g['text'])
    toks = list(llm_ref.llm_inference(sd, cfg.llm, text, sampler_ref.NoiseStream(seed=int(g['seed'])), inference_head_num=1, sampling=sampling,
                                      max_token_text_ratio=5.5, min_token_text_ratio=5.5, use_kv_cache=True))
    assert toks == g['tokens'].tolist() and len(toks) == 352
    del sd
    fsd = W.make_flow_state(cfg.flow, seed=1987, init='fan_in')
    assert state_checksum(fsd) == str(g['flow_sha'])
    mel = flow_ref.flow_inference(torch.tensor(toks)[None], torch.from_numpy(g['emb']), fsd, cfg.flow)
    d_mel = (mel - torch.from_numpy(g['mel'])).abs().max().item() / float(np.abs(g['mel']).max())
    assert tuple(mel.shape) == (1, 80, 704) and d_mel < 1e-4, d_mel
    del fsd
    hsd = W.make_hift_state(cfg.hift, seed=1988, init='fan_in')
    assert state_checksum(hsd) == str(g['hift_sha'])
    tables = hift_ref.make_tables(cfg.hift, seed=9, n_samples=704 * cfg.hift.upsample_total + 480)
    mel_ref = torch.from_numpy(g['mel'])
    f0 = hift_ref.f0_predictor(mel_ref, hsd)
    assert (f0 - torch.from_numpy(g['f0'])).abs().max().item() < 2e-3
    # source from the REFERENCE's f0 (an f0 difference of 1e-5 Hz is a phase difference after 14 s: the stages are held one at a time), decode with it
    s = hift_ref.source_module(F.interpolate(torch.from_numpy(g['f0'])[:, None], scale_factor=float(cfg.hift.upsample_total), mode='nearest').transpose(1, 2),
                               hsd, cfg.hift, tables).transpose(1, 2)
    wav = hift_ref.decode(mel_ref, s, hsd, cfg.hift)
    w, s = wav.reshape(-1).numpy(), s.reshape(-1).numpy()
    assert np.abs(s[::16] - g['src_s16']).max() < 1e-5 and np.abs(s[:32768] - g['src_head']).max() < 1e-5
    d_w = max(np.abs(w[::16] - g['wav_s16']).max(), np.abs(w[:32768] - g['wav_head']).max(), np.abs(w[-32768:] - g['wav_tail']).max())
    print('configs[0], oracle vs the reference: ids equal, mel %.1e of its scale, waveform max |d| %.1e' % (d_mel, d_w))
    assert d_w < 5e-4 and np.abs(w - g['wav_f16'].astype(np.float32)).max() < 1.5e-3, d_w


def test_denoiser_mode_normal_oracle_vs_reference():
    """matcha/hifigan/denoiser.py:20-21 (mode "normal": N(0, 1) probe mel from the global generator): the oracle, seeded like the reference run"""
    from flowmirror_hydravox_amd.config import tiny_hifigan_config
    from oracle import matcha_ref
    g = load_golden('denoiser_normal.npz')
    hc = tiny_hifigan_config()
    sd = W.make_hifigan_state(hc, seed=int(g['weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['weight_sha'])
    torch.manual_seed(int(g['seed']))
    bias = matcha_ref.denoiser_bias(sd, hc, mode='normal')
    assert (bias - torch.from_numpy(g['bias'])).abs().max().item() < 1e-4
    clean = matcha_ref.denoise(torch.from_numpy(g['wav']).squeeze(1), bias, hc, float(g['strength']))
    assert (clean - torch.from_numpy(g['clean'])).abs().max().item() < 1e-4


def test_wave_metrics_and_the_reference_conditioning_fixture():
    """tests/wave_metrics.py on known inputs, and the shape of hift_full_cond.npz (make_golden_fullsize.py hift_cond: the REFERENCE's end-to-end vocoder output
    against itself under last-bit perturbations of its own f0, generator.py:254-260) that tests/test_gpu_refpin.py holds the GPU path to."""
    import wave_metrics as WM
    rng = np.random.default_rng(3)
    t = np.arange(48000) / 24000.0
    a = (0.4 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 660 * t + 1.0) + 0.01 * rng.standard_normal(t.size)).astype(np.float32)
    m0 = WM.all_metrics(a, a)
    assert all(v == 0.0 for v in m0.values()), m0
    m1 = WM.all_metrics(1.01 * a, a)
    assert abs(m1['envelope'] - 0.01) < 2e-3 and abs(m1['stft2048_l2'] - 0.01) < 1e-3 and abs(m1['band_energy'] - 0.0201) < 2e-3, m1
    # a pure phase shift of both partials: large sample-wise, small in the envelope / long-window magnitude
    b = (0.4 * np.sin(2 * np.pi * 220 * t + 0.7) + 0.2 * np.sin(2 * np.pi * 660 * t + 1.0 + 2.1)).astype(np.float32)
    a0 = (0.4 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 660 * t + 1.0)).astype(np.float32)
    m2 = WM.all_metrics(b, a0)
    assert np.abs(b - a0).max() > 0.3 and m2['stft2048_l2'] < 2e-2 and m2['band_energy'] < 2e-2, m2
    g = load_golden('hift_full_cond.npz')
    names = [str(n) for n in g['names']]
    assert names[:5] == ['noise_a', 'noise_b', 'ulp_up', 'ulp_down', 'f0_fp64'] and int(g['T']) == 5632
    for n in names:
        p = g[n + '_per_second']
        assert p.shape == (112,) and float(p.max()) == pytest.approx(float(g[n + '_max']), rel=1e-6) or float(g[n + '_max']) >= float(p.max())
    # the measurement itself: one ulp of f0 moves the reference's own waveform by ~1e-3 within the first second and by > 0.2 of a 0.99 peak over the utterance
    assert float(g['ulp_up_first_second']) < 5e-3 and float(g['ulp_up_max']) > 0.1 and float(g['ulp_down_max']) > 0.1 and 0.9 < float(g['peak']) <= 0.99 + 1e-6
    assert float(g['f0_fp64_minus_fp32']) < 1e-3


def test_whisper_and_kaldi_features_are_pinned_to_an_independent_implementation():
    """SURVEY.md §8(f) N2 (cosyvoice/cli/frontend.py:92-115): openai-whisper and torchaudio are absent, so `whisper.log_mel_spectrogram(n_mels=128)` and
    `kaldi.fbank(num_mel_bins=80, dither=0)` are restatements in oracle/frontend_ref.py — pinned here to the numpy feature extractors of `transformers` (the Whisper
    and SeamlessM4T ones: an implementation that shares no code with the oracle or the product; tests/golden/make_golden_frontend_pins.py minted
    tests/golden/frontend_pins.npz from it), and to the live functions where transformers is importable.  Measured: mel table 7e-16, whisper 6e-5 of a log10 / 4 scale,
    kaldi 6e-4 on natural-log energies up to 26 (fp32 FFTs on both sides)."""
    import os
    import sys
    from oracle import frontend_ref as R
    GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    g = np.load(os.path.join(GOLD, 'frontend_pins.npz'))
    table = R.slaney_mel_table(16000, 400, 128)
    assert np.abs(table.numpy() - g['whisper_mel_table_128']).max() < 1e-12
    worst = [0.0, 0.0]
    for tag in ('a', 'b'):
        y = torch.from_numpy(g['y_' + tag])
        w = R.whisper_log_mel(y, table.float()).numpy()
        assert w.shape == g['whisper128_' + tag].shape == (128, len(y) // 160)
        worst[0] = max(worst[0], float(np.abs(w - g['whisper128_' + tag]).max()))
        k = R.kaldi_fbank(y[None], subtract_mean=False).numpy()
        assert k.shape == g['kaldi80_' + tag].shape == (1 + (len(y) - 400) // 160, 80)
        worst[1] = max(worst[1], float(np.abs(k - g['kaldi80_' + tag]).max()))
        # the frontend's cepstral mean normalisation on top (frontend.py:106)
        km = R.kaldi_fbank(y[None]).numpy()
        assert np.abs(km - (g['kaldi80_' + tag] - g['kaldi80_' + tag].mean(0, keepdims=True))).max() < 2e-3
    print('oracle vs the independent implementation: whisper log-mel %.2e, kaldi fbank %.2e' % tuple(worst))
    assert worst[0] < 3e-4 and worst[1] < 2e-3, worst
    try:
        import transformers  # noqa: F401
    except Exception:
        return
    sys.path.insert(0, GOLD)
    import make_golden_frontend_pins as M
    y = M.waveform(16000 * 2 + 123, seed=77)
    w, tab = M.independent_whisper(y)
    assert np.abs(tab - table.numpy()).max() < 1e-12
    assert np.abs(R.whisper_log_mel(y, table.float()).numpy() - w).max() < 3e-4
    assert np.abs(R.kaldi_fbank(y[None], subtract_mean=False).numpy() - M.independent_kaldi(y)).max() < 2e-3


def test_import_shim_stand_ins_vs_independent_implementations():
    """the reference imports x_transformers' rotary embedding (flow/DiT/modules.py, F7) and diffusers' Attention (matcha's BasicTransformerBlock,
    M3); neither package is in this image, so tests/golden/_ref_shim.py restates them and every reference-run fixture of the DiT / the Matcha decoder went through
    those restatements.  They are held here to independent implementations of the same arithmetic that the image DOES have:
      * rotary: transformers' GPT-J functions (create_sinusoidal_positions / apply_rotary_pos_emb: rotate_every_two on interleaved pairs, sin / cos duplicated
        interleaved) — x_transformers' convention, not the half-split of Llama / Qwen2 —, including the DiT's partial rotation (first 64 of 1024 channels);
      * Attention: torch.nn.functional.scaled_dot_product_attention (what diffusers' AttnProcessor2_0 calls) on the stand-in's own projections, with the additive
        2-D key mask of the Matcha decoder."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import _ref_shim
    S = _ref_shim.stand_ins()
    g = torch.Generator().manual_seed(5)
    # ---- rotary ----
    B, H, T, D = 2, 3, 37, 64
    x = torch.randn(B, H, T, D, generator=g)
    rot = S['RotaryEmbedding'](D)
    freqs, scale = rot.forward_from_seq_len(T)
    mine = S['apply_rotary_pos_emb'](x, freqs, scale)
    try:
        from transformers.models.gptj import modeling_gptj as G
        sincos = G.create_sinusoidal_positions(T, D)
        sin, cos = sincos[:, :D // 2][None], sincos[:, D // 2:][None]
        ind = G.apply_rotary_pos_emb(x.transpose(1, 2), sin, cos).transpose(1, 2)            # GPT-J's layout is (B, T, H, D)
        assert (mine - ind).abs().max().item() < 2e-6, (mine - ind).abs().max().item()
        # the DiT rotates the first 64 channels of its 1024-wide q / k BEFORE the head split (modules.py: apply_rotary_pos_emb on (B, T, 1024)): the rest passes through
        wide = torch.randn(B, T, 1024, generator=g)
        w = S['apply_rotary_pos_emb'](wide, freqs, scale)
        assert torch.equal(w[..., D:], wide[..., D:])
        ind_w = G.apply_rotary_pos_emb(wide[:, :, None, :D], sin, cos)[:, :, 0]
        assert (w[..., :D] - ind_w).abs().max().item() < 2e-6
    except ImportError:
        pass
    # a property no implementation is needed for: position 0 is the identity, and the rotation preserves every (2i, 2i+1) pair's norm
    assert torch.allclose(mine[:, :, 0], x[:, :, 0], atol=1e-7)
    assert torch.allclose(mine.reshape(B, H, T, D // 2, 2).norm(dim=-1), x.reshape(B, H, T, D // 2, 2).norm(dim=-1), atol=1e-5)
    # ---- attention ----
    att = S['Attention'](query_dim=48, heads=4, dim_head=16, bias=False)
    with torch.no_grad():
        for p in att.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    hs = torch.randn(2, 29, 48, generator=g)
    keep = torch.ones(2, 29)
    keep[1, 20:] = 0.0
    add_mask = (1.0 - keep) * -10000.0                                                       # the decoder's additive key mask (matcha/models/components/transformer.py)
    with torch.no_grad():
        mine = att(hs, attention_mask=add_mask)
        q, k, v = (t.view(2, 29, 4, 16).transpose(1, 2) for t in (att.to_q(hs), att.to_k(hs), att.to_v(hs)))
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=add_mask[:, None, None, :])
        ind = att.to_out[0](o.transpose(1, 2).reshape(2, 29, 64))
    assert (mine - ind).abs().max().item() < 2e-6, (mine - ind).abs().max().item()
