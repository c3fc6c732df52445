"""Parity of the Matcha-TTS family through the C-ABI against the golden vectors minted from the reference's own
modules (tests/golden/make_golden.py: gen_matcha) and against the CPU oracle on further seeded inputs.  fp32 throughout; tolerances are
relative to the signal scale and written at each assertion."""
import numpy as np
import pytest
import torch

from conftest import load_golden, state_checksum

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


@pytest.fixture(scope='module')
def golden():
    return load_golden('matcha_tiny.npz')


def test_matcha_decoder_and_euler_solver_vs_reference(golden):
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.config import tiny_matcha_config
    from flowmirror_hydravox_amd.matcha import HvxMatchaCFM
    g = golden
    c = tiny_matcha_config()
    sd = W.make_matcha_state(c, seed=int(g['m_weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['m_weight_sha'])
    cfm = HvxMatchaCFM(c, {'estimator.' + k: v for k, v in sd.items()})          # checkpoint keys as CFM.state_dict() has them
    for r in range(2):
        p = 'm%d_' % r
        x, mu, spks, t = (torch.from_numpy(g[p + k]) for k in ('x', 'mu', 'spks', 't'))
        mask = torch.ones(1, 1, x.shape[-1])
        y = cfm.estimator(x, mask, mu, t, spks).cpu()
        assert _rel(y.numpy(), g[p + 'y']) < 1e-3                                 # one estimator call (M2/M3)
        s = cfm.forward(mu, mask, c.n_timesteps, temperature=c.temperature, spks=spks, noise=torch.from_numpy(g[p + 'noise'])).cpu()
        assert _rel(s.numpy(), g[p + 'sample']) < 2e-3                            # BASECFM.solve_euler (M1)


def test_matcha_decoder_rejects_padded_batches_in_matcha_mode(golden):
    from flowmirror_hydravox_amd import _lib, weights as W
    from flowmirror_hydravox_amd.config import tiny_matcha_config
    from flowmirror_hydravox_amd.matcha import HvxMatchaDecoder
    c = tiny_matcha_config()
    dec = HvxMatchaDecoder(c, W.make_matcha_state(c, seed=3, init='fan_in'))
    x = torch.randn(2, c.mel, 16)
    mask = torch.ones(2, 1, 16)
    mask[1, :, 10:] = 0
    with pytest.raises(_lib.HvxError):       # the Matcha decoder ADDS its 0/1 mask to the scores: only full-length masks are served
        dec(x, mask, x, torch.tensor([0.5, 0.5]), torch.randn(2, c.spk_dim))


def test_conditional_decoder_padded_batch_vs_reference(golden):
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.config import tiny_matcha_config
    from flowmirror_hydravox_amd.matcha import HvxMatchaDecoder
    g = golden
    cc = tiny_matcha_config(cv=True)
    sd = W.make_matcha_state(cc, seed=int(g['c_weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['c_weight_sha'])
    dec = HvxMatchaDecoder(cc, sd)
    x, mask, mu, t, spks, cond = (torch.from_numpy(g['c_' + k]) for k in ('x', 'mask', 'mu', 't', 'spks', 'cond'))
    y = dec(x, mask, mu, t, spks, cond).cpu().numpy()            # B = 2, lengths 37 / 29 of 37 (odd: skip trimming), key-padding masks
    assert _rel(y, g['c_y']) < 1e-3
    assert np.abs(y[1, :, 29:]).max() == 0.0                     # output * mask


def test_conditional_decoder_longer_input_vs_oracle():
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.config import tiny_matcha_config
    from flowmirror_hydravox_amd.matcha import HvxMatchaDecoder
    from oracle import matcha_ref
    cc = tiny_matcha_config(cv=True)
    sd = W.make_matcha_state(cc, seed=31, init='fan_in')
    dec = HvxMatchaDecoder(cc, sd)
    gq = torch.Generator().manual_seed(32)
    T = 301
    x, mu, cond = (torch.randn(1, cc.mel, T, generator=gq) for _ in range(3))
    spks = torch.randn(1, cc.spk_dim, generator=gq)
    mask = torch.ones(1, 1, T)
    t = torch.tensor([0.6])
    ref = matcha_ref.decoder_forward(sd, cc, x, mask, mu, t, spks, cond)
    y = dec(x, mask, mu, t, spks, cond).cpu()
    assert _rel(y.numpy(), ref.numpy()) < 1e-3


def test_hifigan_generator_and_denoiser_vs_reference(golden):
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.config import tiny_hifigan_config
    from flowmirror_hydravox_amd.matcha import HvxDenoiser, HvxHifiGan
    g = golden
    hc = tiny_hifigan_config()
    sd = W.make_hifigan_state(hc, seed=int(g['g_weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['g_weight_sha'])
    voc = HvxHifiGan(hc, sd)
    wav = voc(torch.from_numpy(g['g_mel'])).cpu().numpy()
    assert wav.shape == g['g_wav'].shape
    assert _rel(wav, g['g_wav']) < 1e-3                          # Generator.forward (M4): ConvTranspose as phase convolutions
    den = HvxDenoiser(voc)
    assert _rel(den.bias_spec.cpu().numpy(), g['g_bias'][0, :, 0]) < 1e-3
    clean = den(torch.from_numpy(g['g_wav']).squeeze(1), strength=float(g['g_strength'])).cpu().numpy()
    assert clean.shape == g['g_clean'].shape
    assert _rel(clean, g['g_clean']) < 1e-3                      # Denoiser.forward (M5)


def test_denoiser_mode_normal_vs_reference():
    """matcha/hifigan/denoiser.py:20-21: mode="normal" — the bias spectrum comes from the vocoder's answer to an N(0, 1) probe mel drawn from the global CPU
    generator; seeded like the reference run that minted the fixture, bias spectrum and denoised audio agree"""
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.config import tiny_hifigan_config
    from flowmirror_hydravox_amd.matcha import HvxDenoiser, HvxHifiGan
    g = load_golden('denoiser_normal.npz')
    hc = tiny_hifigan_config()
    sd = W.make_hifigan_state(hc, seed=int(g['weight_seed']), init='fan_in')
    assert state_checksum(sd) == str(g['weight_sha'])
    voc = HvxHifiGan(hc, sd)
    torch.manual_seed(int(g['seed']))
    den = HvxDenoiser(voc, mode='normal')
    assert _rel(den.bias_spec.cpu().numpy(), g['bias'][0, :, 0]) < 1e-3
    clean = den(torch.from_numpy(g['wav']).squeeze(1), strength=float(g['strength'])).cpu().numpy()
    assert clean.shape == g['clean'].shape and _rel(clean, g['clean']) < 1e-3
    zeros = HvxDenoiser(voc)
    assert _rel(zeros.bias_spec.cpu().numpy(), g['bias'][0, :, 0]) > 1e-2          # (it is a different spectrum than the default mode's)
    with pytest.raises(Exception):
        HvxDenoiser(voc, mode='uniform')


def test_hifigan_accepts_checkpoints_with_weight_norm_removed(golden):
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.config import tiny_hifigan_config
    from flowmirror_hydravox_amd.matcha import HvxHifiGan, _fold_wn_old
    hc = tiny_hifigan_config()
    sd = W.make_hifigan_state(hc, seed=40, init='fan_in')
    names = sorted({k.rsplit('.', 1)[0] for k in sd})
    folded = {}
    for n in names:                                              # what Generator.remove_weight_norm() leaves behind
        folded[n + '.weight'] = _fold_wn_old(sd, n)
        folded[n + '.bias'] = sd[n + '.bias']
    mel = torch.randn(1, hc.mel, 12, generator=torch.Generator().manual_seed(41))
    assert torch.equal(HvxHifiGan(hc, sd)(mel), HvxHifiGan(hc, folded)(mel))


def test_prompt_mel_spectrogram_vs_reference(golden):
    """the frontend's prompt log-mel on the device (STFT and mel projection as fp32-MFMA GEMMs)"""
    from flowmirror_hydravox_amd.frontend import HvxMelSpectrogram
    g = golden
    for tag, kw in (('cv3', dict(n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=8000)),
                    ('toy', dict(n_fft=64, num_mels=8, sampling_rate=800, hop_size=32, win_size=64, fmin=0, fmax=400))):
        mel = HvxMelSpectrogram(**kw)
        out = mel(torch.from_numpy(g['mel_%s_y' % tag])).cpu().numpy()
        ref = g['mel_%s_out' % tag]
        assert out.shape == ref.shape
        # log domain: absolute tolerance (values span about [-11.5, 3]); 1e-3 abs is 0.1 % in the linear domain
        assert np.abs(out - ref).max() < 2e-3


def _speechlike(n, seed, sr=16000):
    """a few seconds of a harmonic source with vibrato, an amplitude envelope, a silent gap and noise: exercises the dynamic range
    (whisper's max - 8 floor, kaldi's epsilon floor) better than white noise"""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n, dtype=torch.float64) / sr
    f0 = 140.0 + 30.0 * torch.sin(2 * np.pi * 0.7 * t)
    ph = 2 * np.pi * torch.cumsum(f0, 0) / sr
    y = sum(torch.sin(k * ph) / k for k in range(1, 12)) * (0.5 + 0.5 * torch.sin(2 * np.pi * 1.3 * t)) * 0.2
    y[n // 3: n // 3 + sr // 4] = 0.0
    return (y + 0.003 * torch.randn(n, generator=g, dtype=torch.float64)).float()


@pytest.mark.parametrize('n', [16000 * 3 + 77, 160 * 5, 401, 16000 * 12])
def test_whisper_log_mel_vs_oracle(n):
    """whisper.log_mel_spectrogram(speech, n_mels=128) (frontend.py:95) on the device against the torch.stft
    restatement in oracle/frontend_ref.py (third-party algorithm, parity unpinned: see the oracle's header).  Log10 domain scaled by 1/4:
    2e-3 abs is about 2 % in the linear domain at the floor and far less above it."""
    from flowmirror_hydravox_amd.frontend import HvxWhisperLogMel
    from flowmirror_hydravox_amd.packing import mel_filterbank
    from oracle import frontend_ref
    y = _speechlike(n, seed=n)
    ref = frontend_ref.whisper_log_mel(y, mel_filterbank(16000, 400, 128, 0.0, 8000.0))
    fe = HvxWhisperLogMel(128)
    out = fe(y).cpu()
    assert out.shape == ref.shape == (128, n // 160)
    assert (out - ref).abs().max().item() < 2e-3
    both = fe(torch.stack([y, y.flip(0)])).cpu()                                   # batched call: per-waveform dynamic range
    assert torch.equal(both[0], out)
    assert (both[1] - frontend_ref.whisper_log_mel(y.flip(0), mel_filterbank(16000, 400, 128, 0.0, 8000.0))).abs().max().item() < 2e-3


@pytest.mark.parametrize('n', [16000 * 3 + 77, 400, 560, 16000 * 9])
def test_kaldi_fbank_vs_oracle(n):
    """kaldi.fbank(num_mel_bins=80, dither=0, 16 kHz) + mean subtraction (frontend.py:104-108): the device path folds DC removal,
    pre-emphasis, the Povey window and the FFT zero-padding into one GEMM basis; the oracle applies them one by one with torch.fft.
    Natural-log domain: 5e-3 abs (power-spectrum cancellation in near-silent bins carries the fp32 GEMM's rounding)."""
    from flowmirror_hydravox_amd.frontend import HvxKaldiFbank
    from oracle import frontend_ref
    y = _speechlike(n, seed=n + 1)
    for sub in (True, False):
        ref = frontend_ref.kaldi_fbank(y[None], subtract_mean=sub)
        out = HvxKaldiFbank(80, subtract_mean=sub)(y[None]).cpu()
        assert out.shape == ref.shape == (1 + (n - 400) // 160, 80)
        assert (out - ref).abs().max().item() < 5e-3, (sub, (out - ref).abs().max().item())
    assert HvxKaldiFbank(80)(y[:399]).shape == (0, 80)


def test_device_whisper_and_kaldi_features_vs_the_independent_fixture():
    """The device front ends against tests/golden/frontend_pins.npz directly: arrays computed by the numpy feature extractors of `transformers` (an implementation
    independent of oracle/ and of the product; tests/golden/make_golden_frontend_pins.py) — whisper's 128-bin log-mel and Kaldi's 80-bin fbank (+ the frontend's mean
    subtraction, cosyvoice/cli/frontend.py:92-115).  Same tolerances as against the oracle: 2e-3 of the log10 / 4 scale, 5e-3 natural-log."""
    import os
    from flowmirror_hydravox_amd.frontend import HvxKaldiFbank, HvxWhisperLogMel
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'frontend_pins.npz'))
    wl, kf, kfm = HvxWhisperLogMel(128), HvxKaldiFbank(80, subtract_mean=False), HvxKaldiFbank(80)
    for tag in ('a', 'b'):
        y = torch.from_numpy(g['y_' + tag])
        dw = np.abs(wl(y).cpu().numpy() - g['whisper128_' + tag]).max()
        k = g['kaldi80_' + tag]
        dk = np.abs(kf(y[None]).cpu().numpy() - k).max()
        dkm = np.abs(kfm(y[None]).cpu().numpy() - (k - k.mean(0, keepdims=True))).max()
        print('device vs the independent implementation (%s): whisper %.2e, kaldi %.2e, mean-normalised %.2e' % (tag, dw, dk, dkm))
        assert dw < 2e-3 and dk < 5e-3 and dkm < 5e-3, (dw, dk, dkm)


def test_zero_shot_frontend_graphs_run_on_the_device():
    """SURVEY.md §8(f) N2, the rest of the row: `_extract_speech_token` / `_extract_spk_embedding` (cosyvoice/cli/frontend.py:92-115) with the ONNX graphs
    executed on the device (frontend.HvxSpeechTokenizer / HvxSpeakerEncoder over onnx_graph.OnnxRunner) instead of onnxruntime CPU sessions.  The real
    assets are absent, so the graphs are the synthetic ones of tests/onnx_synth.py at the real input widths (128 mels / 80 fbank bins); the device result
    is held to the numpy oracle of the graph applied to the device features.  Return types and shapes are the reference's."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import onnx_synth
    from flowmirror_hydravox_amd import onnx_graph as og
    from flowmirror_hydravox_amd.frontend import HvxSpeechTokenizer, HvxSpeakerEncoder
    from oracle import onnx_ref
    g = torch.Generator().manual_seed(11)
    speech = (0.1 * torch.randn(1, 16000 * 2 + 123, generator=g)).clamp(-1, 1)
    tok_model = og.save_onnx(onnx_synth.tokenizer_like(n_mels=128))
    tk = HvxSpeechTokenizer(tok_model)
    tok, tok_len = tk(speech)
    T = speech.shape[1] // 160
    assert tok.dtype == torch.int32 and tuple(tok.shape) == (1, (T + 1) // 2) and tok_len.tolist() == [(T + 1) // 2] and tok.is_cuda
    ref = onnx_ref.run(og.load_onnx(tok_model), {'mel': tk.feat(speech).cpu().numpy()})
    safe = (np.abs(np.abs(ref['latent'] * 0.999) - 0.5) > 1e-3).all(-1)
    assert safe.mean() > 0.9 and np.array_equal(tok.cpu().numpy()[safe], ref['tokens'][safe])
    with pytest.raises(AssertionError):
        tk(torch.zeros(1, 16000 * 30 + 1))                                      # (frontend.py:94)
    # the reference's two-input form: (features, their length as int32 [1]) by position; keys past the length are masked inside the graph
    len_model = og.save_onnx(onnx_synth.tokenizer_like(n_mels=128, with_length=True))
    tk2 = HvxSpeechTokenizer(len_model)
    assert tk2.runner.g.inputs == ['mel', 'mel_len']
    tok2, _ = tk2(speech)
    feat = tk2.feat(speech).cpu().numpy()
    ref2 = onnx_ref.run(og.load_onnx(len_model), {'mel': feat, 'mel_len': np.array([feat.shape[2]], np.int32)})
    safe2 = (np.abs(np.abs(ref2['latent'] * 0.999) - 0.5) > 1e-3).all(-1)
    assert np.array_equal(tok2.cpu().numpy()[safe2], ref2['tokens'][safe2])
    short = onnx_ref.run(og.load_onnx(len_model), {'mel': feat, 'mel_len': np.array([feat.shape[2] // 2], np.int32)})
    assert not np.array_equal(short['tokens'], ref2['tokens'])                  # (the mask is live: a shorter length changes the ids)
    got_short = tk2.runner.run({'mel': feat, 'mel_len': np.array([feat.shape[2] // 2], np.int32)})
    assert np.abs(got_short['latent'] - short['latent']).max() < 2e-4
    spk_model = og.save_onnx(onnx_synth.campplus_like(emb=192))
    enc = HvxSpeakerEncoder(spk_model)
    emb = enc(speech)
    assert emb.dtype == torch.float32 and tuple(emb.shape) == (1, 192) and emb.is_cuda
    ref = onnx_ref.run(og.load_onnx(spk_model), {'fbank': enc.feat(speech).unsqueeze(0).cpu().numpy()})['embedding']
    assert np.abs(emb.cpu().numpy() - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
