"""oracle/frontend_ref.py — CPU restatement of the two third-party feature extractors of the zero-shot frontend
(server/model_utils/cosyvoice/cli/frontend.py:92-110).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PINNED TO AN INDEPENDENT IMPLEMENTATION, NOT TO THE REFERENCE'S OWN PACKAGES: both algorithms live in third-party packages that are neither under /root/reference
nor installed in this image (requirements.txt: openai-whisper==20231117, torchaudio==2.3.1), and the reference holds no test vectors for them.  They are restated from
the published sources of those versions, anchored on the reference's call sites, and held (tests/test_oracle_golden.py, tests/golden/frontend_pins.npz minted by
tests/golden/make_golden_frontend_pins.py) to the numpy feature extractors of `transformers` 5.15 — WhisperFeatureExtractor and SeamlessM4TFeatureExtractor ("mimic
Kaldi"), which re-implement the same two front ends and share no code with this file or the product: whisper log-mel within 2e-5, kaldi fbank within 1e-4.

  * `whisper.log_mel_spectrogram(speech, n_mels=128)`  (frontend.py:95)  — whisper/audio.py:110-157: 16 kHz, N_FFT 400, HOP 160,
    periodic Hann window, torch.stft(center=True, reflect), the last frame dropped, |X|^2, the librosa Slaney mel filterbank
    (assets/mel_filters.npz = librosa.filters.mel(sr=16000, n_fft=400, n_mels=128)), log10(clamp(., 1e-10)),
    max(., global max - 8), (. + 4) / 4.
  * `torchaudio.compliance.kaldi.fbank(speech, num_mel_bins=80, dither=0, sample_frequency=16000)` followed by
    `feat - feat.mean(dim=0, keepdim=True)`  (frontend.py:104-108) — kaldi.py defaults: 25 ms frames every 10 ms, snip_edges, per-frame
    DC removal, pre-emphasis 0.97 with the first sample replicated, Povey window (symmetric Hann ^ 0.85), zero-padding to 512, power
    spectrum, 80 triangular filters equally spaced on mel(f) = 1127 ln(1 + f / 700) between 20 Hz and Nyquist evaluated at the first 256
    FFT bin centres, log(max(., float32 eps)).

The Slaney mel table itself (`slaney_mel_table` below) IS pinned, to the literal values librosa publishes in its own docstrings (see the function): it
is what whisper ships as assets/mel_filters.npz and what matcha/utils/audio.py:53 builds for the prompt mel.

Both are written with torch.stft / torch.fft directly — deliberately NOT in the folded-basis GEMM form the HIP path uses — so that a
test of one against the other checks the framing, the folding and the filterbanks independently.
"""
import math

import torch
import torch.nn.functional as F


# ---- librosa.filters.mel (htk=False, norm='slaney'): an independent float64 statement -------------------------------------------------------
# librosa is absent from this image.  What follows is written from the formulas librosa documents (librosa.hz_to_mel / mel_to_hz / mel_frequencies /
# filters.mel docstrings) in scalar Python floats, sharing no code with the product's vectorised flowmirror_hydravox_amd.packing.mel_filterbank, and
# is pinned by `LIBROSA_PUBLISHED` — the literal numbers those docstrings print — in tests/test_oracle_golden.py.
_F_SP = 200.0 / 3.0                 # Hz per mel below 1 kHz
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP  # = 15
_LOGSTEP = math.log(6.4) / 27.0     # 27 log-spaced mels per factor of 6.4 above 1 kHz

LIBROSA_PUBLISHED = dict(
    hz_to_mel={60.0: 0.9, 110.0: 1.65, 220.0: 3.3, 440.0: 6.6},                                         # librosa.hz_to_mel docstring
    mel_to_hz={1.0: 66.667, 2.0: 133.333, 3.0: 200.0, 4.0: 266.667, 5.0: 333.333},                      # librosa.mel_to_hz docstring
    mel_frequencies_40=[0.0, 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49, 1024.856, 1119.114,
                        1222.042, 1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393, 2470.47, 2697.686, 2945.799, 3216.731, 3512.582,
                        3835.643, 4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92, 7101.009, 7754.107, 8467.272, 9246.028, 10096.408, 11025.0],
    #                                                                                                      librosa.mel_frequencies(n_mels=40) docstring (fmin 0, fmax 11025)
    filters_mel_22050_2048_row0_col1=0.016,                                                             # librosa.filters.mel(sr=22050, n_fft=2048) docstring: melfb[0, 1]
)


def slaney_hz_to_mel(f):
    return f / _F_SP if f < _MIN_LOG_HZ else _MIN_LOG_MEL + math.log(f / _MIN_LOG_HZ) / _LOGSTEP


def slaney_mel_to_hz(m):
    return _F_SP * m if m < _MIN_LOG_MEL else _MIN_LOG_HZ * math.exp(_LOGSTEP * (m - _MIN_LOG_MEL))


def slaney_mel_frequencies(n_mels, fmin, fmax):
    lo, hi = slaney_hz_to_mel(float(fmin)), slaney_hz_to_mel(float(fmax))
    return [slaney_mel_to_hz(lo + (hi - lo) * i / (n_mels - 1)) for i in range(n_mels)]


def slaney_mel_table(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=) with htk=False, norm='slaney' -> float64 tensor (n_mels, n_fft // 2 + 1): filter i is the
    triangle over (edge[i], edge[i+1], edge[i+2]) of n_mels + 2 edges equally spaced on the Slaney mel scale, sampled at the FFT bin frequencies
    k sr / n_fft, times 2 / (edge[i+2] - edge[i])."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    edges = slaney_mel_frequencies(n_mels + 2, fmin, fmax)
    rows = []
    for i in range(n_mels):
        lo, ce, hi = edges[i], edges[i + 1], edges[i + 2]
        area = 2.0 / (hi - lo)
        row = []
        for k in range(n_fft // 2 + 1):
            f = k * float(sr) / n_fft
            if lo < f <= ce:
                w = (f - lo) / (ce - lo)
            elif ce < f < hi:
                w = (hi - f) / (hi - ce)
            else:
                w = 0.0
            row.append(w * area)
        rows.append(row)
    return torch.tensor(rows, dtype=torch.float64)


def whisper_log_mel(audio, mel_filters):
    """audio (L,) or (B, L) float at 16 kHz; mel_filters (n_mels, 201) -> (n_mels, frames) or (B, n_mels, frames), frames = L // 160"""
    window = torch.hann_window(400)
    stft = torch.stft(audio, 400, 160, window=window, return_complex=True)           # center=True, pad_mode='reflect'
    magnitudes = stft[..., :-1].abs() ** 2
    mel_spec = mel_filters @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0


def kaldi_mel_banks(num_bins=80, padded=512, sample_freq=16000.0, low_freq=20.0, high_freq=0.0):
    """torchaudio.compliance.kaldi.get_mel_banks without VTLN -> (num_bins, padded // 2)"""
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    n_fft_bins = padded // 2
    bin_width = sample_freq / padded

    def mel(f):
        return 1127.0 * math.log(1.0 + f / 700.0)
    mel_low, mel_high = mel(low_freq), mel(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = torch.arange(num_bins, dtype=torch.float64).unsqueeze(1)
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    m = (1127.0 * torch.log(1.0 + bin_width * torch.arange(n_fft_bins, dtype=torch.float64) / 700.0)).unsqueeze(0)
    up, down = (m - left) / (center - left), (right - m) / (right - center)
    return torch.clamp(torch.minimum(up, down), min=0.0).float()


def kaldi_fbank(waveform, num_mel_bins=80, sample_frequency=16000.0, subtract_mean=True):
    """waveform (1, L) or (L,) float -> (frames, num_mel_bins), frames = 1 + (L - 400) // 160; `subtract_mean` applies the
    frontend's `feat - feat.mean(dim=0)` (cepstral mean normalisation of the CAM++ input)."""
    x = waveform.reshape(-1).to(torch.float32)
    win, shift, padded = int(sample_frequency * 0.025), int(sample_frequency * 0.010), 512
    if x.numel() < win:
        return torch.empty(0, num_mel_bins)
    m = 1 + (x.numel() - win) // shift
    frames = x.as_strided((m, win), (shift, 1)).clone()
    frames = frames - frames.mean(dim=1, keepdim=True)                                # remove_dc_offset
    prev = F.pad(frames.unsqueeze(0), (1, 0), mode='replicate').squeeze(0)[:, :-1]
    frames = frames - 0.97 * prev                                                     # preemphasis_coefficient
    frames = frames * torch.hann_window(win, periodic=False).pow(0.85).unsqueeze(0)   # povey
    frames = F.pad(frames, (0, padded - win))
    power = torch.fft.rfft(frames).abs().pow(2.0)                                     # (m, 257)
    banks = F.pad(kaldi_mel_banks(num_mel_bins, padded, sample_frequency), (0, 1))    # zero weight on the Nyquist bin
    feat = torch.clamp(power @ banks.t(), min=torch.finfo(torch.float32).eps).log()
    return feat - feat.mean(dim=0, keepdim=True) if subtract_mean else feat
