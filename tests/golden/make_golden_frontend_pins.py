#!/usr/bin/env python3
"""Mint tests/golden/frontend_pins.npz: the zero-shot frontend's two acoustic features computed by an INDEPENDENT implementation.

SURVEY.md §8(f) N2: `whisper.log_mel_spectrogram(speech, n_mels=128)` and `torchaudio.compliance.kaldi.fbank(speech, num_mel_bins=80, dither=0,
sample_frequency=16000)` (cosyvoice/cli/frontend.py:92-115) come from packages this image does not have (openai-whisper, torchaudio), so oracle/frontend_ref.py
restates them.  What the image DOES have is `transformers` (5.15.0), whose numpy feature extractors re-implement exactly these two front ends for the Whisper and
SeamlessM4T models (`transformers.audio_utils.spectrogram / mel_filter_bank / window_function`: Slaney-normalised mel table + hann STFT + log10 / max - 8 / (x + 4) / 4
for whisper; "mimic Kaldi": Povey window, DC removal, pre-emphasis 0.97, 512-point FFT, Kaldi mel scale from 20 Hz, natural log with the float epsilon floor).  They share
no code with the oracle or the product, so agreement pins both restatements.  This script runs them on seeded waveforms and stores inputs + outputs; the tests hold
the oracle (CPU) and the device path (GPU) to these arrays, and — where transformers is importable — to the live functions as well.

    python tests/golden/make_golden_frontend_pins.py
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def waveform(n, seed, sr=16000):
    """a harmonic source with vibrato and an envelope, a silent gap (the floors), and noise"""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n, dtype=torch.float64) / sr
    f0 = 140.0 + 30.0 * torch.sin(2 * np.pi * 0.7 * t)
    ph = 2 * np.pi * torch.cumsum(f0, 0) / sr
    y = sum(torch.sin(k * ph) / k for k in range(1, 12)) * (0.5 + 0.5 * torch.sin(2 * np.pi * 1.3 * t)) * 0.2
    y[n // 3: n // 3 + sr // 4] = 0.0
    return (y + 0.003 * torch.randn(n, generator=g, dtype=torch.float64)).float()


def independent_whisper(y, n_mels=128):
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor(feature_size=n_mels)                    # n_fft 400, hop 160, 16 kHz: whisper's constants
    return fe._np_extract_fbank_features(y.numpy()[None], 'cpu')[0].astype(np.float32), fe.mel_filters.T.astype(np.float64)


def independent_kaldi(y):
    """-> (frames, 80) log mel energies of `y` taken as it is (the reference feeds [-1, 1] floats to kaldi.fbank, frontend.py:104).  The extractor multiplies by
    2^15 ("Kaldi compliance: 16-bit signed integers"): dividing the waveform by 2^15 first undoes exactly that (a power of two: no rounding)."""
    from transformers import SeamlessM4TFeatureExtractor
    fe = SeamlessM4TFeatureExtractor()                                   # 80 bins, 16 kHz
    return fe._extract_fbank_features((y / 32768.0).numpy()).astype(np.float32)


def main():
    import transformers
    out = {'transformers_version': np.array(transformers.__version__)}
    for tag, n, seed in (('a', 16000 * 3 + 77, 11), ('b', 16000 + 5, 12)):
        y = waveform(n, seed)
        w, table = independent_whisper(y)
        out['y_' + tag] = y.numpy()
        out['whisper128_' + tag] = w
        out['kaldi80_' + tag] = independent_kaldi(y)
    out['whisper_mel_table_128'] = table
    np.savez_compressed(os.path.join(HERE, 'frontend_pins.npz'), **out)
    for k, v in out.items():
        print(k, getattr(v, 'shape', None))


if __name__ == '__main__':
    main()
