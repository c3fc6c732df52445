"""Phase-insensitive distances between two waveforms (test infrastructure; numpy / torch CPU only).

Why they exist: the reference's end-to-end vocoder output is ill-conditioned in its own f0 (SineGen2 accumulates the per-frame phase increment with an fp32
cumsum and multiplies the sum by 2 pi 480, cosyvoice/hifigan/generator.py:254-260), so two correct implementations drift apart in PHASE over a long
utterance while producing the same harmonics.  tests/golden/make_golden_fullsize.py `hift_cond` measures that drift on the REFERENCE ITSELF (its own f0
perturbed by what a different summation order changes) under the sample-wise distance and under the distances below; tests/test_gpu_refpin.py then holds the
GPU path to the same numbers.

  envelope_rel   RMS of every mel frame (480 samples): max |rms_a - rms_b| / max rms_b            (loudness contour; blind to phase entirely)
  stft_mag_rel   |STFT| with a Hann window of `n_fft` samples: || |A| - |B| ||_F / || |B| ||_F     (n_fft = 2048 separates harmonics >= 50 Hz apart; n_fft = 16, hop 4 is
                 the vocoder's own analysis, generator.py:491-505, where several harmonics share a bin)
  band_energy_rel  energy per (1-second block, 1/3-octave-like band): max relative deviation over blocks / bands holding > 1e-3 of the block's energy
"""
import numpy as np
import torch


def envelope_rel(a, b, frame=480):
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    n = (min(a.size, b.size) // frame) * frame
    ra = np.sqrt((a[:n].reshape(-1, frame) ** 2).mean(axis=1))
    rb = np.sqrt((b[:n].reshape(-1, frame) ** 2).mean(axis=1))
    return float(np.abs(ra - rb).max() / max(rb.max(), 1e-30))


def _mag(x, n_fft, hop):
    x = torch.as_tensor(np.asarray(x, dtype=np.float64).reshape(-1))
    return torch.stft(x, n_fft, hop, n_fft, torch.hann_window(n_fft, dtype=torch.float64), center=False, return_complex=True).abs()


def stft_mag_rel(a, b, n_fft=2048, hop=512):
    """(relative Frobenius distance, max |d| / max |B|) of the magnitude spectrograms"""
    A, B = _mag(a, n_fft, hop), _mag(b, n_fft, hop)
    return float((A - B).norm() / B.norm().clamp_min(1e-30)), float((A - B).abs().max() / B.abs().max().clamp_min(1e-30))


def band_energy_rel(a, b, sr=24000, n_fft=2048, hop=512, block_s=1.0, floor=1e-3):
    """energy per (block of `block_s` seconds, band) from the magnitude spectrogram; bands: 0-125-250-500-1k-2k-4k-8k-12k Hz"""
    A, B = _mag(a, n_fft, hop) ** 2, _mag(b, n_fft, hop) ** 2
    edges = [0, 125, 250, 500, 1000, 2000, 4000, 8000, sr // 2 + 1]
    f = np.arange(A.shape[0]) * sr / n_fft
    per = max(1, int(round(block_s * sr / hop)))
    nb = A.shape[1] // per
    worst = 0.0
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = torch.from_numpy((f >= lo) & (f < hi))
        ea = A[sel][:, :nb * per].reshape(-1, nb, per).sum(dim=(0, 2))
        eb = B[sel][:, :nb * per].reshape(-1, nb, per).sum(dim=(0, 2))
        tot = B[:, :nb * per].reshape(B.shape[0], nb, per).sum(dim=(0, 2))
        keep = eb > floor * tot
        if keep.any():
            worst = max(worst, float(((ea - eb).abs() / eb)[keep].max()))
    return worst


def all_metrics(a, b):
    l2, mx = stft_mag_rel(a, b)
    l2s, mxs = stft_mag_rel(a, b, n_fft=16, hop=4)
    return dict(envelope=envelope_rel(a, b), stft2048_l2=l2, stft2048_max=mx, stft16_l2=l2s, stft16_max=mxs, band_energy=band_energy_rel(a, b))
