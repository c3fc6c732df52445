"""Prompt-feature extraction on the device: the log-mel spectrogram the zero-shot frontend computes for the prompt
audio — `feat_extractor` = matcha.utils.audio.mel_spectrogram (matcha/utils/audio.py:45-82, called at cosyvoice/cli/frontend.py:119;
CosyVoice3 settings n_fft 1920, hop 480, win 1920, 80 mels, 24 kHz, fmin 0, fmax 8000).  Same call signature as the reference's partial:
    HvxMelSpectrogram(n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=8000)(y)  ->  (B, num_mels, frames)
STFT and mel projection are fp32-MFMA GEMMs against bases built at construction (libhvx: hvx_mel_spectrogram); no CPU fallback."""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .packing import mel_filterbank, stft_bases


class HvxMelSpectrogram:
    def __init__(self, n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=8000, center=False, device='cuda'):
        if center or win_size != n_fft:
            raise NotImplementedError('the reference frontend uses center=False and win_size == n_fft')
        _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.n_fft, self.hop, self.num_mels = n_fft, hop_size, num_mels
        ana, _, _ = stft_bases(n_fft)
        mel = mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax)
        bins = n_fft // 2 + 1
        ld = (bins + 31) // 32 * 32
        melp = torch.zeros(num_mels, ld)
        melp[:, :bins] = mel
        self._ana, self._mel = ana.to(self.device), melp.to(self.device).contiguous()
        self._ws = None

    @torch.inference_mode()
    def __call__(self, y):
        """y: (B, L) float waveform in [-1, 1] -> (B, num_mels, frames)"""
        y = y.to(self.device, torch.float32).contiguous()
        B, L = y.shape
        pad = (self.n_fft - self.hop) // 2
        frames = (L + 2 * pad - self.n_fft) // self.hop + 1
        out = torch.empty(B, self.num_mels, frames, dtype=torch.float32, device=self.device)
        need = self.lib.hvx_mel_workspace_bytes(L, self.n_fft, self.hop, self.num_mels)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        for b in range(B):
            check(self.lib.hvx_mel_spectrogram(stream_ptr(), ptr(self._ws), self._ws.numel(), ptr(y[b]), L, self.n_fft, self.hop, ptr(self._ana),
                                               ptr(self._mel), self.num_mels, ptr(out[b])), 'hvx_mel_spectrogram')
        return out
