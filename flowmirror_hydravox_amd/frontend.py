"""Prompt-feature extraction on the device: the three spectral features the zero-shot frontend computes from the prompt
audio (cosyvoice/cli/frontend.py:92-122) — HvxMelSpectrogram (prompt log-mel for the flow, below), HvxWhisperLogMel (128-bin input of the
speech tokenizer, :95) and HvxKaldiFbank (80-bin input of the CAM++ speaker encoder with its mean subtraction, :104-108).  The ONNX
graphs those last two feed are assets of the weights repository and stay outside this build; what runs on the host in the reference —
the feature extraction — runs here as two GEMMs (framing x folded DFT basis, power x mel filterbank) in libhvx: hvx_frame_features.

The log-mel spectrogram the zero-shot frontend computes for the prompt audio — `feat_extractor` = matcha.utils.audio.mel_spectrogram (matcha/utils/audio.py:45-82, called at cosyvoice/cli/frontend.py:119;
CosyVoice3 settings n_fft 1920, hop 480, win 1920, 80 mels, 24 kHz, fmin 0, fmax 8000).  Same call signature as the reference's partial:
    HvxMelSpectrogram(n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=8000)(y)  ->  (B, num_mels, frames)
STFT and mel projection are fp32-MFMA GEMMs against bases built at construction (libhvx: hvx_mel_spectrogram); no CPU fallback."""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
import ctypes as C

from .packing import kaldi_fbank_bases, mel_filterbank, stft_bases, whisper_bases


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


class _FramedFeatures:
    """shared driver of hvx_frame_features: bases on the device, a growing workspace, one launch chain per waveform"""

    def __init__(self, basis, mel, device):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        self._basis, self._mel = basis.to(self.device).contiguous(), mel.to(self.device).contiguous()
        self._ws = None

    def _run(self, y, cfg, out):
        need = self.lib.hvx_frame_features_workspace_bytes(y.numel(), C.byref(cfg))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        check(self.lib.hvx_frame_features(stream_ptr(), ptr(self._ws), self._ws.numel(), ptr(y), y.numel(), C.byref(cfg), ptr(self._basis), ptr(self._mel),
                                          ptr(out)), 'hvx_frame_features')
        return out


class HvxWhisperLogMel(_FramedFeatures):
    """`whisper.log_mel_spectrogram(audio, n_mels=128)` (frontend.py:95; whisper/audio.py:110-157) for 16 kHz audio:
    (L,) or (B, L) -> (n_mels, L // 160) or (B, n_mels, L // 160)"""

    def __init__(self, n_mels=128, device='cuda'):
        basis, mel = whisper_bases(n_mels)
        super().__init__(basis, mel, device)
        self.n_mels = n_mels

    @torch.inference_mode()
    def __call__(self, audio):
        a = audio.to(self.device, torch.float32)
        single = a.dim() == 1
        a = a.reshape(-1, a.shape[-1]).contiguous()
        B, L = a.shape
        if L <= 200:
            raise ValueError('whisper log-mel: the audio must be longer than the 200-sample reflection pad')
        frames = L // 160                                    # 1 + L // 160 STFT frames, the last one dropped (audio.py:149)
        if frames < 1:
            raise ValueError('whisper log-mel: audio shorter than one hop')
        cfg = _lib.FeatureConfig(frame_len=416, hop=160, reflect_pad=200, n_frames=frames, bins=201, power=1, mag_eps=0.0, n_mels=self.n_mels,
                                 log_floor=1e-10, log_scale=0.4342944819032518, post=1, time_major=0)
        out = torch.empty(B, self.n_mels, frames, dtype=torch.float32, device=self.device)
        for b in range(B):                                   # the dynamic-range floor is per call (log_spec.max()): one waveform at a time
            self._run(a[b], cfg, out[b])
        return out[0] if single else out


class HvxKaldiFbank(_FramedFeatures):
    """`kaldi.fbank(speech, num_mel_bins=80, dither=0, sample_frequency=16000)` and the frontend's mean subtraction (frontend.py:104-108):
    (1, L) or (L,) -> (1 + (L - 400) // 160, num_mel_bins)"""

    def __init__(self, num_mel_bins=80, sample_frequency=16000, subtract_mean=True, device='cuda'):
        if int(sample_frequency) != 16000:
            raise NotImplementedError('the reference frontend extracts speaker features at 16 kHz')
        basis, mel = kaldi_fbank_bases(num_mel_bins, float(sample_frequency))
        super().__init__(basis, mel, device)
        self.num_mel_bins, self.subtract_mean = num_mel_bins, subtract_mean

    @torch.inference_mode()
    def __call__(self, waveform):
        a = waveform.to(self.device, torch.float32).reshape(-1).contiguous()
        L = a.numel()
        if L < 400:
            return torch.empty(0, self.num_mel_bins, device=self.device)
        frames = 1 + (L - 400) // 160
        cfg = _lib.FeatureConfig(frame_len=416, hop=160, reflect_pad=0, n_frames=frames, bins=257, power=1, mag_eps=0.0, n_mels=self.num_mel_bins,
                                 log_floor=1.1920928955078125e-07, log_scale=1.0, post=2 if self.subtract_mean else 0, time_major=1)
        return self._run(a, cfg, torch.empty(frames, self.num_mel_bins, dtype=torch.float32, device=self.device))


class HvxSpeechTokenizer:
    """`CosyVoiceFrontEnd._extract_speech_token` on the device (cosyvoice/cli/frontend.py:92-103): 16 kHz prompt audio -> whisper 128-bin log-mel
    (HvxWhisperLogMel) -> the speech-tokenizer ONNX graph through the device executor (onnx_graph.OnnxRunner instead of an onnxruntime CPU session) ->
    (speech_token int32 [1][N], speech_token_len int32 [1]).  The graph's inputs are taken by position, as the reference does
    (`session.get_inputs()[0]` = the features [1][128][T], `[1]` = their length as int32 [1]); a graph with a single input gets the features only.
    `speech_tokenizer_v3.onnx` itself is an asset of the weights repository (not in the tree): parity with onnxruntime on it is unpinned."""

    def __init__(self, onnx_model, device='cuda', allow_uncovered=False):
        from .onnx_graph import OnnxRunner
        self.runner = OnnxRunner(onnx_model, device=device, allow_uncovered=allow_uncovered)          # (refuses a graph outside onnx_graph.COVERED unless told otherwise)
        self.feat = HvxWhisperLogMel(128, device=device)
        self.device = torch.device(device)

    @torch.inference_mode()
    def __call__(self, speech_16k):
        speech = speech_16k.reshape(1, -1)
        if speech.shape[1] / 16000 > 30:
            raise AssertionError('do not support extract speech token for audio longer than 30s')          # (frontend.py:94)
        feat = self.feat(speech)                                                            # [1][128][T]
        names = self.runner.g.inputs
        feeds = {names[0]: feat}                                                            # (stays on the device)
        if len(names) > 1:
            import numpy as np
            feeds[names[1]] = np.array([feat.shape[2]], dtype=np.int32)
        out = self.runner.run(feeds)[self.runner.g.outputs[0]]
        tok = torch.tensor([out.reshape(-1).tolist()], dtype=torch.int32, device=self.device)
        return tok, torch.tensor([tok.shape[1]], dtype=torch.int32, device=self.device)


class HvxSpeakerEncoder:
    """`CosyVoiceFrontEnd._extract_spk_embedding` on the device (cosyvoice/cli/frontend.py:105-115): 16 kHz prompt audio -> kaldi 80-bin fbank with
    mean subtraction (HvxKaldiFbank) -> the CAM++ ONNX graph through the device executor -> embedding float32 [1][D] (D = 192 for campplus.onnx, an
    asset that is not in the tree: parity with onnxruntime on it is unpinned)."""

    def __init__(self, onnx_model, device='cuda', allow_uncovered=False):
        from .onnx_graph import OnnxRunner
        self.runner = OnnxRunner(onnx_model, device=device, allow_uncovered=allow_uncovered)
        self.feat = HvxKaldiFbank(80, 16000, subtract_mean=True, device=device)
        self.device = torch.device(device)

    @torch.inference_mode()
    def __call__(self, speech_16k):
        feat = self.feat(speech_16k.reshape(1, -1))                                         # [frames][80]
        out = self.runner.run({self.runner.g.inputs[0]: feat.unsqueeze(0)})[self.runner.g.outputs[0]]
        return torch.tensor([out.reshape(-1).tolist()], dtype=torch.float32, device=self.device)
