"""HvxHift — drop-in for `CausalHiFTGenerator` inference on MI355X (fp32, like the reference).

Mirrors server/model_utils/cosyvoice/hifigan/generator.py:713-726:
    inference(speech_feat, finalize=True) -> (wav float32 (1, 480*T), source (1, 1, 480*T))
The reference moves the F0 predictor to the CPU ("precision is crucial", :715-717); here it runs on the device in fp32
on the exact-fp32 MFMA path.  The fixed noise tables of the causal generator (`rand_ini`, `sine_waves`, `uv`; plain
attributes drawn from the global RNG at construction, generator.py:223-226, 355-356, absent from hift.pt) are explicit:
`tables=` or a seeded default.  All arithmetic runs in libhvx (csrc/hvx_hift.hip, csrc/hift_ops.hip).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .config import HiftConfig
from .packing import conv_weight, fold_weight_norm, pad_to
from .weights import hift_spec, check_state, DROP_KEYS, hift_source_down_rates


def make_tables(cfg: HiftConfig, seed=0, n_samples=None):
    """uniform [0,1) tables shaped like the reference's: rand_ini (1,H), sine_waves (1,N,H), uv (1,N,1)."""
    n = cfg.noise_seconds * cfg.sampling_rate if n_samples is None else n_samples
    g = torch.Generator()
    g.manual_seed(seed)
    h = cfg.nb_harmonics + 1
    rand_ini = torch.rand(1, h, generator=g)
    rand_ini[:, 0] = 0
    sine_waves = torch.rand(1, n, h, generator=g)
    uv = torch.rand(1, n, 1, generator=g)
    return dict(rand_ini=rand_ini, sine_waves=sine_waves, uv=uv)


def pack_hift_weights(sd, cfg: HiftConfig, device):
    """Device tensors in the order csrc/hvx_hift.hip consumes them (all fp32, weight-norm folded, kernels [Cout][tap][Cin_pad32])."""
    out = []

    def conv(name, wn=True):
        w = fold_weight_norm(sd, name) if wn else sd[name + '.weight'].float()
        out.append(conv_weight(w.to(device)).contiguous())
        out.append(sd[name + '.bias'].float().to(device).contiguous())

    def resblock(pre, n):
        for d in range(n):
            conv('%sconvs1.%d' % (pre, d))
            conv('%sconvs2.%d' % (pre, d))
            out.append(sd['%sactivations1.%d.alpha' % (pre, d)].float().to(device).contiguous())
            out.append(sd['%sactivations2.%d.alpha' % (pre, d)].float().to(device).contiguous())

    for i in (0, 2, 4, 6, 8):
        conv('f0_predictor.condnet.%d' % i)
    out.append(pad_to(sd['f0_predictor.classifier.weight'].float().to(device), 1, 32).contiguous())
    out.append(sd['f0_predictor.classifier.bias'].float().to(device).contiguous())
    out.append(sd['m_source.l_linear.weight'].float().to(device).reshape(-1).contiguous())
    out.append(sd['m_source.l_linear.bias'].float().to(device).contiguous())
    conv('conv_pre')
    nk = len(cfg.resblock_kernel_sizes)
    for i in range(len(cfg.upsample_rates)):
        conv('ups.%d' % i)
        conv('source_downs.%d' % i, wn=False)
        resblock('source_resblocks.%d.' % i, len(cfg.source_resblock_dilations[i]))
        for j in range(nk):
            resblock('resblocks.%d.' % (i * nk + j), len(cfg.resblock_dilations[j]))
    conv('conv_post')
    return out


class HvxHift:
    def __init__(self, cfg: HiftConfig, state_dict=None, device='cuda', tables=None, table_seed=0, exact_fp32=False):
        """exact_fp32: the decode convolutions on the exact fp32 MFMA forms (hvx_hift_config.exact_fp32: the reference's fp32 vocoder arithmetic, 2.8e-5 of it at
        5632 frames) instead of 3 bf16 MFMAs on (hi, lo) operand pairs (1.5e-4; the default, 2.2x faster).  The F0 predictor is exact fp32 either way."""
        _lib.require_gpu()
        self.exact_fp32 = bool(exact_fp32)
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        self.sampling_rate = cfg.sampling_rate
        self.up = cfg.upsample_total
        t = tables if tables is not None else make_tables(cfg, seed=table_seed)
        self.rand_ini = t['rand_ini']
        self.sine_table = t['sine_waves'][0].to(self.device, torch.float32).contiguous()        # [N][H]
        self._h = None
        self._ws = None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, sd, strict=True):
        return self.load_packed(self.pack_state_dict(sd))

    def pack_state_dict(self, sd):
        sd = {k: v for k, v in sd.items() if k not in DROP_KEYS}
        check_state(sd, hift_spec(self.cfg), 'CausalHiFTGenerator')
        return pack_hift_weights(sd, self.cfg, self.device)

    def load_packed(self, ws):
        c = self.cfg
        if any(len(d) != 3 for d in c.resblock_dilations + c.source_resblock_dilations):
            raise _lib.HvxError('hvx_hift supports three dilations per ResBlock')
        ws = [w.to(self.device) for w in ws]
        self._weights = ws
        cc = _lib.HiftConfig()
        cc.mel, cc.base_channels, cc.nb_harmonics, cc.f0_channels = c.mel, c.base_channels, c.nb_harmonics, c.f0_channels
        cc.n_up = len(c.upsample_rates)
        cc.n_rb = len(c.resblock_kernel_sizes)
        for i, (u, k) in enumerate(zip(c.upsample_rates, c.upsample_kernel_sizes)):
            cc.up_rates[i], cc.up_kernels[i] = u, k
        for j, k in enumerate(c.resblock_kernel_sizes):
            cc.rb_kernels[j] = k
            for d in range(3):
                cc.rb_dils[j][d] = c.resblock_dilations[j][d]
        for i, k in enumerate(c.source_resblock_kernel_sizes):
            cc.src_rb_kernels[i] = k
            for d in range(3):
                cc.src_rb_dils[i][d] = c.source_resblock_dilations[i][d]
        cc.n_fft, cc.hop, cc.conv_pre_kernel, cc.conv_post_kernel = c.n_fft, c.hop, c.conv_pre_look_right + 1, 7
        cc.sampling_rate, cc.nsf_alpha, cc.nsf_sigma = c.sampling_rate, c.nsf_alpha, c.nsf_sigma
        cc.voiced_threshold, cc.lrelu_slope, cc.audio_limit = c.nsf_voiced_threshold, c.lrelu_slope, c.audio_limit
        cc.exact_fp32 = 1 if self.exact_fp32 else 0
        assert hift_source_down_rates(c)[-1] == 1
        if self._h is not None:
            self.lib.hvx_hift_destroy(self._h)
        h = C.c_void_p()
        check(self.lib.hvx_hift_create(C.byref(cc), _lib.ptr_array(ws), len(ws), C.byref(h)), 'hvx_hift_create')
        self._h = h
        # (hi, lo) bf16 plane pairs of the decode convolutions' weights (every 2-D tensor behind the F0 predictor / source linear / conv_pre,
        # except the source down-sampling convolutions, whose input is the fp32 source STFT): hvx_hift_set_weight_planes
        skip = {14}                                                           # conv_pre (its input is the fp32 mel)
        i = 16
        for r in range(cc.n_up):
            skip.add(i + 2)                                                   # source_downs.r
            i += 4 + 18 + cc.n_rb * 18
        self._planes = []
        for k, w in enumerate(ws):
            if k >= 14 and w.dim() == 2 and k not in skip:
                hi = w.to(torch.bfloat16)
                lo = (w - hi.float()).to(torch.bfloat16)
                self._planes.append(torch.stack([hi, lo]).contiguous())
            else:
                self._planes.append(None)
        arr = (C.c_void_p * len(ws))(*[None if p is None else p.data_ptr() for p in self._planes])
        check(self.lib.hvx_hift_set_weight_planes(self._h, arr, len(ws)), 'hvx_hift_set_weight_planes')
        return self

    def eval(self):
        return self

    def cuda(self):
        return self

    def to(self, *a, **k):
        return self

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None:
                self.lib.hvx_hift_destroy(self._h)
        except Exception:
            pass

    def _workspace(self, t):
        need = self.lib.hvx_hift_workspace_bytes(self._h, t)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    # ---- stages (generator.py:716-721 / :672-711) ------------------------------------------------------------------------------
    def f0(self, mel):
        """mel (80, T) f32 -> f0 [T]"""
        mel = mel.to(self.device, torch.float32).contiguous()
        T = mel.shape[-1]
        out = torch.empty(T, dtype=torch.float32, device=self.device)
        ws = self._workspace(T)
        check(self.lib.hvx_hift_f0(self._h, stream_ptr(), ptr(ws), ws.numel(), ptr(mel), T, ptr(out)), 'hvx_hift_f0')
        return out

    def source(self, f0):
        """f0 [T] -> excitation [T*up]"""
        f0 = f0.to(self.device, torch.float32).contiguous()
        T = f0.numel()
        if T * self.up > self.sine_table.shape[0]:
            raise ValueError('utterance longer than the fixed noise tables (%d samples)' % self.sine_table.shape[0])
        out = torch.empty(T * self.up, dtype=torch.float32, device=self.device)
        ws = self._workspace(T)
        check(self.lib.hvx_hift_source(self._h, stream_ptr(), ptr(ws), ws.numel(), ptr(f0), T, ptr(self.sine_table), ptr(out)), 'hvx_hift_source')
        return out

    def decode(self, mel, source):
        """mel (80, T), source [T*up] -> wav [T*up]"""
        mel = mel.to(self.device, torch.float32).contiguous()
        source = source.to(self.device, torch.float32).contiguous().view(-1)
        T = mel.shape[-1]
        out = torch.empty(T * self.up, dtype=torch.float32, device=self.device)
        ws = self._workspace(T)
        check(self.lib.hvx_hift_decode(self._h, stream_ptr(), ptr(ws), ws.numel(), ptr(mel), ptr(source), T, ptr(out)), 'hvx_hift_decode')
        return out

    def decode_chunk(self, mel, source):
        """finalize=False (generator.py:672-711): the last conv_pre_look_right frames of mel (80, T_in) are look-ahead context;
        source [T_in*up] -> wav [(T_in - look_right - 1) * up] (the last up samples are dropped, generator.py:708-709)"""
        mel = mel.to(self.device, torch.float32).contiguous()
        source = source.to(self.device, torch.float32).contiguous().view(-1)
        T_in, look = mel.shape[-1], self.cfg.conv_pre_look_right
        if T_in - look - 1 <= 0:
            raise ValueError('a non-final chunk needs more than %d mel frames' % (look + 1 + self.f0_look_right))
        out = torch.empty((T_in - look) * self.up, dtype=torch.float32, device=self.device)
        ws = self._workspace(T_in)
        check(self.lib.hvx_hift_decode_chunk(self._h, stream_ptr(), ptr(ws), ws.numel(), ptr(mel), ptr(source), T_in, look, ptr(out)),
              'hvx_hift_decode_chunk')
        return out[:(T_in - look - 1) * self.up]

    # the F0 predictor's first conv looks (kernel 4, causal_type 'right') 3 frames ahead: convolution.py:172, f0_predictor.py:97-100
    f0_look_right = 3

    @torch.inference_mode()
    def inference(self, speech_feat, finalize=True):
        assert speech_feat.shape[0] == 1
        mel = speech_feat[0]
        f0 = self.f0(mel)
        if finalize:
            s = self.source(f0)
            wav = self.decode(mel, s)
        else:
            # f0_predictor.py:97-100: the last 3 frames are real right context, so the f0 of the frames before them is what the full
            # (zero-padded) pass gives for them; source and decode then run on mel[:, :-3] (generator.py:717-725)
            n = mel.shape[-1] - self.f0_look_right
            if n <= 0:
                raise ValueError('a non-final chunk needs more than %d mel frames' % self.f0_look_right)
            s = self.source(f0[:n])
            wav = self.decode_chunk(mel[:, :n], s)
        return wav.unsqueeze(0), s.view(1, 1, -1)
