"""Matcha-TTS family of the path behind the reference's own call signatures:

  HvxMatchaDecoder(x, mask, mu, t, spks=None, cond=None)      matcha/models/components/decoder.py:363 Decoder.forward
                                                             (cosyvoice/flow/decoder.py:210 ConditionalDecoder.forward with cv_variant)
  HvxMatchaCFM.forward(mu, mask, n_timesteps, temperature, spks, cond)   matcha/models/components/flow_matching.py:32 BASECFM.forward
  HvxHifiGan(mel)                                            matcha/hifigan/models.py:181 Generator.forward
  HvxDenoiser(vocoder)(audio, strength)                      matcha/hifigan/denoiser.py:57 Denoiser.forward

Host code only re-lays weights out at load time and hands device pointers to libhvx (fp32 throughout, as the reference runs this
family); there is no CPU fallback.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .config import HifiGanConfig, MatchaConfig
from .packing import conv_weight, convtranspose_phases, linear_weight, stft_bases
from .weights import DROP_KEYS, check_state, hifigan_spec, matcha_spec


def _fold_wn_old(sd, name):
    """torch.nn.utils.weight_norm (weight_g / weight_v, norm over all dims but 0) or an already folded `.weight`"""
    if name + '.weight_g' in sd:
        v, g = sd[name + '.weight_v'].float(), sd[name + '.weight_g'].float()
        return v * (g / v.norm(2, dim=(1, 2), keepdim=True))
    return sd[name + '.weight'].float()


def matcha_euler_schedule(n_timesteps):
    """fp32 (t, dt) pairs visited by BASECFM.solve_euler for t_span = linspace(0, 1, n + 1) (flow_matching.py:51, 67-83)"""
    t_span = torch.linspace(0, 1, n_timesteps + 1)
    t, dt = t_span[0], t_span[1] - t_span[0]
    ts, dts = [], []
    for step in range(1, len(t_span)):
        ts.append(float(t))
        dts.append(float(dt))
        t = t + dt
        if step < len(t_span) - 1:
            dt = t_span[step + 1] - t
    return ts, dts


class HvxMatchaDecoder:
    def __init__(self, cfg: MatchaConfig, state_dict, device='cuda', prefix=''):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        self._h = None
        self._ws = None
        self.load_state_dict(state_dict, prefix=prefix)

    def load_state_dict(self, sd, prefix=''):
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix) and k not in DROP_KEYS}
        c = self.cfg
        check_state(sd, matcha_spec(c), 'Decoder')
        dev = self.device

        def f(t):
            return t.to(dev, torch.float32).contiguous()

        def W(k):
            return sd[k].float()

        ws = [f(linear_weight(W('time_mlp.linear_1.weight'))), f(W('time_mlp.linear_1.bias')),
              f(W('time_mlp.linear_2.weight')), f(W('time_mlp.linear_2.bias'))]

        def resnet(p):
            return [f(W(p + 'mlp.1.weight')), f(W(p + 'mlp.1.bias')),
                    f(conv_weight(W(p + 'block1.block.0.weight'))), f(W(p + 'block1.block.0.bias')), f(W(p + 'block1.block.1.weight')), f(W(p + 'block1.block.1.bias')),
                    f(conv_weight(W(p + 'block2.block.0.weight'))), f(W(p + 'block2.block.0.bias')), f(W(p + 'block2.block.1.weight')), f(W(p + 'block2.block.1.bias')),
                    f(conv_weight(W(p + 'res_conv.weight'))), f(W(p + 'res_conv.bias'))]

        def tblock(p):
            wqkv = torch.cat([W(p + 'attn1.to_q.weight'), W(p + 'attn1.to_k.weight'), W(p + 'attn1.to_v.weight')], 0)
            snake = torch.cat([torch.exp(W(p + 'ff.net.0.alpha')), torch.exp(W(p + 'ff.net.0.beta'))], 0)       # alpha_logscale (transformer.py:69-71)
            return [f(W(p + 'norm1.weight') - 1.0), f(W(p + 'norm1.bias')), f(wqkv), f(W(p + 'attn1.to_out.0.weight')), f(W(p + 'attn1.to_out.0.bias')),
                    f(W(p + 'norm3.weight') - 1.0), f(W(p + 'norm3.bias')), f(W(p + 'ff.net.0.proj.weight')), f(W(p + 'ff.net.0.proj.bias')), f(snake),
                    f(W(p + 'ff.net.2.weight')), f(W(p + 'ff.net.2.bias'))]

        n_st = len(c.channels)
        for i in range(n_st):
            p = 'down_blocks.%d.' % i
            ws += resnet(p + '0.')
            for j in range(c.n_blocks):
                ws += tblock(p + '1.%d.' % j)
            last = i == n_st - 1
            ws += [f(conv_weight(W(p + ('2.weight' if last else '2.conv.weight')))), f(W(p + ('2.bias' if last else '2.conv.bias')))]
        for i in range(c.num_mid_blocks):
            p = 'mid_blocks.%d.' % i
            ws += resnet(p + '0.')
            for j in range(c.n_blocks):
                ws += tblock(p + '1.%d.' % j)
        for i in range(n_st):
            p = 'up_blocks.%d.' % i
            ws += resnet(p + '0.')
            for j in range(c.n_blocks):
                ws += tblock(p + '1.%d.' % j)
            if i == n_st - 1:
                ws += [f(conv_weight(W(p + '2.weight'))), f(W(p + '2.bias'))]
            else:
                ws += [f(convtranspose_phases(W(p + '2.conv.weight'), 2, 1)), f(W(p + '2.conv.bias'))]
        ws += [f(conv_weight(W('final_block.block.0.weight'))), f(W('final_block.block.0.bias')), f(W('final_block.block.1.weight')), f(W('final_block.block.1.bias')),
               f(conv_weight(W('final_proj.weight'))), f(W('final_proj.bias'))]
        self._weights = ws
        cc = _lib.MatchaConfig(in_channels=c.in_channels, out_channels=c.mel, n_stages=n_st, n_blocks=c.n_blocks, n_mid=c.num_mid_blocks,
                               heads=c.num_heads, ff_mult=c.ff_mult, cv_variant=int(c.cv_variant), max_t=c.max_t)
        if c.head_dim != 64:
            raise _lib.HvxError('hvx kernels are specialised for head_dim 64')
        for i, ch in enumerate(c.channels):
            cc.channels[i] = ch
        if self._h is not None:
            self.lib.hvx_matcha_destroy(self._h)
        h = C.c_void_p()
        check(self.lib.hvx_matcha_create(C.byref(cc), _lib.ptr_array(ws), len(ws), C.byref(h)), 'hvx_matcha_create')
        self._h = h
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None:
                self.lib.hvx_matcha_destroy(self._h)
        except Exception:
            pass

    def _workspace(self, B, T):
        need = self.lib.hvx_matcha_workspace_bytes(self._h, B, T)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _lens(self, mask, B, T):
        """valid lengths per U-Net level from a (B, 1, T) 0/1 mask: masks.append(mask[:, :, ::2]) (decoder.py:406); None when full"""
        if mask is None:
            return None
        m = mask.to(self.device).reshape(B, T).ne(0)
        n = m.sum(dim=1).to(torch.int32)
        if bool((n == T).all()):                   # (one host read per call site; the solver reads it once for all steps)
            return None
        rows = []
        for _ in range(len(self.cfg.channels)):
            rows.append(n.clone())
            n = (n + 1) // 2
        return torch.stack(rows, 0).contiguous()

    def _prep(self, x, mu, spks, cond):
        f = lambda a: None if a is None else a.to(self.device, torch.float32).contiguous()
        x, mu, spks, cond = f(x), f(mu), f(spks), f(cond)
        c = self.cfg
        if (spks is None) != (c.spk_dim == 0) or (cond is None) != (not c.use_cond):
            raise ValueError('decoder built for spk_dim=%d, cond=%s' % (c.spk_dim, c.use_cond))
        return x, mu, spks, cond

    @torch.inference_mode()
    def __call__(self, x, mask, mu, t, spks=None, cond=None):
        """Decoder.forward: x, mu, cond (B, mel, T); mask (B, 1, T); t (B,) or scalar; spks (B, spk_dim) -> (B, mel, T)"""
        B, mel, T = x.shape
        x, mu, spks, cond = self._prep(x, mu, spks, cond)
        t = torch.as_tensor(t, dtype=torch.float32).reshape(-1).to(self.device)
        if t.numel() == 1 and B > 1:
            t = t.repeat(B)
        lens = self._lens(mask, B, T)
        out = torch.empty(B, mel, T, dtype=torch.float32, device=self.device)
        ws = self._workspace(B, T)
        check(self.lib.hvx_matcha_estimator(self._h, stream_ptr(), ptr(ws), ws.numel(), B, T, ptr(x), ptr(mu), ptr(spks), self.cfg.spk_dim, ptr(cond),
                                            ptr(lens), ptr(t.contiguous()), ptr(out)), 'hvx_matcha_estimator')
        return out

    forward = __call__

    @torch.inference_mode()
    def solve(self, z, mask, mu, n_timesteps, spks=None, cond=None):
        """BASECFM.solve_euler with t_span = linspace(0, 1, n + 1): z (B, mel, T) -> sample"""
        B, mel, T = z.shape
        z, mu, spks, cond = self._prep(z, mu, spks, cond)
        x = z.clone()
        lens = self._lens(mask, B, T)
        ts, dts = matcha_euler_schedule(n_timesteps)
        ws = self._workspace(B, T)
        check(self.lib.hvx_matcha_solve(self._h, stream_ptr(), ptr(ws), ws.numel(), B, T, ptr(x), ptr(mu), ptr(spks), self.cfg.spk_dim, ptr(cond), ptr(lens),
                                        n_timesteps, (C.c_float * n_timesteps)(*ts), (C.c_float * n_timesteps)(*dts)), 'hvx_matcha_solve')
        return x


class HvxMatchaCFM:
    """CFM(BASECFM) (flow_matching.py:120-132): `.estimator` + `.forward`; checkpoint keys carry the `estimator.` prefix."""

    def __init__(self, cfg: MatchaConfig, state_dict, device='cuda'):
        prefix = 'estimator.' if any(k.startswith('estimator.') for k in state_dict) else ''
        self.cfg = cfg
        self.estimator = HvxMatchaDecoder(cfg, state_dict, device=device, prefix=prefix)
        self.device = self.estimator.device

    @torch.inference_mode()
    def forward(self, mu, mask, n_timesteps, temperature=1.0, spks=None, cond=None, noise=None):
        """z = randn_like(mu) * temperature (drawn here from torch's global generator on the device of `mu`, as the reference does,
        unless `noise` is given), then solve_euler"""
        if noise is None:
            noise = torch.randn_like(mu)
        z = noise.to(self.device, torch.float32) * temperature
        return self.estimator.solve(z, mask, mu, n_timesteps, spks=spks, cond=cond)

    __call__ = forward


class HvxHifiGan:
    def __init__(self, cfg: HifiGanConfig, state_dict, device='cuda', exact_fp32=False):
        _lib.require_gpu()
        self.exact_fp32 = bool(exact_fp32)          # hvx_hifigan_config.exact_fp32: exact fp32 MFMA convolutions instead of split-bf16 pairs
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        self._h = None
        self._ws = None
        self.load_state_dict(state_dict)

    def load_state_dict(self, sd):
        sd = {k: v for k, v in sd.items() if k not in DROP_KEYS}
        c = self.cfg
        folded = not any(k.endswith('weight_g') for k in sd)
        if not folded:
            check_state(sd, hifigan_spec(c), 'Generator')
        dev = self.device

        def f(t):
            return t.to(dev, torch.float32).contiguous()

        ws = [f(conv_weight(_fold_wn_old(sd, 'conv_pre'))), f(sd['conv_pre.bias'])]
        nk = len(c.resblock_kernel_sizes)
        for i, (u, k) in enumerate(zip(c.upsample_rates, c.upsample_kernel_sizes)):
            ws += [f(convtranspose_phases(_fold_wn_old(sd, 'ups.%d' % i), u, (k - u) // 2)), f(sd['ups.%d.bias' % i])]
            for j in range(nk):
                p = 'resblocks.%d.' % (i * nk + j)
                for d in range(3):
                    ws += [f(conv_weight(_fold_wn_old(sd, p + 'convs1.%d' % d))), f(sd[p + 'convs1.%d.bias' % d]),
                           f(conv_weight(_fold_wn_old(sd, p + 'convs2.%d' % d))), f(sd[p + 'convs2.%d.bias' % d])]
        ws += [f(conv_weight(_fold_wn_old(sd, 'conv_post'))), f(sd['conv_post.bias'])]
        self._weights = ws
        cc = _lib.HifiGanConfig(mel=c.mel, initial_channel=c.initial_channel, n_up=len(c.upsample_rates), n_rb=nk, exact_fp32=1 if self.exact_fp32 else 0)
        for i, (u, k) in enumerate(zip(c.upsample_rates, c.upsample_kernel_sizes)):
            cc.up_rates[i], cc.up_kernels[i] = u, k
        for j, k in enumerate(c.resblock_kernel_sizes):
            cc.rb_kernels[j] = k
            for d in range(3):
                cc.rb_dils[j][d] = c.resblock_dilations[j][d]
        if self._h is not None:
            self.lib.hvx_hifigan_destroy(self._h)
        h = C.c_void_p()
        check(self.lib.hvx_hifigan_create(C.byref(cc), _lib.ptr_array(ws), len(ws), C.byref(h)), 'hvx_hifigan_create')
        self._h = h
        return self

    def eval(self):
        return self

    def remove_weight_norm(self):          # weight norm is folded at load
        return None

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None:
                self.lib.hvx_hifigan_destroy(self._h)
        except Exception:
            pass

    @torch.inference_mode()
    def __call__(self, mel):
        """(B, mel, T) -> (B, 1, T * upsample)"""
        mel = mel.to(self.device, torch.float32).contiguous()
        B, _, T = mel.shape
        L = T * self.cfg.upsample
        out = torch.empty(B, 1, L, dtype=torch.float32, device=self.device)
        need = self.lib.hvx_hifigan_workspace_bytes(self._h, T)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        for b in range(B):
            check(self.lib.hvx_hifigan_forward(self._h, stream_ptr(), ptr(self._ws), self._ws.numel(), ptr(mel[b]), T, ptr(out[b])), 'hvx_hifigan_forward')
        return out

    forward = __call__


class HvxDenoiser:
    """Denoiser(vocoder, filter_length, n_overlap, mode) (denoiser.py:10-55): bias spectrum = |STFT| of the vocoder's answer to a probe mel of 88 frames —
    all zeros (mode 'zeros', the default) or N(0, 1) (mode 'normal', denoiser.py:20-21).  The 'normal' probe is drawn the way the reference draws it when its
    vocoder lives on the CPU — `torch.randn((1, 80, 88))` from the GLOBAL CPU generator (or from `generator` when one is passed) — so that a seeded run gives the
    reference's bias spectrum; with the vocoder on a GPU the reference would draw from that device's generator, whose stream nothing else can reproduce."""

    def __init__(self, vocoder: HvxHifiGan, filter_length=None, n_overlap=None, mode='zeros', generator=None):
        if mode not in ('zeros', 'normal'):
            raise Exception(f"Mode {mode} if not supported")          # (the reference's own message, denoiser.py:23)
        self.lib = vocoder.lib
        self.device = vocoder.device
        c = vocoder.cfg
        self.n_fft = filter_length or c.n_fft
        self.hop = self.n_fft // (n_overlap or c.n_overlap)
        ana, syn, wsq = stft_bases(self.n_fft)
        self._ana, self._syn, self._wsq = ana.to(self.device), syn.to(self.device), wsq.to(self.device)
        self._ws = None
        bins = self.n_fft // 2 + 1
        mel_input = torch.zeros(1, c.mel, 88) if mode == 'zeros' else torch.randn((1, c.mel, 88), generator=generator)
        probe = vocoder(mel_input.to(self.device)).reshape(-1)
        self.bias_spec = self._magnitude_first_frame(probe, bins)

    def _workspace(self, L):
        need = self.lib.hvx_denoise_workspace_bytes(L, self.n_fft, self.hop)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _magnitude_first_frame(self, audio, bins):
        """bias_spec[:, :, 0] (denoiser.py:53-55): |STFT| of the probe, first frame"""
        L = audio.numel()
        frames = 1 + L // self.hop
        mag = torch.empty(frames, bins, dtype=torch.float32, device=self.device)
        ws = self._workspace(L)
        check(self.lib.hvx_stft_magnitude(stream_ptr(), ptr(ws), ws.numel(), ptr(audio.contiguous()), L, self.n_fft, self.hop, ptr(self._ana), ptr(mag)),
              'hvx_stft_magnitude')
        return mag[0].contiguous()

    @torch.inference_mode()
    def __call__(self, audio, strength=0.0005):
        """(B, L) -> (B, hop * (L // hop))"""
        audio = audio.to(self.device, torch.float32).contiguous()
        B, L = audio.shape
        frames = 1 + L // self.hop
        out = torch.empty(B, self.hop * (frames - 1), dtype=torch.float32, device=self.device)
        ws = self._workspace(L)
        for b in range(B):
            check(self.lib.hvx_denoise(stream_ptr(), ptr(ws), ws.numel(), ptr(audio[b]), L, self.n_fft, self.hop, ptr(self._ana), ptr(self._syn),
                                       ptr(self._wsq), ptr(self.bias_spec), float(strength), ptr(out[b])), 'hvx_denoise')
        return out

    forward = __call__
