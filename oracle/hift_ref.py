"""CPU oracle for the causal HiFT vocoder (TEST INFRASTRUCTURE — see oracle/__init__.py).

Follows server/model_utils/cosyvoice/hifigan/generator.py
  * CausalHiFTGenerator.inference / decode   :713-726 / :672-711
  * SourceModuleHnNSF.forward                :358-375
  * SineGen2.forward / _f02sine              :289-317 / :233-287   (causal, eval: fixed rand_ini / noise tables)
  * _stft / _istft                           :491-505
  * ResBlock.forward                         :110-117
and cosyvoice/hifigan/f0_predictor.py:95-103 (CausalConvRNNF0Predictor),
cosyvoice/transformer/convolution.py:150-258 (CausalConv1d / DownSample / Upsample),
cosyvoice/transformer/activation.py:73-84 (Snake).
Weights: flat dict with the reference's hift.pt keys; weight-norm parametrisations
(`...parametrizations.weight.original0/1`, generator.py:26-29) are folded as w = g * v / ||v||.
The fixed noise tables (rand_ini, sine_waves, uv — plain attributes drawn at construction, not in
any state_dict) are explicit inputs (`tables` dict).
"""
import numpy as np
import torch
import torch.nn.functional as F


def fold_weight_norm(sd, name):
    """weight of a (possibly weight-normed) conv: original0 = g (Cout,1,1), original1 = v."""
    k0 = name + '.parametrizations.weight.original0'
    if k0 in sd:
        g = sd[k0]
        v = sd[name + '.parametrizations.weight.original1']
        return v * (g / v.norm(2, dim=(1, 2), keepdim=True))     # torch._weight_norm(v, g, dim=0)
    return sd[name + '.weight']


def causal_padding(k, dilation=1):
    return int((k * dilation - dilation) / 2) * 2 + (k + 1) % 2


def causal_conv(x, w, b, dilation=1, causal_type='left', cache=None):
    """CausalConv1d (convolution.py:150-187); `cache` replaces the zero padding (left or right context)."""
    pad = causal_padding(w.shape[-1], dilation)
    if cache is None:
        cache = torch.zeros(x.shape[0], x.shape[1], pad, dtype=x.dtype)
    assert cache.shape[2] == pad
    x = torch.cat([cache, x], dim=2) if causal_type == 'left' else torch.cat([x, cache], dim=2)
    return F.conv1d(x, w, b, dilation=dilation)


def causal_conv_down(x, w, b, stride):
    """CausalConv1dDownSample (convolution.py:190-221)."""
    return F.conv1d(F.pad(x, (stride - 1, 0)), w, b, stride=stride)


def causal_conv_up(x, w, b, stride):
    """CausalConv1dUpsample (convolution.py:224-258): nearest upsample, left pad k-1, conv."""
    x = F.interpolate(x, scale_factor=float(stride), mode='nearest')
    return F.conv1d(F.pad(x, (w.shape[-1] - 1, 0)), w, b)


def snake(x, alpha):
    a = alpha[None, :, None]
    return x + (1.0 / (a + 1e-9)) * torch.sin(x * a) ** 2


def resblock(x, sd, pre, dilations):
    for i, d in enumerate(dilations):
        xt = snake(x, sd['%sactivations1.%d.alpha' % (pre, i)])
        xt = causal_conv(xt, fold_weight_norm(sd, '%sconvs1.%d' % (pre, i)), sd['%sconvs1.%d.bias' % (pre, i)], dilation=d)
        xt = snake(xt, sd['%sactivations2.%d.alpha' % (pre, i)])
        xt = causal_conv(xt, fold_weight_norm(sd, '%sconvs2.%d' % (pre, i)), sd['%sconvs2.%d.bias' % (pre, i)], dilation=1)
        x = xt + x
    return x


def f0_predictor(mel, sd, pre='f0_predictor.', finalize=True):
    """mel (1,80,T) -> f0 (1,T), or (1,T-3) with finalize=False where the last 3 frames are right context (f0_predictor.py:95-103)."""
    w0 = fold_weight_norm(sd, pre + 'condnet.0')
    if finalize:
        x = causal_conv(mel, w0, sd[pre + 'condnet.0.bias'], causal_type='right')
    else:
        p = causal_padding(w0.shape[-1])
        x = causal_conv(mel[:, :, :-p], w0, sd[pre + 'condnet.0.bias'], causal_type='right', cache=mel[:, :, -p:])
    x = F.elu(x)
    for i in (2, 4, 6, 8):
        x = F.elu(causal_conv(x, fold_weight_norm(sd, pre + 'condnet.%d' % i), sd[pre + 'condnet.%d.bias' % i]))
    x = x.transpose(1, 2)
    return torch.abs(F.linear(x, sd[pre + 'classifier.weight'], sd[pre + 'classifier.bias']).squeeze(-1))


def make_tables(cfg, seed=0, n_samples=None):
    """Fixed noise tables of the causal generator (generator.py:223-226, 355-356): uniform [0,1)."""
    n = cfg.noise_seconds * cfg.sampling_rate if n_samples is None else n_samples
    g = torch.Generator()
    g.manual_seed(seed)
    h = cfg.nb_harmonics + 1
    rand_ini = torch.rand(1, h, generator=g)
    rand_ini[:, 0] = 0
    sine_waves = torch.rand(1, n, h, generator=g)
    uv = torch.rand(1, n, 1, generator=g)
    return dict(rand_ini=rand_ini, sine_waves=sine_waves, uv=uv)


def sine_gen2(f0, cfg, tables):
    """SineGen2.forward, causal + eval (generator.py:233-317). f0 (1,L,1) at sample rate -> (sine (1,L,9), uv)."""
    H = cfg.nb_harmonics + 1
    up = cfg.upsample_total
    fn = f0 * torch.arange(1, H + 1, dtype=torch.float32)[None, None, :]
    rad = (fn / cfg.sampling_rate) % 1
    rad[:, 0, :] = rad[:, 0, :] + tables['rand_ini']
    rad = F.interpolate(rad.transpose(1, 2), scale_factor=1 / up, mode='linear').transpose(1, 2)
    phase = torch.cumsum(rad, dim=1) * 2 * np.pi
    phase = F.interpolate(phase.transpose(1, 2) * up, scale_factor=float(up), mode='nearest').transpose(1, 2)
    sines = torch.sin(phase) * cfg.nsf_alpha
    uv = (f0 > cfg.nsf_voiced_threshold).float()
    noise_amp = uv * cfg.nsf_sigma + (1 - uv) * cfg.nsf_alpha / 3
    noise = noise_amp * tables['sine_waves'][:, :sines.shape[1]]
    return sines * uv + noise, uv


def source_module(f0_up, sd, cfg, tables):
    """SourceModuleHnNSF.forward (generator.py:358-375) -> sine_merge (1,L,1)."""
    sine_wavs, uv = sine_gen2(f0_up, cfg, tables)
    return torch.tanh(F.linear(sine_wavs, sd['m_source.l_linear.weight'], sd['m_source.l_linear.bias']))


def hann_window(n_fft):
    # scipy.signal.get_window("hann", n, fftbins=True) == periodic Hann
    return torch.hann_window(n_fft, periodic=True, dtype=torch.float32)


def stft(x, cfg):
    spec = torch.stft(x, cfg.n_fft, cfg.hop, cfg.n_fft, window=hann_window(cfg.n_fft), return_complex=True)
    spec = torch.view_as_real(spec)
    return spec[..., 0], spec[..., 1]


def istft(mag, phase, cfg):
    mag = torch.clip(mag, max=1e2)
    real = mag * torch.cos(phase)
    img = mag * torch.sin(phase)
    return torch.istft(torch.complex(real, img), cfg.n_fft, cfg.hop, cfg.n_fft, window=hann_window(cfg.n_fft))


def decode(mel, s, sd, cfg, taps=None, finalize=True):
    """CausalHiFTGenerator.decode (generator.py:672-711). mel (1,80,T), s (1,1,480T); finalize=False: the last conv_pre_look_right
    frames are context, the source spectrogram is cut to match and the last 480 samples are dropped."""
    sr, si = stft(s.squeeze(1), cfg)
    if finalize:
        x = causal_conv(mel, fold_weight_norm(sd, 'conv_pre'), sd['conv_pre.bias'], causal_type='right')
    else:
        look = cfg.conv_pre_look_right
        x = causal_conv(mel[:, :, :-look], fold_weight_norm(sd, 'conv_pre'), sd['conv_pre.bias'], causal_type='right', cache=mel[:, :, -look:])
        cut = int(np.prod(cfg.upsample_rates) * look)
        sr, si = sr[:, :, :-cut], si[:, :, :-cut]
    s_stft = torch.cat([sr, si], dim=1)
    nk = len(cfg.resblock_kernel_sizes)
    nu = len(cfg.upsample_rates)
    down_rates = [1] + cfg.upsample_rates[::-1][:-1]
    down_cum = list(np.cumprod(down_rates))[::-1]
    for i in range(nu):
        x = F.leaky_relu(x, cfg.lrelu_slope)
        x = causal_conv_up(x, fold_weight_norm(sd, 'ups.%d' % i), sd['ups.%d.bias' % i], cfg.upsample_rates[i])
        if i == nu - 1:
            x = F.pad(x, (1, 0), mode='reflect')
        u = int(down_cum[i])
        if u == 1:
            sx = causal_conv(s_stft, sd['source_downs.%d.weight' % i], sd['source_downs.%d.bias' % i])
        else:
            sx = causal_conv_down(s_stft, sd['source_downs.%d.weight' % i], sd['source_downs.%d.bias' % i], u)
        sx = resblock(sx, sd, 'source_resblocks.%d.' % i, cfg.source_resblock_dilations[i])
        x = x + sx
        xs = None
        for j in range(nk):
            r = resblock(x, sd, 'resblocks.%d.' % (i * nk + j), cfg.resblock_dilations[j])
            xs = r if xs is None else xs + r
        x = xs / nk
        if taps is not None:
            taps['stage%d' % i] = x.clone()
    x = F.leaky_relu(x)
    x = causal_conv(x, fold_weight_norm(sd, 'conv_post'), sd['conv_post.bias'])
    nb = cfg.n_fft // 2 + 1
    mag = torch.exp(x[:, :nb, :])
    phase = torch.sin(x[:, nb:, :])
    w = istft(mag, phase, cfg)
    if not finalize:
        w = w[:, :-int(np.prod(cfg.upsample_rates) * cfg.hop)]
    return torch.clamp(w, -cfg.audio_limit, cfg.audio_limit)


def hift_inference(mel, sd, cfg, tables, taps=None, finalize=True):
    """CausalHiFTGenerator.inference (generator.py:713-726). mel (1,80,T) fp32 -> (wav (1,480T), source); finalize=False is the
    non-final chunk of streaming synthesis: (wav (1, 480 (T - 8)), source (1, 1, 480 (T - 3)))."""
    f0 = f0_predictor(mel, sd, finalize=finalize)
    s = F.interpolate(f0[:, None], scale_factor=float(cfg.upsample_total), mode='nearest').transpose(1, 2)
    s = source_module(s, sd, cfg, tables).transpose(1, 2)
    if taps is not None:
        taps['f0'] = f0.clone()
        taps['source'] = s.clone()
    if not finalize:
        p = causal_padding(fold_weight_norm(sd, 'f0_predictor.condnet.0').shape[-1])
        return decode(mel[:, :, :-p], s, sd, cfg, taps, finalize=False), s
    return decode(mel, s, sd, cfg, taps), s
