"""CPU oracle (TEST INFRASTRUCTURE — never imported by the product path) for the Matcha-TTS family of SURVEY.md §8(a) M1-M5.

Functional fp32 PyTorch restatement, one function per reference function, weights in the reference's own state-dict keys:
  decoder_forward      matcha/models/components/decoder.py:363-443 (Decoder.forward) and, with cv=True,
                       server/model_utils/cosyvoice/flow/decoder.py:210-291 (ConditionalDecoder.forward)
  basic_transformer    matcha/models/components/transformer.py:243-316 + diffusers.models.attention_processor.Attention
                       (third-party, diffusers==0.25.0 per the reference requirements: restated — to_q/to_k/to_v without bias, to_out.0 with
                       bias, scale = dim_head ** -0.5, additive attention mask broadcast over heads; parity of this sub-step is pinned only
                       through the reference's call sites, see tests/golden/make_golden.py)
  snake_beta           matcha/models/components/transformer.py:17-80
  solve_euler          matcha/models/components/flow_matching.py:53-85
  generator_forward    matcha/hifigan/models.py:181-197, ResBlock1 :90-97
  denoise              matcha/hifigan/denoiser.py:57-64
"""
import math

import torch
import torch.nn.functional as F


def sinusoidal_pos_emb(t, dim, scale=1000):                       # decoder.py:12-29
    if t.ndim < 1:
        t = t.unsqueeze(0)
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half).float() * -emb)
    emb = scale * t.unsqueeze(1) * emb.unsqueeze(0)
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def _block1d(x, mask, sd, p):                                     # decoder.py:40-54
    h = F.conv1d(x * mask, sd[p + 'block.0.weight'], sd[p + 'block.0.bias'], padding=1)
    h = F.group_norm(h, 8, sd[p + 'block.1.weight'], sd[p + 'block.1.bias'], eps=1e-5)
    return F.mish(h) * mask


def _resnet(x, mask, temb, sd, p):                                # decoder.py:57-75
    h = _block1d(x, mask, sd, p + 'block1.')
    h = h + F.linear(F.mish(temb), sd[p + 'mlp.1.weight'], sd[p + 'mlp.1.bias']).unsqueeze(-1)
    h = _block1d(h, mask, sd, p + 'block2.')
    return h + F.conv1d(x * mask, sd[p + 'res_conv.weight'], sd[p + 'res_conv.bias'])


def snake_beta(x, sd, p):                                         # transformer.py:61-80 (alpha_logscale=True)
    x = F.linear(x, sd[p + 'proj.weight'], sd[p + 'proj.bias'])
    alpha, beta = torch.exp(sd[p + 'alpha']), torch.exp(sd[p + 'beta'])
    return x + (1.0 / (beta + 0.000000001)) * torch.pow(torch.sin(x * alpha), 2)


def _attention(x, bias, sd, p, heads):
    """diffusers Attention (self-attention): bias is an additive (B, 1 or T, T) score bias or None"""
    B, T, _ = x.shape
    q, k, v = F.linear(x, sd[p + 'to_q.weight']), F.linear(x, sd[p + 'to_k.weight']), F.linear(x, sd[p + 'to_v.weight'])
    d = q.shape[-1] // heads
    q, k, v = (t.view(B, T, heads, d).transpose(1, 2) for t in (q, k, v))
    s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    if bias is not None:
        s = s + bias[:, None]
    o = torch.matmul(torch.softmax(s, dim=-1), v).transpose(1, 2).reshape(B, T, heads * d)
    return F.linear(o, sd[p + 'to_out.0.weight'], sd[p + 'to_out.0.bias'])


def basic_transformer(x, bias, sd, p, heads):                     # transformer.py:279-316 (norm_type layer_norm, no cross attention)
    n = F.layer_norm(x, x.shape[-1:], sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], eps=1e-5)
    x = _attention(n, bias, sd, p + 'attn1.', heads) + x
    n = F.layer_norm(x, x.shape[-1:], sd[p + 'norm3.weight'], sd[p + 'norm3.bias'], eps=1e-5)
    ff = F.linear(snake_beta(n, sd, p + 'ff.net.0.'), sd[p + 'ff.net.2.weight'], sd[p + 'ff.net.2.bias'])
    return ff + x


def _attn_bias(mask, cv, T):
    """Matcha: the (B, T) float mask itself is handed to diffusers as `attention_mask` and ADDED to the scores (decoder.py:396-401);
    CosyVoice: mask_to_bias of the key-padding mask, (1 - m) * -1e10 (cosyvoice/flow/decoder.py:245-246, utils/mask.py)."""
    m = mask[:, 0]                                                # (B, T)
    if cv:
        return ((1.0 - m) * -1.0e10)[:, None, :].expand(-1, T, -1)
    return m[:, None, :]


def decoder_forward(sd, c, x, mask, mu, t, spks=None, cond=None):
    """-> (B, mel, T).  c: MatchaConfig"""
    cv = c.cv_variant
    temb = sinusoidal_pos_emb(t, c.in_channels)
    temb = F.linear(F.silu(F.linear(temb, sd['time_mlp.linear_1.weight'], sd['time_mlp.linear_1.bias'])),
                    sd['time_mlp.linear_2.weight'], sd['time_mlp.linear_2.bias'])
    x = torch.cat([x, mu], dim=1)
    if spks is not None:
        x = torch.cat([x, spks[:, :, None].expand(-1, -1, x.shape[-1])], dim=1)
    if cond is not None:
        x = torch.cat([x, cond], dim=1)
    hiddens, masks = [], [mask]
    n_st = len(c.channels)

    def tstack(x, m, p):
        x = x.transpose(1, 2)
        bias = _attn_bias(m, cv, x.shape[1])
        for j in range(c.n_blocks):
            x = basic_transformer(x, bias, sd, p + '%d.' % j, c.num_heads)
        return x.transpose(1, 2)

    for i in range(n_st):
        m = masks[-1]
        p = 'down_blocks.%d.' % i
        x = _resnet(x, m, temb, sd, p + '0.')
        x = tstack(x, m, p + '1.')
        hiddens.append(x)
        last = i == n_st - 1
        if last:
            x = F.conv1d(x * m, sd[p + '2.weight'], sd[p + '2.bias'], padding=1)
        else:
            x = F.conv1d(x * m, sd[p + '2.conv.weight'], sd[p + '2.conv.bias'], stride=2, padding=1)
        masks.append(m[:, :, ::2])
    masks = masks[:-1]
    m_mid = masks[-1]
    for i in range(c.num_mid_blocks):
        p = 'mid_blocks.%d.' % i
        x = _resnet(x, m_mid, temb, sd, p + '0.')
        x = tstack(x, m_mid, p + '1.')
    m_up = None
    for i in range(n_st):
        m_up = masks.pop()
        skip = hiddens.pop()
        p = 'up_blocks.%d.' % i
        x = torch.cat([x[:, :, :skip.shape[-1]], skip], dim=1)
        x = _resnet(x, m_up, temb, sd, p + '0.')
        x = tstack(x, m_up, p + '1.')
        last = i == n_st - 1
        if last:
            x = F.conv1d(x * m_up, sd[p + '2.weight'], sd[p + '2.bias'], padding=1)
        else:
            x = F.conv_transpose1d(x * m_up, sd[p + '2.conv.weight'], sd[p + '2.conv.bias'], stride=2, padding=1)
    x = _block1d(x, m_up, sd, 'final_block.')
    out = F.conv1d(x * m_up, sd['final_proj.weight'], sd['final_proj.bias'])
    return out * mask


def euler_schedule(n_timesteps):
    """the fp32 (t, dt) pairs BASECFM.solve_euler visits for t_span = linspace(0, 1, n + 1) (flow_matching.py:51, 67-83)"""
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


def solve_euler(sd, c, z, mask, mu, n_timesteps, spks=None, cond=None):
    x = z
    ts, dts = euler_schedule(n_timesteps)
    for t, dt in zip(ts, dts):
        tt = torch.full((x.shape[0],), t, dtype=torch.float32)
        x = x + dt * decoder_forward(sd, c, x, mask, mu, tt, spks, cond)
    return x


def _wn(sd, name):
    if name + '.weight_g' in sd:
        v, g = sd[name + '.weight_v'].float(), sd[name + '.weight_g'].float()
        return v * (g / v.norm(2, dim=(1, 2), keepdim=True))
    return sd[name + '.weight'].float()


def generator_forward(sd, c, mel):
    """(B, mel, T) -> (B, 1, T * prod(upsample_rates))"""
    x = F.conv1d(mel, _wn(sd, 'conv_pre'), sd['conv_pre.bias'], padding=3)
    nk = len(c.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(c.upsample_rates, c.upsample_kernel_sizes)):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, _wn(sd, 'ups.%d' % i), sd['ups.%d.bias' % i], stride=u, padding=(k - u) // 2)
        xs = None
        for j, (kk, dils) in enumerate(zip(c.resblock_kernel_sizes, c.resblock_dilations)):
            p = 'resblocks.%d.' % (i * nk + j)
            r = x
            for d in range(3):
                xt = F.leaky_relu(r, 0.1)
                xt = F.conv1d(xt, _wn(sd, p + 'convs1.%d' % d), sd[p + 'convs1.%d.bias' % d], dilation=dils[d], padding=dils[d] * (kk - 1) // 2)
                xt = F.leaky_relu(xt, 0.1)
                xt = F.conv1d(xt, _wn(sd, p + 'convs2.%d' % d), sd[p + 'convs2.%d.bias' % d], padding=(kk - 1) // 2)
                r = xt + r
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, _wn(sd, 'conv_post'), sd['conv_post.bias'], padding=3)
    return torch.tanh(x)


def denoiser_bias(sd, c, mode='zeros', generator=None):
    """Denoiser.__init__ (denoiser.py:17-23, 49-55): |STFT| of the vocoder's response to the probe mel (88 frames: zeros, or N(0, 1) from the CPU generator
    in mode 'normal'), first frame"""
    mel_input = torch.zeros(1, c.mel, 88) if mode == 'zeros' else torch.randn((1, c.mel, 88), generator=generator)
    audio = generator_forward(sd, c, mel_input).float().squeeze(0)
    hop = c.n_fft // c.n_overlap
    spec = torch.stft(audio, c.n_fft, hop, c.n_fft, torch.hann_window(c.n_fft), return_complex=True)
    return spec.abs()[:, :, 0][:, :, None]                       # (1, bins, 1)


def denoise(audio, bias_spec, c, strength=0.0005):
    """(B, L) -> (B, hop * (frames - 1))"""
    hop = c.n_fft // c.n_overlap
    win = torch.hann_window(c.n_fft)
    spec = torch.view_as_real(torch.stft(audio, c.n_fft, hop, c.n_fft, win, return_complex=True))
    mag, ang = torch.sqrt(spec.pow(2).sum(-1)), torch.atan2(spec[..., -1], spec[..., 0])
    mag = torch.clamp(mag - bias_spec * strength, 0.0)
    return torch.istft(torch.complex(mag * torch.cos(ang), mag * torch.sin(ang)), c.n_fft, hop, c.n_fft, win)


def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, mel_basis):
    """matcha/utils/audio.py:45-82 with the librosa filterbank passed in (librosa is absent from this image; its table is restated in
    flowmirror_hydravox_amd.packing.mel_filterbank): (B, L) -> (B, num_mels, frames)"""
    p = int((n_fft - hop_size) / 2)
    y = F.pad(y.unsqueeze(1), (p, p), mode='reflect').squeeze(1)
    spec = torch.view_as_real(torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=torch.hann_window(win_size), center=False,
                                         pad_mode='reflect', normalized=False, onesided=True, return_complex=True))
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(mel_basis, spec), min=1e-5))
