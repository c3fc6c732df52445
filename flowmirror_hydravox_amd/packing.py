"""Load-time weight layouts (torch ops used as plumbing: permute / pad / cast, no arithmetic on the hot path).

* `pack_frag`      — [N][K] row-major -> MFMA fragment order [N/16][K/32][64 lanes][8] consumed by the decode GEMMs
                      (csrc/gemm_skinny.hip): lane l = g*16 + r holds row r of the 16-row tile, k = 8g..8g+7.
* `conv_weight`    — torch Conv1d weight [Cout][Cin][k] -> [Cout][k][Cin_pad32] (k-major so that a tap is a row shift
                      of the time-major activation; csrc/gemm_tiled.hip).
* `fold_weight_norm` — w = v * (g / ||v||), the new-style parametrisation of hifigan/generator.py:26-29.
"""
import torch


def pad_to(x, dim, mult):
    n = x.shape[dim]
    tgt = (n + mult - 1) // mult * mult
    if tgt == n:
        return x
    shape = list(x.shape)
    shape[dim] = tgt - n
    return torch.cat([x, x.new_zeros(shape)], dim=dim)


def pack_frag(w):
    """[N][K] -> flat tensor in [N/16][K/32][g=4][r=16][8] order (N padded to 16, K must be a multiple of 32)."""
    assert w.dim() == 2 and w.shape[1] % 32 == 0, w.shape
    w = pad_to(w, 0, 16)
    N, K = w.shape
    return w.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


def pack_narrow4(w):
    """[N][K] -> flat tensor in [N/4][K/128][g=4][ksub=4][c=4][8] order (N % 4 == 0, K % 128 == 0): lane (c + 4*ksub, g) of the
    4-column workgroup form (csrc/gemm_skinny.hip: gemm_narrow_resid_kernel) holds W[4*nt + c][128*kb + 32*ksub + 8*g ..+8]."""
    assert w.dim() == 2 and w.shape[0] % 4 == 0 and w.shape[1] % 128 == 0, w.shape
    N, K = w.shape
    return w.view(N // 4, 4, K // 128, 4, 4, 8).permute(0, 2, 4, 3, 1, 5).contiguous().view(N, K)


def qkv_row_perm(n_heads, head_dim=64):
    """Row order of the fused [q | k | v] projection consumed by the SK_QKV_ROPE epilogue: inside every 64-row head, 16-row tile t
    holds d = 8t..8t+7 followed by their rotate-half partners d + 32, so a RoPE pair sits in one MFMA tile (lanes fr and fr ^ 8)."""
    assert head_dim == 64
    idx = []
    for h in range(n_heads):
        for t in range(4):
            idx += [h * 64 + 8 * t + i for i in range(8)] + [h * 64 + 32 + 8 * t + i for i in range(8)]
    return torch.tensor(idx, dtype=torch.long)


def interleave_gate_up(wg, wu):
    """[I][H] gate and up projections -> [2 I][H] with alternating 16-row tiles (the SwiGLU epilogue pairs tile 2p with 2p + 1)"""
    I, H = wg.shape
    assert I % 16 == 0 and wu.shape == wg.shape
    g = wg.reshape(I // 16, 1, 16, H)
    u = wu.reshape(I // 16, 1, 16, H)
    return torch.cat([g, u], dim=1).reshape(2 * I, H)


def pack_gate_up(wg, wu):
    """Interleave gate / up projections as alternating 16-row tiles, then pack (SwiGLU epilogue pairs tile 2p with 2p+1)."""
    return pack_frag(interleave_gate_up(wg, wu))


def frag_fp8_to_frag(codes_packed, scale, dtype):
    """pack_frag_fp8 codes [N][K] + row scales [N] -> the pack_frag tensor of the dequantised matrix (what a bf16 kernel streams): the two
    fragment orders are permutations of each other, so a packed cache can hold the codes alone (checkpoint.save_packed)."""
    N, K = codes_packed.shape
    v = codes_packed.view(torch.float8_e4m3fn).float().view(N // 16, K // 64, 4, 16, 2, 8)        # [t][ds][g][r][half][8]
    v = v * scale.view(N // 16, 1, 1, 16, 1, 1)
    return v.permute(0, 1, 4, 2, 3, 5).contiguous().view(N, K).to(dtype)


E4M3_MAX = 448.0


def quantize_e4m3_pow2(w):
    """[N][K] float -> (codes uint8 [N][K], scale fp32 [N], dequantised fp32 [N][K]): every row of W (one output column of the GEMM) as OCP e4m3
    codes times ONE power-of-two scale, the smallest that brings the row's largest magnitude inside +-448.  A power of two makes code * scale exact
    in bf16 (3 mantissa bits scaled by an exponent shift) and lets the scale commute with fp32 accumulation: the bf16 tensor `dequantised` and the
    (codes, scale) pair describe the same GEMM bit for bit (csrc/gemm_dec.hip, W8).  Rounding is torch's float8_e4m3fn cast (nearest even)."""
    assert w.dim() == 2
    w = w.float()
    amax = w.abs().amax(dim=1).clamp_min(2.0 ** -100)
    scale = torch.exp2(torch.ceil(torch.log2(amax / E4M3_MAX)))
    scale = torch.where(amax / scale > E4M3_MAX, scale * 2, scale)          # (log2 rounding at exact powers of two)
    q8 = (w / scale[:, None]).to(torch.float8_e4m3fn)
    deq = q8.float() * scale[:, None]
    return q8.view(torch.uint8), scale.contiguous(), deq


def pack_frag_fp8(codes):
    """uint8 codes [N][K] (N % 16 == 0, K % 64 == 0) -> flat [N/16][K/64][64 lanes][16]: lane (r, g) of a fragment holds the 8 codes
    k = 64 j + 8 g .. + 8 of row 16 t + r followed by those of k + 32 — the two k-steps one 1 KiB ring fragment of the fp8 stream carries
    (csrc/gemm_dec.hip, W8), in the lane order of pack_frag."""
    assert codes.dim() == 2 and codes.dtype == torch.uint8 and codes.shape[0] % 16 == 0 and codes.shape[1] % 64 == 0, (codes.shape, codes.dtype)
    N, K = codes.shape
    return codes.view(N // 16, 16, K // 64, 2, 4, 8).permute(0, 2, 4, 1, 3, 5).contiguous().view(N, K)


def conv_weight(w, cin_pad=None):
    """[Cout][Cin][k] -> [Cout][k][Cin_pad] (flattened to [Cout][k*Cin_pad])."""
    cout, cin, k = w.shape
    cp = cin_pad or (cin + 31) // 32 * 32
    o = w.new_zeros(cout, k, cp)
    o[:, :, :cin] = w.permute(0, 2, 1)
    return o.reshape(cout, k * cp).contiguous()


def grouped_conv_weight(w, groups):
    """[D][Cg][k] (groups blocks of Cg output channels) -> [groups][Cg][k*Cg]."""
    D, cg, k = w.shape
    assert D % groups == 0 and D // groups == cg and cg % 32 == 0
    return w.view(groups, cg, cg, k).permute(0, 1, 3, 2).reshape(groups, cg, k * cg).contiguous()


def linear_weight(w, k_pad=None):
    """[N][K] with K zero-padded to a multiple of 32."""
    return pad_to(w, 1, 32).contiguous()


def fold_weight_norm(sd, name):
    k0 = name + '.parametrizations.weight.original0'
    if k0 in sd:
        g = sd[k0].float()
        v = sd[name + '.parametrizations.weight.original1'].float()
        return v * (g / v.norm(2, dim=(1, 2), keepdim=True))
    return sd[name + '.weight'].float()


def convtranspose_phases(w, stride, padding):
    """ConvTranspose1d weight [Cin][Cout][k] -> [stride][Cout][k/stride taps][Cin_pad32]: output phase p (rows t*stride + p) is a
    convolution over the input rows t + c_p - (taps-1) .. t + c_p with c_p = (p + padding) // stride; tap tau multiplies
    w[:, :, (p + padding) % stride + stride * (taps - 1 - tau)]   (y[n] = sum_i x[i] w[n - i*stride + padding])."""
    cin, cout, k = w.shape
    assert k % stride == 0
    taps = k // stride
    cp = (cin + 31) // 32 * 32
    out = w.new_zeros(stride, cout, taps, cp)
    for p in range(stride):
        j0 = (p + padding) % stride
        for tau in range(taps):
            out[p, :, tau, :cin] = w[:, :, j0 + stride * (taps - 1 - tau)].t()
    return out.reshape(stride, cout, taps * cp).contiguous()


def stft_bases(n_fft, win_length=None):
    """Windowed DFT bases of torch.stft / torch.istft (hann, periodic) as GEMM operands:
    analysis [2*bins][n_fft] (rows re_0..re_{N/2}, im_0..im_{N/2}), synthesis [n_fft][pad32(2*bins)] (irfft, window and 1/N folded in),
    wsq [n_fft] = window^2."""
    import math
    N = n_fft
    bins = N // 2 + 1
    win = torch.hann_window(win_length or N, dtype=torch.float64)
    n = torch.arange(N, dtype=torch.float64)
    k = torch.arange(bins, dtype=torch.float64)
    ang = 2.0 * math.pi * k[:, None] * n[None, :] / N
    ana = torch.cat([torch.cos(ang) * win[None, :], -torch.sin(ang) * win[None, :]], 0)           # X_k = sum_n w x e^{-i ang}
    wk = torch.full((bins,), 2.0, dtype=torch.float64)
    wk[0] = 1.0
    wk[-1] = 1.0
    syn_re = (torch.cos(ang) * wk[:, None]).t() / N                                                   # [N][bins]
    syn_im = (-torch.sin(ang) * wk[:, None]).t() / N
    syn_im[:, 0] = 0.0                                                                                # irfft ignores Im of DC and Nyquist
    syn_im[:, -1] = 0.0
    syn = torch.cat([syn_re, syn_im], 1) * win[:, None]
    ld = (2 * bins + 31) // 32 * 32
    syn_p = torch.zeros(N, ld, dtype=torch.float64)
    syn_p[:, :2 * bins] = syn
    return ana.float().contiguous(), syn_p.float().contiguous(), (win * win).float().contiguous()


def whisper_bases(n_mels=128):
    """GEMM operands of whisper.log_mel_spectrogram (whisper/audio.py:110-157: N_FFT 400, HOP 160, periodic Hann, 201 bins, librosa Slaney
    mel filters for sr 16000) -> (basis f32 [2*201][416], mel f32 [n_mels][pad32(201)]); the 400-sample frame is given as 416 samples
    (13 rows of 32) whose last 16 basis columns are zero."""
    import math
    N, bins, FL = 400, 201, 416
    win = torch.hann_window(N, dtype=torch.float64)
    n = torch.arange(N, dtype=torch.float64)
    k = torch.arange(bins, dtype=torch.float64)
    ang = 2.0 * math.pi * k[:, None] * n[None, :] / N
    ana = torch.zeros(2 * bins, FL, dtype=torch.float64)
    ana[:bins, :N] = torch.cos(ang) * win[None, :]
    ana[bins:, :N] = -torch.sin(ang) * win[None, :]
    mel = torch.zeros(n_mels, (bins + 31) // 32 * 32)
    mel[:, :bins] = mel_filterbank(16000, N, n_mels, 0.0, 8000.0)
    return ana.float().contiguous(), mel.contiguous()


def kaldi_fbank_bases(num_mel_bins=80, sample_frequency=16000.0, low_freq=20.0, preemphasis=0.97):
    """GEMM operands of torchaudio.compliance.kaldi.fbank at its defaults (25 ms / 10 ms frames, DC removal, pre-emphasis, Povey window,
    FFT zero-padded to 512, power spectrum, mel(f) = 1127 ln(1 + f / 700) triangles from `low_freq` to Nyquist).  Everything up to the
    FFT is linear in the frame x:  X = F_512 . pad . diag(povey) . P . (I - 1 1^T / N) x,  so it is folded into ONE basis matrix in
    float64 -> (basis f32 [2*257][416], mel f32 [num_mel_bins][pad32(257)])."""
    import math
    N, padded, FL = int(sample_frequency * 0.025), 512, 416
    bins = padded // 2 + 1
    assert N == 400
    eye = torch.eye(N, dtype=torch.float64)
    dc = eye - torch.full((N, N), 1.0 / N, dtype=torch.float64)
    pre = eye.clone()
    pre[0, 0] -= preemphasis                                  # the first sample is its own predecessor (replicate padding)
    pre[torch.arange(1, N), torch.arange(0, N - 1)] -= preemphasis
    win = torch.hann_window(N, periodic=False, dtype=torch.float64).pow(0.85)
    lin = win[:, None] * (pre @ dc)                           # frame -> windowed frame
    n = torch.arange(N, dtype=torch.float64)
    k = torch.arange(bins, dtype=torch.float64)
    ang = 2.0 * math.pi * k[:, None] * n[None, :] / padded
    ana = torch.zeros(2 * bins, FL, dtype=torch.float64)
    ana[:bins, :N] = torch.cos(ang) @ lin
    ana[bins:, :N] = -torch.sin(ang) @ lin
    nyq = 0.5 * sample_frequency
    ml, mh = 1127.0 * math.log(1.0 + low_freq / 700.0), 1127.0 * math.log(1.0 + nyq / 700.0)
    delta = (mh - ml) / (num_mel_bins + 1)
    b = torch.arange(num_mel_bins, dtype=torch.float64)[:, None]
    mk = (1127.0 * torch.log(1.0 + (sample_frequency / padded) * torch.arange(padded // 2, dtype=torch.float64) / 700.0))[None, :]
    up, down = (mk - (ml + b * delta)) / delta, ((ml + (b + 2.0) * delta) - mk) / delta
    mel = torch.zeros(num_mel_bins, (bins + 31) // 32 * 32)
    mel[:, :padded // 2] = torch.clamp(torch.minimum(up, down), min=0.0).float()
    return ana.float().contiguous(), mel.contiguous()


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults htk=False, norm='slaney' (librosa is what
    matcha/utils/audio.py:53 calls; it is not installed here, this is its published algorithm): triangular filters on the Slaney mel
    scale (linear below 1 kHz at 200/3 Hz per mel, logarithmic above with step ln(6.4)/27), each scaled by 2 / (its band width in Hz).
    -> float32 [n_mels][n_fft // 2 + 1]"""
    import numpy as np
    fmax = sr / 2.0 if fmax is None else fmax
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return torch.from_numpy(w.astype(np.float32))
