"""CPU oracle for the flow-matching mel decoder (TEST INFRASTRUCTURE — see oracle/__init__.py).

Follows
  * CausalMaskedDiffWithDiT.inference   server/model_utils/cosyvoice/flow/flow.py:367-430
  * PreLookaheadLayer.forward           cosyvoice/transformer/upsample_encoder.py:82-103
  * CausalConditionalCFM.forward        cosyvoice/flow/flow_matching.py:204-228
  * ConditionalCFM.solve_euler          cosyvoice/flow/flow_matching.py:71-124
  * DiT.forward                         cosyvoice/flow/DiT/dit.py:145-176
  * TimestepEmbedding / SinusPositionEmbedding   cosyvoice/flow/DiT/modules.py:606-616, 71-83
  * InputEmbedding / CausalConvPositionEmbedding cosyvoice/flow/DiT/dit.py:76-98, modules.py:115-144
  * DiTBlock / AdaLayerNormZero / AttnProcessor / FeedForward  modules.py:516-530, 230-244, 349-407, 271-282
  * AdaLayerNormZero_Final              modules.py:251-265
Rotary embedding arithmetic is third-party x_transformers==2.12.2 (requirements.txt:51, absent from
/root/reference): interleaved-pair rotation of the FIRST head_dim channels of the full q/k rows
(applied before the head split, so only head 0 rotates), fp32 -> PARITY UNPINNED for this sub-step.
Weights: flat dict with the reference's flow.pt keys.
"""
import math
import torch
import torch.nn.functional as F


# bf16-faithful mode (`emu=True`, see oracle/llm_ref.py): every GEMM / conv / attention operand rounded to bf16 where the product's bf16
# path holds it in bf16 (weights, adaLN-modulated rows, q / k / v, probabilities, GELU / Mish / LeakyReLU outputs that feed the next
# GEMM), fp32 accumulation, fp32 residual stream, fp32 Euler state (csrc/hvx_flow.hip).  Three switches follow the product's bf16-mode options:
#   resid16  the DiT's residual stream rounded to IEEE fp16 after every residual add          (hvx_flow_set_half_stream; default on)
#   lin16    QKV / FF1 / FF2 of every block on fp16 operands instead of bf16                   (hvx_flow_set_f16_linears; default on)
#   small32  time MLP, adaLN modulation Linears, input and output projection left in fp32      (hvx_flow_set_f32_small;   default on)
def bf16r(t):
    return t.to(torch.bfloat16).float()


def _r(t, emu):
    return bf16r(t) if emu else t


def _h(t, on):
    """resid16: the residual stream of the DiT blocks stored in IEEE fp16 (the product's bf16 mode, csrc/hvx_flow.hip half_stream; the
    reference holds it in fp16 too when it runs `.half()`, infer_speech_model.py:103) — one rounding per residual add"""
    return t.to(torch.float16).float() if on else t


def _lin(x, w, b=None, emu=False, f16=False):
    """f16: the product runs this Linear on IEEE fp16 operands (the four Linears of a DiT block with hvx_flow_set_f16_linears)"""
    if emu and f16:
        return F.linear(x.to(torch.float16).float(), w.to(torch.float16).float(), b)
    return F.linear(bf16r(x), bf16r(w), b) if emu else F.linear(x, w, b)


def pre_lookahead(x, sd, cfg, pre='pre_lookahead_layer.', context=None, emu=False):
    """x: (1, N, 80) -> (1, N, 80); zero right pad, or `context` (1, pre_lookahead_len, 80) as the look-ahead (finalize=False)."""
    o = x.transpose(1, 2).contiguous()
    if context is None:
        o = F.pad(o, (0, cfg.pre_lookahead_len), value=0.0)
    else:
        assert context.shape[1] == cfg.pre_lookahead_len
        o = torch.cat([o, context.transpose(1, 2)], dim=2)
    o = F.leaky_relu(F.conv1d(_r(o, emu), _r(sd[pre + 'conv1.weight'], emu), sd[pre + 'conv1.bias']))
    k2 = sd[pre + 'conv2.weight'].shape[-1]
    o = F.pad(o, (k2 - 1, 0), value=0.0)
    o = F.conv1d(_r(o, emu), _r(sd[pre + 'conv2.weight'], emu), sd[pre + 'conv2.bias'])
    return o.transpose(1, 2).contiguous() + x


def sinus_pos_emb(t, dim, scale=1000.0):
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half).float() * -e)
    e = scale * t.unsqueeze(1) * e.unsqueeze(0)
    return torch.cat((e.sin(), e.cos()), dim=-1)


def time_embed(t, sd, cfg, pre, emu=False):   # (callers pass emu=False for the product's f32_small mode)
    h = sinus_pos_emb(t, cfg.time_freq_dim).to(t.dtype)
    h = _lin(h, sd[pre + 'time_embed.time_mlp.0.weight'], sd[pre + 'time_embed.time_mlp.0.bias'], emu)
    return _lin(F.silu(h), sd[pre + 'time_embed.time_mlp.2.weight'], sd[pre + 'time_embed.time_mlp.2.bias'], emu)


def rope_freqs(T, head_dim):
    inv = 1.0 / (10000 ** (torch.arange(0, head_dim, 2).float() / head_dim))
    fr = torch.einsum('i,j->ij', torch.arange(T).float(), inv)
    return torch.stack((fr, fr), dim=-1).flatten(-2)          # (T, head_dim), interleaved duplicate


def apply_rope_first(x, freqs):
    """Rotate the first freqs.shape[-1] channels of x (B,T,D) in interleaved pairs, pass the rest."""
    r = freqs.shape[-1]
    xr, xp = x[..., :r].float(), x[..., r:]
    x2 = xr.reshape(*xr.shape[:-1], -1, 2)
    a, b = x2.unbind(-1)
    rot = torch.stack((-b, a), dim=-1).flatten(-2)
    xr = xr * freqs.cos() + rot * freqs.sin()
    return torch.cat((xr.to(x.dtype), xp), dim=-1)


def causal_conv_pos_embed(x, sd, cfg, pre, emu=False):
    k = cfg.conv_kernel
    h = x.permute(0, 2, 1)
    for name in ('conv1.0.', 'conv2.0.'):
        h = F.pad(h, (k - 1, 0))
        h = F.mish(F.conv1d(_r(h, emu), _r(sd[pre + name + 'weight'], emu), sd[pre + name + 'bias'], groups=cfg.conv_groups))
    return h.permute(0, 2, 1)


def _sdpa_emu(q, k, v, am):
    """softmax(q k^T / sqrt(d)) v with bf16 operands: scores fp32, probabilities rounded to bf16 for both the value product and the row
    sum (csrc/attention.hip: attn_dit_kernel sums the bf16 probabilities on the matrix cores), output rounded to bf16."""
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    s = s.masked_fill(~am, float('-inf'))
    e = bf16r(torch.exp(s - s.amax(dim=-1, keepdim=True)))
    return bf16r(torch.matmul(e, v) / e.sum(dim=-1, keepdim=True))


def dit_block(x, t_emb, sd, cfg, pre, freqs, key_mask, attn_mask=None, emu=False, resid16=False, lin16=False, small32=False):
    B, T, D = x.shape
    H, dh = cfg.heads, cfg.head_dim
    emb = _lin(F.silu(t_emb), sd[pre + 'attn_norm.linear.weight'], sd[pre + 'attn_norm.linear.bias'], emu and not small32)
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = torch.chunk(emb, 6, dim=1)
    n = F.layer_norm(x, (D,), eps=1e-6) * (1 + sc_a[:, None]) + sh_a[:, None]
    q = _lin(n, sd[pre + 'attn.to_q.weight'], sd[pre + 'attn.to_q.bias'], emu, lin16)
    k = _lin(n, sd[pre + 'attn.to_k.weight'], sd[pre + 'attn.to_k.bias'], emu, lin16)
    v = _lin(n, sd[pre + 'attn.to_v.weight'], sd[pre + 'attn.to_v.bias'], emu, lin16)
    q = apply_rope_first(q, freqs)
    k = apply_rope_first(k, freqs)
    q = _r(q, emu).view(B, T, H, dh).transpose(1, 2)
    k = _r(k, emu).view(B, T, H, dh).transpose(1, 2)
    v = _r(v, emu).view(B, T, H, dh).transpose(1, 2)
    am = key_mask[:, None, None, :].expand(B, H, T, T)           # (B,1,T,T) repeated pad mask (dit.py:166)
    if attn_mask is not None:
        am = attn_mask[:, None].expand(B, H, T, T)                # streaming: pad mask & static chunk mask (dit.py:163-164)
    if emu:
        a = _sdpa_emu(q, k, v, am)
    else:
        a = F.scaled_dot_product_attention(q, k, v, attn_mask=am, dropout_p=0.0, is_causal=False)
    a = a.transpose(1, 2).reshape(B, T, H * dh)
    a = _lin(a, sd[pre + 'attn.to_out.0.weight'], sd[pre + 'attn.to_out.0.bias'], emu)          # (stays bf16 in the product: its input is the attention's bf16 output)
    a = a.masked_fill(~key_mask[:, :, None], 0.0)                 # mask[:, 0, -1] row == pad mask (modules.py:400-405)
    x = _h(x + g_a.unsqueeze(1) * a, resid16)
    f = F.layer_norm(x, (D,), eps=1e-6) * (1 + sc_m[:, None]) + sh_m[:, None]
    f = _lin(f, sd[pre + 'ff.ff.0.0.weight'], sd[pre + 'ff.ff.0.0.bias'], emu, lin16)
    f = _lin(F.gelu(f, approximate='tanh'), sd[pre + 'ff.ff.2.weight'], sd[pre + 'ff.ff.2.bias'], emu, lin16)
    return _h(x + g_m.unsqueeze(1) * f, resid16)


def chunk_attn_mask(key_mask, chunk):
    """add_optional_chunk_mask(xs, masks, False, False, 0, static_chunk_size, -1) (cosyvoice/utils/mask.py:161-236 with
    subsequent_chunk_mask :128-158): key j visible to row i iff mask[j] and j < (i // chunk + 1) * chunk; all-false rows -> all true."""
    T = key_mask.shape[1]
    pos = torch.arange(T)
    cm = pos[None, :] < ((pos // chunk + 1) * chunk)[:, None]
    am = key_mask[:, None, :] & cm[None]
    am[am.sum(dim=-1) == 0] = True
    return am


def dit_forward(x, mask, mu, t, spks, cond, sd, cfg, pre='decoder.estimator.', n_blocks=None, taps=None, streaming=False, emu=False, resid16=False, lin16=False, small32=False):
    # small32: the product's f32_small mode — time MLP, adaLN modulation, input and output projection in fp32 (hvx_flow_set_f32_small)
    """Estimator call, TRT argument order (flow_matching.py:126-153): x,mu,cond (B,80,T); mask (B,1,T);
    t (B,); spks (B,80) -> (B,80,T).  streaming=True: static chunk mask of cfg.static_chunk_size frames."""
    x = x.transpose(1, 2)
    mu = mu.transpose(1, 2)
    cond = cond.transpose(1, 2)
    B, T, _ = x.shape
    if t.ndim == 0:
        t = t.repeat(B)
    t_emb = time_embed(t, sd, cfg, pre, emu and not small32)
    h = torch.cat([x, cond, mu, spks[:, None, :].expand(B, T, spks.shape[-1])], dim=-1)
    h = _lin(h, sd[pre + 'input_embed.proj.weight'], sd[pre + 'input_embed.proj.bias'], emu and not small32)
    h = _h(causal_conv_pos_embed(h, sd, cfg, pre + 'input_embed.conv_pos_embed.', emu) + h, resid16)
    if taps is not None:
        taps['input_embed'] = h.clone()
    freqs = rope_freqs(T, cfg.head_dim)
    key_mask = mask.bool()[:, 0, :]
    nb = cfg.depth if n_blocks is None else n_blocks
    attn_mask = chunk_attn_mask(key_mask, cfg.static_chunk_size) if streaming else None
    for i in range(nb):
        h = dit_block(h, t_emb, sd, cfg, pre + 'transformer_blocks.%d.' % i, freqs, key_mask, attn_mask, emu, resid16, lin16, small32)
        if taps is not None:
            taps['block%d' % i] = h.clone()
    emb = _lin(F.silu(t_emb), sd[pre + 'norm_out.linear.weight'], sd[pre + 'norm_out.linear.bias'], emu and not small32)
    scale, shift = torch.chunk(emb, 2, dim=1)
    h = F.layer_norm(h, (h.shape[-1],), eps=1e-6) * (1 + scale)[:, None, :] + shift[:, None, :]
    return _lin(h, sd[pre + 'proj_out.weight'], sd[pre + 'proj_out.bias'], emu and not small32).transpose(1, 2)


def cosine_t_span(n_timesteps, dtype=torch.float32):
    t = torch.linspace(0, 1, n_timesteps + 1, dtype=dtype)
    return 1 - torch.cos(t * 0.5 * torch.pi)


def cfm_noise(cfg):
    """CausalConditionalCFM.__init__: set_all_random_seed(0); randn(1,80,50*300) (flow_matching.py:200-201)."""
    g = torch.Generator()
    g.manual_seed(0)
    return torch.randn([1, cfg.mel, cfg.noise_frames], generator=g)


def solve_euler(x, t_span, mu, mask, spks, cond, estimator, cfg_rate):
    """flow_matching.py:71-124 with the batch-2 CFG trick (row 1 = unconditional zeros)."""
    t, dt = t_span[0], t_span[1] - t_span[0]
    t = t.unsqueeze(0)
    T = x.size(2)
    x_in = torch.zeros([2, x.size(1), T], dtype=spks.dtype)
    mask_in = torch.zeros([2, 1, T], dtype=spks.dtype)
    mu_in = torch.zeros([2, x.size(1), T], dtype=spks.dtype)
    t_in = torch.zeros([2], dtype=spks.dtype)
    spks_in = torch.zeros([2, spks.size(1)], dtype=spks.dtype)
    cond_in = torch.zeros([2, x.size(1), T], dtype=spks.dtype)
    traj = []
    for step in range(1, len(t_span)):
        x_in[:] = x
        mask_in[:] = mask
        mu_in[0] = mu
        t_in[:] = t.unsqueeze(0)
        spks_in[0] = spks
        cond_in[0] = cond
        d = estimator(x_in, mask_in, mu_in, t_in, spks_in, cond_in)
        d, d_cfg = torch.split(d, [x.size(0), x.size(0)], dim=0)
        d = (1.0 + cfg_rate) * d - cfg_rate * d_cfg
        x = x + dt * d
        t = t + dt
        traj.append(x)
        if step < len(t_span) - 1:
            dt = t_span[step + 1] - t
    return traj[-1].float(), traj


def cfm_forward(mu, mask, spks, cond, sd, cfg, noise=None, estimator=None, n_timesteps=None, streaming=False, emu=False, resid16=False, lin16=False, small32=False):
    """CausalConditionalCFM.forward (flow_matching.py:204-228)."""
    noise = cfm_noise(cfg) if noise is None else noise
    z = noise[:, :, :mu.size(2)].to(mu.dtype)
    n = cfg.n_timesteps if n_timesteps is None else n_timesteps
    t_span = cosine_t_span(n, mu.dtype)
    if estimator is None:
        def estimator(x, m, mu_, t, s, c):
            return dit_forward(x, m, mu_, t, s, c, sd, cfg, streaming=streaming, emu=emu, resid16=resid16, lin16=lin16, small32=small32)
    out, _ = solve_euler(z, t_span, mu, mask, spks, cond, estimator, cfg.cfg_rate)
    return out


def flow_inference(token, embedding, sd, cfg, prompt_token=None, prompt_feat=None, noise=None, finalize=True, streaming=False, emu=False, resid16=False, lin16=False, small32=False):
    """flow.py:367-430, fp32; finalize=False: the last pre_lookahead_len tokens are look-ahead context only.
    token (1,N) int, embedding (1,192), prompt_token (1,Np) int, prompt_feat (1,Tp,80) -> mel (1,80,2N)."""
    emb = F.normalize(embedding.float(), dim=1)
    emb = F.linear(emb, sd['spk_embed_affine_layer.weight'], sd['spk_embed_affine_layer.bias'])
    if prompt_token is not None:
        token = torch.cat([prompt_token, token], dim=1)
    h = sd['input_embedding.weight'][torch.clamp(token.long(), min=0)]          # mask is all ones for B=1
    L = cfg.pre_lookahead_len
    h = pre_lookahead(h, sd, cfg, emu=emu) if finalize else pre_lookahead(h[:, :-L], sd, cfg, context=h[:, -L:], emu=emu)
    h = h.repeat_interleave(cfg.token_mel_ratio, dim=1)
    T = h.shape[1]
    mel_len1 = prompt_feat.shape[1] if prompt_feat is not None else 0
    cond = torch.zeros(1, T, cfg.mel)
    if prompt_feat is not None:
        cond[:, :mel_len1] = prompt_feat
    mask = torch.ones(1, 1, T)
    feat = cfm_forward(h.transpose(1, 2).contiguous(), mask, emb, cond.transpose(1, 2), sd, cfg, noise=noise, streaming=streaming, emu=emu, resid16=resid16, lin16=lin16, small32=small32)
    return feat[:, :, mel_len1:].float()
