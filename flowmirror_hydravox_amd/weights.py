"""Checkpoint layout of the hot path and seeded synthetic checkpoints.

The key names and shapes are exactly those of the reference's flat state-dict files
(`llm.pt`, `flow.pt`, `hift.pt`; server/model_utils/infer_speech_model.py:71-94, SURVEY.md
Appendix A.4): tests/golden/make_golden.py asserts the spec below against `state_dict()` of the
reference modules.  There is no network in the build environment, so benchmarks and tests use
seeded random weights of the right architecture (`make_*_state`).
"""
from typing import Dict, List, Tuple
import math
import torch

from .config import LLMConfig, FlowConfig, HiftConfig

Spec = List[Tuple[str, Tuple[int, ...], str]]   # (key, shape, kind)
DROP_KEYS = ('epoch', 'step', '_original_metadata', '_conversion_info')   # infer_speech_model.py:80-89


# --------------------------------------------------------------------------------------------------
# specs
# --------------------------------------------------------------------------------------------------
def _qwen2_layer_spec(pre: str, H: int, qd: int, kvd: int, inter: int) -> Spec:
    return [
        (pre + 'self_attn.q_proj.weight', (qd, H), 'w'), (pre + 'self_attn.q_proj.bias', (qd,), 'b'),
        (pre + 'self_attn.k_proj.weight', (kvd, H), 'w'), (pre + 'self_attn.k_proj.bias', (kvd,), 'b'),
        (pre + 'self_attn.v_proj.weight', (kvd, H), 'w'), (pre + 'self_attn.v_proj.bias', (kvd,), 'b'),
        (pre + 'self_attn.o_proj.weight', (H, qd), 'w'),
        (pre + 'mlp.gate_proj.weight', (inter, H), 'w'),
        (pre + 'mlp.up_proj.weight', (inter, H), 'w'),
        (pre + 'mlp.down_proj.weight', (H, inter), 'w'),
        (pre + 'input_layernorm.weight', (H,), 'g'),
        (pre + 'post_attention_layernorm.weight', (H,), 'g'),
    ]


def llm_spec(c: LLMConfig, with_lm_head: bool = True) -> Spec:
    H = c.hidden
    s: Spec = [('llm.model.model.embed_tokens.weight', (c.text_vocab, H), 'e')]
    for i in range(c.layers):
        s += _qwen2_layer_spec('llm.model.model.layers.%d.' % i, H, c.q_heads * c.head_dim,
                               c.kv_heads * c.head_dim, c.inter)
    s.append(('llm.model.model.norm.weight', (H,), 'g'))
    if with_lm_head:
        s.append(('llm.model.lm_head.weight', (c.text_vocab, H), 'e'))     # unused on the hot path
    s.append(('llm_decoder.weight', (c.vocab, H), 'w'))
    for j in range(c.head_num):
        a = c.mtp_attn_dim
        s += _qwen2_layer_spec('mtp_block.%d.' % j, H, a, a, c.mtp_inter)
    s.append(('speech_embedding.weight', (c.vocab, H), 'e'))
    return s


def flow_spec(c: FlowConfig) -> Spec:
    D, Cg = c.dim, c.dim // c.conv_groups
    e = 'decoder.estimator.'
    s: Spec = [
        ('input_embedding.weight', (c.vocab, c.mel), 'e'),
        ('spk_embed_affine_layer.weight', (c.mel, c.spk_embed_dim), 'w'),
        ('spk_embed_affine_layer.bias', (c.mel,), 'b'),
        ('pre_lookahead_layer.conv1.weight', (c.pre_lookahead_channels, c.mel, c.pre_lookahead_len + 1), 'w'),
        ('pre_lookahead_layer.conv1.bias', (c.pre_lookahead_channels,), 'b'),
        ('pre_lookahead_layer.conv2.weight', (c.mel, c.pre_lookahead_channels, 3), 'w'),
        ('pre_lookahead_layer.conv2.bias', (c.mel,), 'b'),
        (e + 'time_embed.time_mlp.0.weight', (D, c.time_freq_dim), 'w'), (e + 'time_embed.time_mlp.0.bias', (D,), 'b'),
        (e + 'time_embed.time_mlp.2.weight', (D, D), 'w'), (e + 'time_embed.time_mlp.2.bias', (D,), 'b'),
        (e + 'input_embed.proj.weight', (D, c.in_dim), 'w'), (e + 'input_embed.proj.bias', (D,), 'b'),
        (e + 'input_embed.conv_pos_embed.conv1.0.weight', (D, Cg, c.conv_kernel), 'w'),
        (e + 'input_embed.conv_pos_embed.conv1.0.bias', (D,), 'b'),
        (e + 'input_embed.conv_pos_embed.conv2.0.weight', (D, Cg, c.conv_kernel), 'w'),
        (e + 'input_embed.conv_pos_embed.conv2.0.bias', (D,), 'b'),
        (e + 'rotary_embed.inv_freq', (c.head_dim // 2,), 'inv_freq'),
    ]
    for i in range(c.depth):
        p = e + 'transformer_blocks.%d.' % i
        s += [
            (p + 'attn_norm.linear.weight', (6 * D, D), 'w'), (p + 'attn_norm.linear.bias', (6 * D,), 'b'),
            (p + 'attn.to_q.weight', (D, D), 'w'), (p + 'attn.to_q.bias', (D,), 'b'),
            (p + 'attn.to_k.weight', (D, D), 'w'), (p + 'attn.to_k.bias', (D,), 'b'),
            (p + 'attn.to_v.weight', (D, D), 'w'), (p + 'attn.to_v.bias', (D,), 'b'),
            (p + 'attn.to_out.0.weight', (D, D), 'w'), (p + 'attn.to_out.0.bias', (D,), 'b'),
            (p + 'ff.ff.0.0.weight', (c.ff, D), 'w'), (p + 'ff.ff.0.0.bias', (c.ff,), 'b'),
            (p + 'ff.ff.2.weight', (D, c.ff), 'w'), (p + 'ff.ff.2.bias', (D,), 'b'),
        ]
    s += [
        (e + 'norm_out.linear.weight', (2 * D, D), 'w'), (e + 'norm_out.linear.bias', (2 * D,), 'b'),
        (e + 'proj_out.weight', (c.mel, D), 'w'), (e + 'proj_out.bias', (c.mel,), 'b'),
    ]
    return s


def _wn_conv(name: str, cout: int, cin: int, k: int) -> Spec:
    return [(name + '.bias', (cout,), 'b'),
            (name + '.parametrizations.weight.original0', (cout, 1, 1), 'wn_g'),
            (name + '.parametrizations.weight.original1', (cout, cin, k), 'w')]


def _resblock_spec(pre: str, ch: int, k: int, n: int) -> Spec:
    s: Spec = []
    for i in range(n):
        s += _wn_conv('%sconvs1.%d' % (pre, i), ch, ch, k)
    for i in range(n):
        s += _wn_conv('%sconvs2.%d' % (pre, i), ch, ch, k)
    for i in range(n):
        s.append(('%sactivations1.%d.alpha' % (pre, i), (ch,), 'alpha'))
    for i in range(n):
        s.append(('%sactivations2.%d.alpha' % (pre, i), (ch,), 'alpha'))
    return s


def hift_source_down_rates(c: HiftConfig) -> List[int]:
    """downsample_cum_rates[::-1] of generator.py:637-640."""
    rates = [1] + c.upsample_rates[::-1][:-1]
    cum, acc = [], 1
    for r in rates:
        acc *= r
        cum.append(acc)
    return cum[::-1]


def hift_spec(c: HiftConfig) -> Spec:
    B = c.base_channels
    s: Spec = [('m_source.l_linear.weight', (1, c.nb_harmonics + 1), 'w'), ('m_source.l_linear.bias', (1,), 'b')]
    s += _wn_conv('conv_pre', B, c.mel, c.conv_pre_look_right + 1)
    for i, k in enumerate(c.upsample_kernel_sizes):
        s += _wn_conv('ups.%d' % i, B // (2 ** (i + 1)), B // (2 ** i), k)
    for i, u in enumerate(hift_source_down_rates(c)):
        ch = B // (2 ** (i + 1))
        kk = 1 if u == 1 else u * 2
        s += [('source_downs.%d.weight' % i, (ch, c.n_fft + 2, kk), 'w'), ('source_downs.%d.bias' % i, (ch,), 'b')]
    for i, k in enumerate(c.source_resblock_kernel_sizes):
        s += _resblock_spec('source_resblocks.%d.' % i, B // (2 ** (i + 1)), k, len(c.source_resblock_dilations[i]))
    n = 0
    for i in range(len(c.upsample_rates)):
        ch = B // (2 ** (i + 1))
        for j, k in enumerate(c.resblock_kernel_sizes):
            s += _resblock_spec('resblocks.%d.' % n, ch, k, len(c.resblock_dilations[j]))
            n += 1
    s += _wn_conv('conv_post', c.n_fft + 2, B // (2 ** len(c.upsample_rates)), 7)
    F0 = c.f0_channels
    s += _wn_conv('f0_predictor.condnet.0', F0, c.mel, 4)
    for i in (2, 4, 6, 8):
        s += _wn_conv('f0_predictor.condnet.%d' % i, F0, F0, 3)
    s += [('f0_predictor.classifier.weight', (1, F0), 'w'), ('f0_predictor.classifier.bias', (1,), 'b')]
    return s


# --------------------------------------------------------------------------------------------------
# Matcha-TTS family: state-dict keys of matcha.models.components.decoder.Decoder /
# cosyvoice.flow.decoder.ConditionalDecoder and matcha.hifigan.models.Generator
# --------------------------------------------------------------------------------------------------
def _matcha_resnet_spec(pre: str, cin: int, cout: int, te: int) -> Spec:
    return [(pre + 'mlp.1.weight', (cout, te), 'w'), (pre + 'mlp.1.bias', (cout,), 'b'),
            (pre + 'block1.block.0.weight', (cout, cin, 3), 'w'), (pre + 'block1.block.0.bias', (cout,), 'b'),
            (pre + 'block1.block.1.weight', (cout,), 'g'), (pre + 'block1.block.1.bias', (cout,), 'b'),
            (pre + 'block2.block.0.weight', (cout, cout, 3), 'w'), (pre + 'block2.block.0.bias', (cout,), 'b'),
            (pre + 'block2.block.1.weight', (cout,), 'g'), (pre + 'block2.block.1.bias', (cout,), 'b'),
            (pre + 'res_conv.weight', (cout, cin, 1), 'w'), (pre + 'res_conv.bias', (cout,), 'b')]


def _matcha_tblock_spec(pre: str, dim: int, inner: int, ff: int) -> Spec:
    return [(pre + 'norm1.weight', (dim,), 'g'), (pre + 'norm1.bias', (dim,), 'b'),
            (pre + 'attn1.to_q.weight', (inner, dim), 'w'), (pre + 'attn1.to_k.weight', (inner, dim), 'w'), (pre + 'attn1.to_v.weight', (inner, dim), 'w'),
            (pre + 'attn1.to_out.0.weight', (dim, inner), 'w'), (pre + 'attn1.to_out.0.bias', (dim,), 'b'),
            (pre + 'norm3.weight', (dim,), 'g'), (pre + 'norm3.bias', (dim,), 'b'),
            (pre + 'ff.net.0.proj.weight', (ff, dim), 'w'), (pre + 'ff.net.0.proj.bias', (ff,), 'b'),
            (pre + 'ff.net.0.alpha', (ff,), 'z'), (pre + 'ff.net.0.beta', (ff,), 'z'),
            (pre + 'ff.net.2.weight', (dim, ff), 'w'), (pre + 'ff.net.2.bias', (dim,), 'b')]


def matcha_spec(c) -> Spec:
    """Decoder.__init__ (decoder.py:201-300): module order down_blocks / mid_blocks / up_blocks / final_block / final_proj."""
    ch = tuple(c.channels)
    cin = c.in_channels
    te = ch[0] * 4
    inner = c.num_heads * c.head_dim
    s: Spec = [('time_mlp.linear_1.weight', (te, cin), 'w'), ('time_mlp.linear_1.bias', (te,), 'b'),
               ('time_mlp.linear_2.weight', (te, te), 'w'), ('time_mlp.linear_2.bias', (te,), 'b')]
    out = cin
    for i, co in enumerate(ch):
        inp, out = out, co
        p = 'down_blocks.%d.' % i
        s += _matcha_resnet_spec(p + '0.', inp, out, te)
        for j in range(c.n_blocks):
            s += _matcha_tblock_spec(p + '1.%d.' % j, out, inner, c.ff_mult * out)
        last = i == len(ch) - 1
        s += [(p + ('2.weight' if last else '2.conv.weight'), (out, out, 3), 'w'), (p + ('2.bias' if last else '2.conv.bias'), (out,), 'b')]
    for i in range(c.num_mid_blocks):
        p = 'mid_blocks.%d.' % i
        s += _matcha_resnet_spec(p + '0.', ch[-1], out, te)
        for j in range(c.n_blocks):
            s += _matcha_tblock_spec(p + '1.%d.' % j, out, inner, c.ff_mult * out)
    rev = ch[::-1] + (ch[0],)
    for i in range(len(rev) - 1):
        inp, out = rev[i], rev[i + 1]
        p = 'up_blocks.%d.' % i
        s += _matcha_resnet_spec(p + '0.', 2 * inp, out, te)
        for j in range(c.n_blocks):
            s += _matcha_tblock_spec(p + '1.%d.' % j, out, inner, c.ff_mult * out)
        last = i == len(rev) - 2
        if last:
            s += [(p + '2.weight', (out, out, 3), 'w'), (p + '2.bias', (out,), 'b')]
        else:
            s += [(p + '2.conv.weight', (out, out, 4), 'w'), (p + '2.conv.bias', (out,), 'b')]       # ConvTranspose1d: [Cin][Cout][k]
    s += [('final_block.block.0.weight', (rev[-1], rev[-1], 3), 'w'), ('final_block.block.0.bias', (rev[-1],), 'b'),
          ('final_block.block.1.weight', (rev[-1],), 'g'), ('final_block.block.1.bias', (rev[-1],), 'b'),
          ('final_proj.weight', (c.mel, rev[-1], 1), 'w'), ('final_proj.bias', (c.mel,), 'b')]
    return s


def hifigan_spec(c) -> Spec:
    s: Spec = []

    def wn(name, shape, nbias):
        s.extend([(name + '.weight_g', (shape[0], 1, 1), 'wn_g_old'), (name + '.weight_v', tuple(shape), 'w'), (name + '.bias', (nbias,), 'b')])

    C = c.initial_channel
    wn('conv_pre', (C, c.mel, 7), C)
    for i, (u, k) in enumerate(zip(c.upsample_rates, c.upsample_kernel_sizes)):
        wn('ups.%d' % i, (C // (2 ** i), C // (2 ** (i + 1)), k), C // (2 ** (i + 1)))               # ConvTranspose1d: [Cin][Cout][k]
    n = 0
    for i in range(len(c.upsample_rates)):
        chn = C // (2 ** (i + 1))
        for k in c.resblock_kernel_sizes:
            for d in range(3):
                wn('resblocks.%d.convs1.%d' % (n, d), (chn, chn, k), chn)
            for d in range(3):
                wn('resblocks.%d.convs2.%d' % (n, d), (chn, chn, k), chn)
            n += 1
    wn('conv_post', (1, C // (2 ** len(c.upsample_rates)), 7), 1)
    return s


# --------------------------------------------------------------------------------------------------
# seeded synthetic checkpoints
# --------------------------------------------------------------------------------------------------
def _fill(spec: Spec, seed: int, init: str, head_dim: int = 64) -> Dict[str, torch.Tensor]:
    """init='normal02': every Linear/Conv ~ N(0, 0.02), gains 1 (SURVEY.md §8(d) bench weights);
    init='fan_in'  : N(0, 1/fan_in) weights and perturbed gains/biases so that activations stay O(1)
                     through depth — used by parity tests so errors are not hidden by tiny signals."""
    g = torch.Generator()
    g.manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for key, shape, kind in spec:
        if kind in ('w', 'e'):
            if init == 'fan_in' and kind == 'w':
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
                std = 1.0 / math.sqrt(max(fan_in, 1))
            elif init == 'fan_in':
                std = 0.5
            else:
                std = 0.02
            t = torch.randn(shape, generator=g) * std
        elif kind == 'b':
            t = torch.randn(shape, generator=g) * (0.1 if init == 'fan_in' else 0.02)
        elif kind == 'g':
            t = torch.ones(shape)
            if init == 'fan_in':
                t = t + 0.1 * torch.randn(shape, generator=g)
        elif kind == 'alpha':
            t = torch.ones(shape)
            if init == 'fan_in':
                t = (t + 0.2 * torch.randn(shape, generator=g)).abs() + 0.05
        elif kind in ('wn_g', 'wn_g_old'):
            t = None          # filled after its v is drawn
        elif kind == 'z':     # log-scale parameter initialised at 0 (SnakeBeta alpha / beta)
            t = torch.zeros(shape)
            if init == 'fan_in':
                t = 0.2 * torch.randn(shape, generator=g)
        elif kind == 'inv_freq':
            t = 1.0 / (10000 ** (torch.arange(0, head_dim, 2).float() / head_dim))
        else:
            raise ValueError(kind)
        sd[key] = t
    for key, shape, kind in spec:
        if kind in ('wn_g', 'wn_g_old'):
            v = sd[key.replace('original0', 'original1') if kind == 'wn_g' else key.replace('weight_g', 'weight_v')]
            n = v.norm(2, dim=(1, 2), keepdim=True)
            sd[key] = n.clone() if init != 'fan_in' else n * (1.0 + 0.1 * torch.randn(shape, generator=g))
    return sd


def make_llm_state(c: LLMConfig, seed: int = 1986, init: str = 'normal02', with_lm_head: bool = False):
    sd = _fill(llm_spec(c, with_lm_head), seed, init)
    if init == 'fan_in':
        # damp the 200 stop-id rows so that random weights neither stop immediately nor never stop
        sd['llm_decoder.weight'][c.speech_tokens:] *= 0.3
    return sd


def make_flow_state(c: FlowConfig, seed: int = 1987, init: str = 'normal02'):
    return _fill(flow_spec(c), seed, init, head_dim=c.head_dim)


def make_hift_state(c: HiftConfig, seed: int = 1988, init: str = 'normal02'):
    sd = _fill(hift_spec(c), seed, init)
    if init == 'fan_in':
        # keep the synthetic vocoder in a well-conditioned regime: |mag| = exp(conv_post) ~ 0.2 so the
        # waveform is not clamped, and an F0 head that produces both voiced (>10 Hz) and unvoiced frames.
        sd['conv_post.parametrizations.weight.original0'] = sd['conv_post.parametrizations.weight.original0'] * 0.3
        sd['conv_post.bias'] = sd['conv_post.bias'] - 1.5
        sd['f0_predictor.classifier.weight'] = sd['f0_predictor.classifier.weight'] * 200.0
    return sd


def make_matcha_state(c, seed: int = 1989, init: str = 'normal02'):
    return _fill(matcha_spec(c), seed, init)


def make_hifigan_state(c, seed: int = 1990, init: str = 'normal02'):
    sd = _fill(hifigan_spec(c), seed, init)
    if init == 'fan_in':
        sd['conv_post.weight_g'] = sd['conv_post.weight_g'] * 0.5          # keep tanh out of saturation
    return sd


def check_state(sd: Dict[str, torch.Tensor], spec: Spec, what: str, optional=()):
    """Strict load_state_dict-style validation (infer_speech_model.py:92-94): raises on missing /
    unexpected keys or shape mismatch."""
    want = {k: s for k, s, _ in spec}
    have = {k for k in sd.keys() if k not in DROP_KEYS}
    missing = [k for k in want if k not in have and k not in optional]
    extra = [k for k in have if k not in want]
    if missing or extra:
        raise RuntimeError('Error(s) in loading state_dict for %s: missing keys %s, unexpected keys %s'
                           % (what, missing[:8], extra[:8]))
    for k, s in want.items():
        if k in sd and tuple(sd[k].shape) != tuple(s):
            raise RuntimeError('size mismatch for %s.%s: checkpoint %s vs model %s'
                               % (what, k, tuple(sd[k].shape), tuple(s)))


def accept_stress_llm_state(sd, decoder_scale=6.0, v_bias_scale=30.0):
    """SURVEY.md §8(d) "accept-stress" variant of a synthetic LM checkpoint: low-entropy, context-insensitive logits so that the
    repetition-aware sampler falls back to full-softmax resampling on roughly a third of its calls.  `llm_decoder` has no bias in
    CosyVoice3 (llm_multi_head_v3.py:652), so the bias is built from what the model has: the V-projection biases pass through every
    softmax-weighted average unchanged, i.e. they add a position-independent vector to the residual stream; scaled up they dominate
    the final hidden state, and a scaled decoder turns that into a sharp, nearly constant distribution.  Returns a new dict."""
    out = dict(sd)
    out['llm_decoder.weight'] = sd['llm_decoder.weight'] * decoder_scale
    for k in sd:
        if k.startswith('llm.model.model.layers.') and k.endswith('self_attn.v_proj.bias'):
            out[k] = sd[k] * v_bias_scale
    return out
