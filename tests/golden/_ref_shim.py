"""Import shim for the upstream reference (THIS CONTAINER ONLY, golden-vector generation).

The reference lives read-only at /root/reference and is pure Python.  Several of its imports are
absent from this image (torchaudio, x_transformers, diffusers, omegaconf, conformer, hydra...).
This shim installs the minimal stand-in *modules* needed so the reference's own hot-path classes
import unmodified.  Nothing here is shipped, nothing here is imported by the product or by the
`-m gpu` tests; it only serves tests/golden/make_golden.py.

Arithmetic restated here (third-party packages the reference imports but this image lacks):
  * diffusers.models.attention_processor.Attention (diffusers==0.25.0 in the reference's requirements; self-attention use only):
    to_q / to_k / to_v Linear without bias, to_out = [Linear with bias, Dropout], heads of dim_head, scale dim_head**-0.5, the
    `attention_mask` (2-D (B, T) or 3-D (B, T, T) float) is ADDED to the scores, broadcast over heads (prepare_attention_mask +
    AttnProcessor2_0).  Used by matcha's BasicTransformerBlock; parity of that sub-step is therefore pinned on the reference's call
    sites, not on diffusers itself.
  * x_transformers' RotaryEmbedding / apply_rotary_pos_emb
(third-party, pinned x_transformers==2.12.2 in the reference's requirements.txt:51, not installed):
    inv_freq = 1 / 10000^(2i/dim); freqs = pos (x) inv_freq, each value duplicated *interleaved*
    (f0,f0,f1,f1,...); rotate_half on interleaved pairs (-x2, x1); only the first rot_dim channels
    of the tensor are rotated, the rest pass through; computed in fp32, cast back; scale == 1.
"""
import sys
import types
import logging

REF_ROOT = '/root/reference'


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def stand_ins():
    """The restated third-party classes / functions (module docstring), importable WITHOUT the reference: tests/test_oracle_golden.py holds them to independent
    implementations that this image does have (torch's own scaled_dot_product_attention = what diffusers' AttnProcessor2_0 calls; transformers' GPT-J rotary
    functions = the same interleaved-pair rotation as x_transformers')."""
    import torch
    from torch import nn

    class Attention(nn.Module):
        def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False, upcast_attention=False, **kw):
            super().__init__()
            assert cross_attention_dim is None
            inner = heads * dim_head
            self.heads, self.scale = heads, dim_head ** -0.5
            self.to_q = nn.Linear(query_dim, inner, bias=bias)
            self.to_k = nn.Linear(query_dim, inner, bias=bias)
            self.to_v = nn.Linear(query_dim, inner, bias=bias)
            self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

        def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
            assert encoder_hidden_states is None
            B, T, _ = hidden_states.shape
            q, k, v = self.to_q(hidden_states), self.to_k(hidden_states), self.to_v(hidden_states)
            d = q.shape[-1] // self.heads
            q, k, v = (t.view(B, T, self.heads, d).transpose(1, 2) for t in (q, k, v))
            s = torch.matmul(q, k.transpose(-1, -2)) * self.scale
            if attention_mask is not None:
                m = attention_mask
                m = m[:, None, None, :] if m.dim() == 2 else m[:, None]
                s = s + m.to(s.dtype)
            o = torch.matmul(torch.softmax(s, dim=-1), v).transpose(1, 2).reshape(B, T, self.heads * d)
            return self.to_out[1](self.to_out[0](o))


    class RotaryEmbedding(nn.Module):
        def __init__(self, dim, base=10000):
            super().__init__()
            inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
            self.register_buffer('inv_freq', inv_freq)

        def forward_from_seq_len(self, seq_len):
            t = torch.arange(seq_len, device=self.inv_freq.device)
            return self.forward(t)

        def forward(self, t):
            if t.ndim == 1:
                t = t[None]
            freqs = torch.einsum('b i , j -> b i j', t.type_as(self.inv_freq), self.inv_freq)
            freqs = torch.stack((freqs, freqs), dim=-1).flatten(-2)   # interleaved duplicate
            return freqs, 1.0

    def rotate_half(x):
        x = x.reshape(*x.shape[:-1], -1, 2)
        x1, x2 = x.unbind(dim=-1)
        return torch.stack((-x2, x1), dim=-1).flatten(-2)

    def apply_rotary_pos_emb(t, freqs, scale=1):
        rot_dim, seq_len, orig_dtype = freqs.shape[-1], t.shape[-2], t.dtype
        freqs = freqs[:, -seq_len:, :]
        if t.ndim == 4 and freqs.ndim == 3:
            freqs = freqs[:, None]
        t_rot, t_pass = t[..., :rot_dim], t[..., rot_dim:]
        t_rot = (t_rot * freqs.cos() * scale) + (rotate_half(t_rot) * freqs.sin() * scale)
        return torch.cat((t_rot, t_pass), dim=-1).type(orig_dtype)

    return dict(Attention=Attention, RotaryEmbedding=RotaryEmbedding, rotate_half=rotate_half, apply_rotary_pos_emb=apply_rotary_pos_emb)


def install():
    import torch
    from torch import nn

    if REF_ROOT + '/server/model_utils' not in sys.path:
        sys.path[:0] = [REF_ROOT + '/server/model_utils', REF_ROOT]

    # transformers must be imported before torchaudio is stubbed
    import transformers.models.qwen2.modeling_qwen2 as mq  # noqa

    if 'torchaudio' not in sys.modules:
        ta = _mod('torchaudio')
        ta.transforms = _mod('torchaudio.transforms')
        ta.compliance = _mod('torchaudio.compliance')
        ta.compliance.kaldi = _mod('torchaudio.compliance.kaldi')

    # ---- Qwen2DecoderLayer call convention of transformers 4.40.1 on the installed 5.x ----------
    if not getattr(mq.Qwen2DecoderLayer, '_hvx_wrapped', False):
        orig_forward = mq.Qwen2DecoderLayer.forward

        def forward(self, hidden_states, *args, **kwargs):
            legacy = kwargs.get('position_embeddings', None) is None and len(args) == 0
            if legacy:
                L = hidden_states.shape[1]
                hd = self.self_attn.head_dim
                cfg = self.self_attn.config
                theta = getattr(cfg, 'rope_theta', None)
                if theta is None:
                    rp = getattr(cfg, 'rope_parameters', None) or {}
                    theta = rp.get('rope_theta', 10000.0)
                inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
                fr = torch.outer(torch.arange(L, dtype=torch.float32), inv)
                emb = torch.cat([fr, fr], dim=-1)[None]
                kwargs['position_embeddings'] = (emb.cos().to(hidden_states.dtype), emb.sin().to(hidden_states.dtype))
            out = orig_forward(self, hidden_states, *args, **kwargs)
            if legacy and not isinstance(out, tuple):
                out = (out,)
            return out

        mq.Qwen2DecoderLayer.forward = forward
        mq.Qwen2DecoderLayer._hvx_wrapped = True

    # ---- dummies for names imported (not used on the DiT/HiFT path) -----------------------------
    class _Dummy(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    _mod('conformer', ConformerBlock=_Dummy)
    _mod('diffusers')
    _mod('diffusers.models')
    _mod('diffusers.models.activations', get_activation=lambda n: {'silu': nn.SiLU(), 'swish': nn.SiLU(), 'mish': nn.Mish(), 'gelu': nn.GELU()}[n])
    _mod('diffusers.models.attention', GEGLU=_Dummy, GELU=_Dummy, AdaLayerNorm=_Dummy, AdaLayerNormZero=_Dummy, ApproximateGELU=_Dummy)
    Attention = stand_ins()['Attention']
    _mod('diffusers.models.attention_processor', Attention=Attention)
    _mod('diffusers.models.lora', LoRACompatibleLinear=nn.Linear)
    _mod('diffusers.utils')
    _mod('diffusers.utils.torch_utils', maybe_allow_in_graph=lambda c: c)

    class DictConfig(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)
    _mod('omegaconf', DictConfig=DictConfig)

    # librosa (absent): matcha/utils/audio.py only needs filters.mel — the oracle's own float64 statement of the Slaney table (oracle/frontend_ref.py:
    # scalar arithmetic pinned to the literal values of librosa's docstrings; it shares no code with the product's packing.mel_filterbank, which
    # tests/test_oracle_golden.py::test_slaney_mel_table_is_pinned_and_the_product_table_equals_it holds to it at 1e-7)
    def _librosa_mel(sr, n_fft, n_mels, fmin, fmax):
        sys.path.insert(0, '/root/repo') if '/root/repo' not in sys.path else None
        from oracle.frontend_ref import slaney_mel_table
        return slaney_mel_table(sr, n_fft, n_mels, fmin, fmax).float().numpy()
    _mod('librosa')
    _mod('librosa.filters', mel=_librosa_mel)

    import matcha  # real package root (namespace only)
    _mod('matcha.utils', __path__=[])
    _mod('matcha.utils.pylogger', get_pylogger=lambda name=None: logging.getLogger(name))

    # ---- x_transformers: restated arithmetic (see module docstring) -------------------------------
    RotaryEmbedding, apply_rotary_pos_emb = stand_ins()['RotaryEmbedding'], stand_ins()['apply_rotary_pos_emb']

    _mod('x_transformers')
    _mod('x_transformers.x_transformers', RotaryEmbedding=RotaryEmbedding, apply_rotary_pos_emb=apply_rotary_pos_emb)
    return DictConfig
