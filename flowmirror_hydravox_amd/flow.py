"""HvxFlow — drop-in for `CausalMaskedDiffWithDiT` + `CausalConditionalCFM` + `DiT` inference on MI355X.

Mirrors server/model_utils/cosyvoice/flow/flow.py:367-430:
    inference(token, token_len, embedding, finalize, prompt_token=None, prompt_token_len=None, prompt_feat=None,
              prompt_feat_len=None, streaming=False) -> (mel float32 (1, 80, 2*N_tok), None)
with the loader-set attributes `bf16` / `fp16` (infer_speech_model.py:99-124).  The host keeps only plumbing: prompt
concatenation, the zero-padded `cond`, slicing the fixed noise (`set_all_random_seed(0); randn(1,80,15000)`,
flow_matching.py:200-201) and the t-grid `1 - cos(linspace(0,1,11) * pi/2)` accumulated exactly as solve_euler does
(flow_matching.py:71-124, 225-227).  All arithmetic runs in libhvx (csrc/hvx_flow.hip).
"""
import ctypes as C

import os

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .config import FlowConfig
from .packing import conv_weight, grouped_conv_weight, linear_weight
from .weights import flow_spec, check_state, DROP_KEYS


def dit_rope_tables(max_t, head_dim):
    """x_transformers RotaryEmbedding(dim_head).forward_from_seq_len: inv_freq = 1/10000^(2i/d), freqs = pos * inv_freq
    (each duplicated for an interleaved pair) -> cos/sin [max_t][head_dim/2], fp32 on the host."""
    inv = 1.0 / (10000 ** (torch.arange(0, head_dim, 2).float() / head_dim))
    fr = torch.einsum('i,j->ij', torch.arange(max_t).float(), inv)
    return fr.cos().contiguous(), fr.sin().contiguous()


def euler_schedule(n_timesteps):
    """(t_k, dt_k) per step with the reference's accumulation `t = t + dt; dt = t_span[k+1] - t` (flow_matching.py:91-122)."""
    t_span = torch.linspace(0, 1, n_timesteps + 1, dtype=torch.float32)
    t_span = 1 - torch.cos(t_span * 0.5 * torch.pi)
    t, dt = t_span[0], t_span[1] - t_span[0]
    ts, dts = [], []
    for step in range(1, n_timesteps + 1):
        ts.append(float(t))
        dts.append(float(dt))
        t = t + dt
        if step < n_timesteps:
            dt = t_span[step + 1] - t
    return ts, dts


class HvxFlow:
    def __init__(self, cfg: FlowConfig, state_dict=None, dtype=torch.bfloat16, device='cuda', max_t=None, half_stream=None, f16_linears=None, f32_small=None):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.cfg = cfg
        self.dtype = dtype
        self.device = torch.device(device)
        self.bf16 = dtype == torch.bfloat16
        self.fp16 = False
        # bf16 mode: the DiT's residual stream is kept in fp16, the arithmetic the reference itself runs (`flow.eval().cuda().half()`,
        # infer_speech_model.py:103); half_stream=False keeps it in fp32 (the fp32 mode always does).  HVX_FLOW_HALF_STREAM=0/1 overrides.
        if half_stream is None:
            half_stream = os.environ.get('HVX_FLOW_HALF_STREAM', '1') != '0'
        self.half_stream = bool(half_stream) and self.bf16
        # bf16 mode: the four Linears of every DiT block on IEEE fp16 operands (the reference deploys this decoder as `.half()`); q / k / v and the attention
        # stay bf16.  Together with f32_small below the mode sits at the reference's own fp16 distance from fp32 instead of 3.5x it (tools/
        # dit_rounding_study.py, tests/test_gpu_cv3d.py) for 2 % of the flow's time: ON by default where the 256-tile kernel applies (dim >= 256, dim and ff
        # multiples of 64); f16_linears=False / HVX_FLOW_F16_LINEARS=0 turn it off.
        if f16_linears is None:
            f16_linears = os.environ.get('HVX_FLOW_F16_LINEARS', '1') != '0'
        self.f16_linears = bool(f16_linears) and self.bf16 and cfg.dim >= 256 and cfg.dim % 64 == 0 and (cfg.dim * cfg.ff_mult) % 64 == 0
        # ... and the small Linears (time MLP, adaLN modulation, input / output projection) in fp32: with fp16 block Linears they carry what is left of the
        # bf16 mode's distance from fp32.  ON by default (ff >= 2 dim); f32_small=False / HVX_FLOW_F32_SMALL=0 turn it off.
        if f32_small is None:
            f32_small = os.environ.get('HVX_FLOW_F32_SMALL', '1') != '0'
        self.f32_small = bool(f32_small) and self.bf16 and cfg.ff_mult >= 2
        self.token_mel_ratio = cfg.token_mel_ratio
        self.pre_lookahead_len = cfg.pre_lookahead_len
        self.static_chunk_size = cfg.static_chunk_size            # DiT(static_chunk_size=...), dit.py:119,142
        self.output_size = cfg.mel
        self.max_t = max_t or cfg.noise_frames
        g = torch.Generator()
        g.manual_seed(0)
        self.rand_noise = torch.randn([1, cfg.mel, cfg.noise_frames], generator=g).to(self.device)
        self._h = None
        self._ws = None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, sd, strict=True):
        return self.load_packed(self.pack_state_dict(sd))

    def pack_state_dict(self, sd):
        """reference checkpoint -> the device tensors libhvx consumes, in the order include/hvx.h documents"""
        sd = {k: v for k, v in sd.items() if k not in DROP_KEYS}
        check_state(sd, flow_spec(self.cfg), 'CausalMaskedDiffWithDiT', optional=('decoder.estimator.rotary_embed.inv_freq',))
        c, dt, dev = self.cfg, self.dtype, self.device
        e = 'decoder.estimator.'

        def W(k):
            return sd[k].to(dev).float()

        def mat(t):
            return t.to(dt).contiguous()

        def smat(t):                                       # a small Linear of the estimator: fp32 operands when the handle runs them so
            return t.float().contiguous() if self.f32_small else t.to(dt).contiguous()

        def lmat(t):                                       # a DiT block Linear: fp16 operands when the handle runs them so
            return t.to(torch.float16 if self.f16_linears else dt).contiguous()

        def vec(t):
            return t.float().contiguous()

        cos, sin = dit_rope_tables(self.max_t, c.head_dim)
        melp = (c.mel + 31) // 32 * 32
        ws = [cos.to(dev), sin.to(dev), vec(W('input_embedding.weight')), vec(W('spk_embed_affine_layer.weight')),
              vec(W('spk_embed_affine_layer.bias')),
              mat(conv_weight(W('pre_lookahead_layer.conv1.weight'), melp)), vec(W('pre_lookahead_layer.conv1.bias')),
              mat(conv_weight(W('pre_lookahead_layer.conv2.weight'))), vec(W('pre_lookahead_layer.conv2.bias')),
              smat(W(e + 'time_embed.time_mlp.0.weight')), vec(W(e + 'time_embed.time_mlp.0.bias')),
              smat(W(e + 'time_embed.time_mlp.2.weight')), vec(W(e + 'time_embed.time_mlp.2.bias')),
              smat(linear_weight(W(e + 'input_embed.proj.weight'))), vec(W(e + 'input_embed.proj.bias')),
              mat(grouped_conv_weight(W(e + 'input_embed.conv_pos_embed.conv1.0.weight'), c.conv_groups)),
              vec(W(e + 'input_embed.conv_pos_embed.conv1.0.bias')),
              mat(grouped_conv_weight(W(e + 'input_embed.conv_pos_embed.conv2.0.weight'), c.conv_groups)),
              vec(W(e + 'input_embed.conv_pos_embed.conv2.0.bias'))]
        for i in range(c.depth):
            p = e + 'transformer_blocks.%d.' % i
            wqkv = torch.cat([W(p + 'attn.to_q.weight'), W(p + 'attn.to_k.weight'), W(p + 'attn.to_v.weight')], 0)
            bqkv = torch.cat([W(p + 'attn.to_q.bias'), W(p + 'attn.to_k.bias'), W(p + 'attn.to_v.bias')], 0)
            ws += [smat(W(p + 'attn_norm.linear.weight')), vec(W(p + 'attn_norm.linear.bias')), lmat(wqkv), vec(bqkv),
                   mat(W(p + 'attn.to_out.0.weight')), vec(W(p + 'attn.to_out.0.bias')),
                   lmat(W(p + 'ff.ff.0.0.weight')), vec(W(p + 'ff.ff.0.0.bias')), lmat(W(p + 'ff.ff.2.weight')), vec(W(p + 'ff.ff.2.bias'))]
        ws += [smat(W(e + 'norm_out.linear.weight')), vec(W(e + 'norm_out.linear.bias')), smat(W(e + 'proj_out.weight')), vec(W(e + 'proj_out.bias'))]
        return ws

    def load_packed(self, ws):
        c, dt, dev = self.cfg, self.dtype, self.device
        ws = [w.to(dev) for w in ws]
        self._weights = ws
        if self.bf16 and c.depth > 0:
            self.f16_linears = ws[19 + 2].dtype == torch.float16          # (packed weights say how the block Linears were packed)
        if self.bf16:
            self.f32_small = ws[9].dtype == torch.float32
        cc = _lib.FlowConfig(dtype=_lib.dtype_code(dt), vocab=c.vocab, mel=c.mel, spk_dim=c.spk_embed_dim, pla_channels=c.pre_lookahead_channels,
                             pla_len=c.pre_lookahead_len, dim=c.dim, depth=c.depth, heads=c.heads, ff=c.ff, conv_kernel=c.conv_kernel,
                             conv_groups=c.conv_groups, time_freq_dim=c.time_freq_dim, max_t=self.max_t, cfg_rate=c.cfg_rate)
        if self._h is not None:
            self.lib.hvx_flow_destroy(self._h)
        h = C.c_void_p()
        check(self.lib.hvx_flow_create(C.byref(cc), _lib.ptr_array(ws), len(ws), C.byref(h)), 'hvx_flow_create')
        self._h = h
        # adaLN modulation cache: one slot per distinct Euler step time (the t-grid is a model constant)
        slot = (c.depth * 2 * 6 * c.dim + 2 * 2 * c.dim) * 4
        self._mod_cache = torch.zeros(16 * slot, dtype=torch.uint8, device=dev)
        check(self.lib.hvx_flow_set_mod_cache(self._h, ptr(self._mod_cache), self._mod_cache.numel()), 'hvx_flow_set_mod_cache')
        check(self.lib.hvx_flow_set_half_stream(self._h, 1 if self.half_stream else 0), 'hvx_flow_set_half_stream')
        check(self.lib.hvx_flow_set_f16_linears(self._h, 1 if self.f16_linears else 0), 'hvx_flow_set_f16_linears')
        check(self.lib.hvx_flow_set_f32_small(self._h, 1 if self.f32_small else 0), 'hvx_flow_set_f32_small')
        return self

    def eval(self):
        return self

    def cuda(self):
        return self

    def half(self):
        return self

    def to(self, *a, **k):
        return self

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None:
                self.lib.hvx_flow_destroy(self._h)
        except Exception:
            pass

    def _workspace(self, batch, t):
        need = self.lib.hvx_flow_workspace_bytes(self._h, batch, t)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    # ---- stage entry points (also used by the parity tests) ----------------------------------------------------------------
    def prelookahead(self, x, context=False):
        """x f32 [n][mel] -> [n][mel] (upsample_encoder.py:82-103); context=True: the last pre_lookahead_len rows are the look-ahead
        context of the rest (`forward(inputs[:-len], context=inputs[-len:])`) -> [n - len][mel]"""
        x = x.to(self.device, torch.float32).contiguous()
        n = x.shape[0]
        ws = self._workspace(2, max(n, 32))
        if context:
            y = torch.empty(n - self.pre_lookahead_len, x.shape[1], dtype=torch.float32, device=self.device)
            check(self.lib.hvx_flow_prelookahead_context(self._h, stream_ptr(), ptr(ws), ws.numel(), ptr(x), n, ptr(y)),
                  'hvx_flow_prelookahead_context')
            return y
        y = torch.empty_like(x)
        check(self.lib.hvx_flow_prelookahead(self._h, stream_ptr(), ptr(ws), ws.numel(), ptr(x), n, ptr(y)), 'hvx_flow_prelookahead')
        return y

    def encode(self, token, embedding, finalize=True):
        """token int [n] (prompt already prepended), embedding f32 [spk_dim] -> mu (mel, 2n), spk (mel,); finalize=False: the last
        pre_lookahead_len tokens are context only (flow.py:401-404) -> mu (mel, 2 (n - len))"""
        token = token.to(self.device, torch.int32).contiguous().view(-1)
        emb = embedding.to(self.device, torch.float32).contiguous().view(-1)
        n = token.numel()
        n_out = n if finalize else n - self.pre_lookahead_len
        if n_out <= 0:
            raise ValueError('a non-final chunk needs more than %d tokens' % self.pre_lookahead_len)
        mu = torch.empty(self.cfg.mel, self.token_mel_ratio * n_out, dtype=torch.float32, device=self.device)
        spk = torch.empty(self.cfg.mel, dtype=torch.float32, device=self.device)
        ws = self._workspace(2, max(2 * n, 32))
        check(self.lib.hvx_flow_encode_chunk(self._h, stream_ptr(), ptr(ws), ws.numel(), ptr(token), n, ptr(emb), 1 if finalize else 0, ptr(mu),
                                             ptr(spk)), 'hvx_flow_encode_chunk')
        return mu, spk

    def estimator(self, x, mask, mu, t, spks, cond, streaming=False):
        """TensorRT-order estimator call (flow_matching.py:126-153): x, mu, cond (B,80,T); mask (B,1,T); t (B,); spks (B,80);
        streaming=True adds the static chunk mask (dit.py:163-164)."""
        B, mel, T = x.shape
        f = lambda a: a.to(self.device, torch.float32).contiguous()
        x, mu, cond, spks, t = f(x), f(mu), f(cond), f(spks), f(t).view(-1)
        if t.numel() == 1 and B > 1:
            t = t.repeat(B)
        kv_len = None
        if mask is not None:
            kv_len = mask.to(self.device).reshape(B, T).ne(0).sum(dim=1).to(torch.int32).contiguous()
        out = torch.empty(B, mel, T, dtype=torch.float32, device=self.device)
        ws = self._workspace(B, T)
        check(self.lib.hvx_cfm_estimator_streaming(self._h, stream_ptr(), ptr(ws), ws.numel(), B, T, ptr(x), ptr(kv_len), ptr(mu), ptr(t),
                                                   ptr(spks), ptr(cond), self.static_chunk_size if streaming else 0, ptr(out)),
              'hvx_cfm_estimator_streaming')
        return out

    def solve(self, mu, spks, cond, n_timesteps=None, noise=None, streaming=False):
        """CausalConditionalCFM.forward (flow_matching.py:204-228): mu, cond (mel,T) f32; spks (mel,) -> mel (mel,T)"""
        T = mu.shape[-1]
        n = n_timesteps or self.cfg.n_timesteps
        z = (self.rand_noise if noise is None else noise.to(self.device))[0, :, :T].to(torch.float32).contiguous().clone()
        ts, dts = euler_schedule(n)
        ta = (C.c_float * n)(*ts)
        da = (C.c_float * n)(*dts)
        ws = self._workspace(2, T)
        check(self.lib.hvx_cfm_solve_streaming(self._h, stream_ptr(), ptr(ws), ws.numel(), T, ptr(z), ptr(mu.contiguous()),
                                               ptr(spks.contiguous()), ptr(cond.contiguous()), n, ta, da,
                                               self.static_chunk_size if streaming else 0), 'hvx_cfm_solve_streaming')
        return z

    def solve_batch(self, mus, spks, conds, n_timesteps=None, streaming=False):
        """Padded multi-utterance solve (hvx_cfm_solve_batch): lists of mu (mel, T_i), spk (mel,), cond (mel, T_i) -> list of mel (mel, T_i).
        The utterances are padded to the longest one; every one draws the same noise prefix the reference gives a single utterance
        (`rand_noise[:, :, :T_i]`, flow_matching.py:222) and key-padding masks keep the padding out of its attention."""
        n_utt = len(mus)
        lens = [int(m.shape[-1]) for m in mus]
        T = max(lens)
        mel, dev = self.cfg.mel, self.device
        n = n_timesteps or self.cfg.n_timesteps
        x = torch.zeros(n_utt, mel, T, dtype=torch.float32, device=dev)
        mu = torch.zeros(n_utt, mel, T, dtype=torch.float32, device=dev)
        cond = torch.zeros(n_utt, mel, T, dtype=torch.float32, device=dev)
        for i in range(n_utt):
            x[i, :, :lens[i]] = self.rand_noise[0, :, :lens[i]]
            mu[i, :, :lens[i]] = mus[i]
            cond[i, :, :lens[i]] = conds[i]
        spk = torch.stack([s_.to(dev, torch.float32).view(-1) for s_ in spks]).contiguous()
        t_len = None if min(lens) == T else torch.tensor(lens, dtype=torch.int32).pin_memory().to(dev, non_blocking=True)
        ts, dts = euler_schedule(n)
        ta = (C.c_float * n)(*ts)
        da = (C.c_float * n)(*dts)
        ws = self._workspace(2 * n_utt, T)
        check(self.lib.hvx_cfm_solve_batch(self._h, stream_ptr(), ptr(ws), ws.numel(), n_utt, T, ptr(t_len), ptr(x), ptr(mu), ptr(spk), ptr(cond), n, ta, da,
                                           self.static_chunk_size if streaming else 0), 'hvx_cfm_solve_batch')
        return [x[i, :, :lens[i]] for i in range(n_utt)]

    @torch.inference_mode()
    def inference_batch(self, tokens, embeddings, prompt_tokens=None, prompt_feats=None, streaming=False):
        """Several utterances through ONE solve (length-bucketed acoustic batches; the reference's `inference` asserts
        batch 1, flow.py:387).  tokens: list of int tensors (n_i,); embeddings: list of (spk_dim,); prompt_tokens / prompt_feats: lists with
        None where an utterance has no prompt.  Returns a list of mels (1, mel, 2 n_i), each what `inference` returns for that utterance."""
        n_utt = len(tokens)
        mus, spks, conds, cut = [], [], [], []
        for i in range(n_utt):
            tok = tokens[i].reshape(-1)
            pt = None if prompt_tokens is None else prompt_tokens[i]
            pf = None if prompt_feats is None else prompt_feats[i]
            if pt is not None:
                tok = torch.concat([pt.reshape(-1).to(tok.device), tok])
            mu, spk = self.encode(tok, embeddings[i].reshape(-1), finalize=True)
            cond = torch.zeros(self.cfg.mel, mu.shape[-1], dtype=torch.float32, device=self.device)
            n1 = 0
            if pf is not None:
                pf = pf.reshape(-1, self.cfg.mel)
                n1 = pf.shape[0]
                cond[:, :n1] = pf.to(self.device, torch.float32).t()
            mus.append(mu)
            spks.append(spk)
            conds.append(cond)
            cut.append(n1)
        feats = self.solve_batch(mus, spks, conds, streaming=streaming)
        return [f[:, c:].unsqueeze(0).float() for f, c in zip(feats, cut)]

    # ---- reference surface ---------------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def inference(self, token, token_len, embedding, finalize=True, prompt_token=None, prompt_token_len=None, prompt_feat=None,
                  prompt_feat_len=None, streaming=False):
        assert token.shape[0] == 1                                       # flow.py:387
        if prompt_token is not None and prompt_token_len is not None:
            token = torch.concat([prompt_token.to(token.device), token], dim=1)
        mu, spk = self.encode(token[0], embedding.reshape(-1), finalize=finalize)
        T = mu.shape[-1]
        mel_len1 = 0
        cond = torch.zeros(self.cfg.mel, T, dtype=torch.float32, device=self.device)
        if prompt_feat is not None and prompt_feat_len is not None:
            mel_len1 = prompt_feat.shape[1]
            cond[:, :mel_len1] = prompt_feat[0].to(self.device, torch.float32).t()
        feat = self.solve(mu, spk, cond, streaming=streaming)
        feat = feat[:, mel_len1:]
        assert feat.shape[1] == T - mel_len1
        return feat.unsqueeze(0).float(), None
