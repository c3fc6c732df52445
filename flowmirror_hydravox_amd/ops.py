"""Operator-level wrappers over the C-ABI (used by the parity tests and by the model classes).
Every function launches HIP kernels from libhvx on the current torch stream; tensors are plain device buffers."""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import GemmArgs, AttnArgs, SampleArgs, ptr, stream_ptr, check, dtype_code


def _pad32(n):
    return (n + 31) // 32 * 32


def conv1d(x, w_packed, bias, *, n_out, taps, cin_pad, pad_left=0, dil=1, stride=1, up=1, m_out=None, groups=1,
           act=_lib.ACT_NONE, act_param=0.0, act_alpha=None, gate=None, res=None, res_row_off=0, scale=1.0,
           out=None, out_dtype=None, out_row_off=0, out2=None, act2=_lib.ACT_NONE, act2_param=0.0, act2_alpha=None, out2_row_off=0, x3=False):
    """Implicit-GEMM Conv1d / Linear on time-major rows.  x: [B][rows_in][lda] (f32 or bf16), w_packed: [groups][n_out][taps*cin_pad]."""
    lib = _lib.load()
    B, rows_in, lda = x.shape
    M = rows_in if m_out is None else m_out
    a = GemmArgs()
    a.dtype = dtype_code(x.dtype)
    a.M, a.N, a.K, a.batch, a.groups = M, n_out, taps * cin_pad, B, groups
    a.A, a.a_bs, a.lda, a.a_gs, a.rows_in = ptr(x), rows_in * lda, lda, (cin_pad if groups > 1 else 0), rows_in
    a.cin_pad, a.conv_stride, a.conv_dil, a.pad_left, a.up = cin_pad, stride, dil, pad_left, up
    a.W, a.w_gs = ptr(w_packed), n_out * taps * cin_pad
    a.bias = ptr(bias)
    a.act, a.act_param, a.act_alpha = act, act_param, ptr(act_alpha)
    if gate is not None:
        a.gate, a.gate_bs = ptr(gate), gate.shape[-1]
    total = groups * n_out
    if res is not None:
        a.res, a.res_bs, a.ldres, a.res_row_off = ptr(res), res.shape[1] * res.shape[2], res.shape[2], res_row_off
        a.res_f16 = int(res.dtype == torch.float16)                 # the DiT's half residual stream (hvx_flow_set_half_stream)
    a.scale = scale
    a.x3 = int(x3)
    if out is None and out2 is None:
        od = out_dtype or torch.float32
        out = torch.zeros(B, M + max(out_row_off, 0), _pad32(total) if od != torch.float32 else total, dtype=od, device=x.device)
    if out is not None:
        a.out, a.out_f32, a.out_bs, a.ldo = ptr(out), int(out.dtype == torch.float32), out.shape[1] * out.shape[2], out.shape[2]
        a.out_f16 = int(out.dtype == torch.float16)
        a.out_row_off, a.out_cols = out_row_off, out.shape[2]
    if out2 is not None:
        assert out2.dtype == x.dtype
        a.out2, a.act2, a.act2_param, a.act2_alpha = ptr(out2), act2, act2_param, ptr(act2_alpha)
        a.out2_bs, a.ldo2, a.out2_row_off, a.out2_cols = out2.shape[1] * out2.shape[2], out2.shape[2], out2_row_off, out2.shape[2]
    check(lib.hvx_op_gemm(C.byref(a), stream_ptr()), 'hvx_op_gemm')
    return out


def attention(q, k, vT, t, *, kv_len=None, causal=False, scale=None, n_splits=1, split_chunk=0, chunk=0, q_log2=False):
    """q, k: [B][H][Tpad][64]; vT: [B][H][64][Tpad]  ->  [B][t][H*64]"""
    lib = _lib.load()
    B, H, Tp, d = q.shape
    assert d == 64 and vT.shape == (B, H, 64, Tp) and Tp % 32 == 0
    out = torch.empty(B, t, H * 64, dtype=q.dtype, device=q.device)
    a = AttnArgs()
    a.dtype, a.batch, a.heads, a.t, a.t_pad = dtype_code(q.dtype), B, H, t, Tp
    a.q, a.k, a.vT, a.out, a.kv_len = ptr(q), ptr(k), ptr(vT), ptr(out), ptr(kv_len)
    a.causal, a.scale = int(causal), (1.0 / math.sqrt(64)) if scale is None else scale
    a.chunk = int(chunk)
    a.q_log2 = int(q_log2)
    keep = []
    if n_splits > 1:
        rp = _pad32(t)
        po = torch.empty(B * H * n_splits * rp * 64, dtype=torch.float32, device=q.device)
        pm = torch.empty(B * H * n_splits * rp * 2, dtype=torch.float32, device=q.device)
        keep = [po, pm]
        a.n_splits, a.split_chunk, a.part_o, a.part_ml = n_splits, split_chunk, ptr(po), ptr(pm)
    else:
        a.n_splits = 1
    check(lib.hvx_op_attention(C.byref(a), stream_ptr()), 'hvx_op_attention')
    return out


def skinny_gemm(x, w_packed, n_out, bias=None, split_k=1):
    """x: [M][K] (f32/bf16), w_packed: pack_frag([N][K]) -> f32 [M][N]"""
    lib = _lib.load()
    M, K = x.shape
    out = torch.empty(M, n_out, dtype=torch.float32, device=x.device)
    part = torch.empty(max(split_k, 1) * M * n_out, dtype=torch.float32, device=x.device)
    check(lib.hvx_op_skinny_gemm(dtype_code(x.dtype), M, n_out, K, ptr(x), K, ptr(w_packed), ptr(bias), split_k, ptr(part), ptr(out),
                                 n_out, stream_ptr()), 'hvx_op_skinny_gemm')
    return out


def ras_sample(logp, hist, hist_len, min_len, noise, cursor, *, speech_tokens, top_k, top_p, win_size, rep_thresh, active=None,
               max_trials=100):
    """logp f32 [S][K][V]; hist int32 [S][Hcap]; hist_len/min_len int32 [S]; noise f32 [S][Ncap]; cursor int64 [S] (updated in place)
    -> ids int32 [S][K]"""
    lib = _lib.load()
    S, K, V = logp.shape
    out = torch.empty(S, K, dtype=torch.int32, device=logp.device)
    a = SampleArgs()
    a.n_seq, a.head_k, a.vocab, a.speech_tokens = S, K, V, speech_tokens
    a.logp, a.logp_seq_stride, a.logp_head_stride = ptr(logp), K * V, V
    a.hist, a.hist_seq_stride, a.hist_len = ptr(hist), hist.shape[1], ptr(hist_len)
    a.min_len, a.active = ptr(min_len), ptr(active)
    a.top_k, a.top_p, a.win_size, a.rep_thresh = top_k, top_p, win_size, rep_thresh
    a.noise, a.noise_seq_stride, a.noise_len = ptr(noise), noise.shape[1], noise.shape[1]
    a.cursor, a.out_ids, a.max_trials = ptr(cursor), ptr(out), max_trials
    check(lib.hvx_ras_sample(C.byref(a), stream_ptr()), 'hvx_ras_sample')
    return out


def resample_linear(x, t_out):
    """F.interpolate(x, size=t_out, mode='linear') for a float32 (..., T) tensor on the device (the `speed` knob of the synthesis calls)"""
    lib = _lib.load()
    x = x.to(torch.float32).contiguous()
    t_in = x.shape[-1]
    rows = x.numel() // max(t_in, 1)
    y = torch.empty(*x.shape[:-1], t_out, dtype=torch.float32, device=x.device)
    check(lib.hvx_op_resample_linear(ptr(x), rows, t_in, ptr(y), t_out, stream_ptr()), 'hvx_op_resample_linear')
    return y
