// hvx_kernels.h — host-side launch interface of the HIP kernels (internal to libhvx).
// Every launcher enqueues on the caller's stream, allocates nothing and never synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hvx_options.h"

namespace hvx {

// DT_F16: operands of the DiT block Linears (256-tile form only) and the activations feeding them, when the flow handle runs them in fp16
enum DType : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };
inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }

void set_error(const char* fmt, ...);

// ---- sampled per-kernel timing (bench.py: roofline.achieved) -----------------------------------------------------------
// When enabled, every `period`-th launch of a kernel class is bracketed by hipEvents on its own stream.
enum ProfKind : int { PK_SKINNY = 0, PK_GEMM = 1, PK_ATTN = 2, PK_SAMPLER = 3, PK_GEMM_F32 = 4, PK_ATTN_LLM = 5, PK_COUNT = 6 };
int prof_begin(int kind, double work, hipStream_t s);     // returns a slot (>= 0) or -1 when this launch is not sampled
void prof_end(int slot, hipStream_t s);
bool prof_enabled();

// ------------------------------------------------------------------------------------------------
// Tiled implicit-GEMM:  out[b, m, g*N + n] = epi( sum_{tap, ci} A[b, src(m, tap), g*a_gs + ci] * W[g, n, tap*cin_pad + ci] )
//   src(m, tap) = floor((m*conv_stride + tap*conv_dil - pad_left) / up), zero outside [0, rows_in*up)
// A, W are `dtype` (bf16 or f32), both K-contiguous ("NT" GEMM); accumulation is fp32 on MFMA.
// A plain Linear is taps == 1 (cin_pad == K).  Covers every Conv1d flavour of the path:
// CausalConv1d left/right (pad_left), dilation, CausalConv1dDownSample (conv_stride),
// CausalConv1dUpsample (up = nearest-neighbour factor), grouped conv (groups, a_gs).
// ------------------------------------------------------------------------------------------------
enum Epi : int { EPI_GENERIC = 0, EPI_QKV_DIT = 1 };

struct GemmArgs {
    int dtype;                    // DType of A and W
    int M, N, K;                  // per (batch, group); K % 32 == 0
    int batch, groups;
    const void* A; long long a_bs; int lda; int a_gs; int rows_in;
    int cin_pad, conv_stride, conv_dil, pad_left, up;
    const void* W; long long w_gs;            // [groups][N][K]
    int epi;
    // ---- EPI_GENERIC: v = act(acc + bias[gc]) * gate[b, gc] + res[b, m, gc] + res2[b, m, gc]; v *= scale
    //      out[b, m + out_row_off, gc] = v (f32 or dtype); out2[b, m + out2_row_off, gc] = dtype(act2(v));  gc = g*N + n
    //      (res is read at row m + res_row_off; rows whose shifted index is negative are skipped)
    const float* bias;
    int act; float act_param; const float* act_alpha;
    const float* gate; long long gate_bs;
    const float* res; long long res_bs; int ldres; int res_row_off;
    const float* res2; long long res2_bs; int ldres2;
    float scale; float div;       // v *= scale; if (div != 0) v /= div
    void* out; int out_f32; long long out_bs; int ldo; int out_row_off; int out_cols;   // cols [groups*N, out_cols) zero-filled
    void* out2; int act2; float act2_param; const float* act2_alpha; long long out2_bs; int ldo2; int out2_row_off; int out2_cols;
    // ---- EPI_QKV_DIT: N = 3*heads*64; + bias; interleaved-pair RoPE on channels [0,64) of q and k (head 0);
    //      q,k -> [b][head][t_pad][64], v -> vT [b][head][64][t_pad]   (all `dtype`)
    void* q; void* k; void* vT; int heads; int t_pad; const float* rope_cos; const float* rope_sin;   // [t][32]
    int x3;                       // fp32 operands may be split into bf16 pairs and multiplied on the bf16 matrix cores (gemm_x3.hip; ~1e-6 relative)
    float q_scale;                // != 0: q is stored multiplied by this (softmax scale * log2 e: the scores then leave the attention MFMAs in log2 units)
    // ---- fp32 values as (hi, lo) bf16 PLANE PAIRS (dtype DT_F32, x3 path): x == hi + lo up to 2^-17 |x|, hi = bf16(x), lo = bf16(x - hi).
    //      A pair is addressed like the fp32 tensor it stands for (same row / batch strides, in elements), the lo plane `*_plane` bf16
    //      elements behind the hi plane.  With a_planes and w_planes the split-bf16 GEMM reads its operands straight into LDS by LDS-DMA
    //      (gemm_x3.hip: gemm_x3p_kernel) instead of splitting fp32 values on the way in; out_planes / out2_planes make the epilogue
    //      write a pair instead of fp32 values (any fp32 GEMM form), so the next convolution finds its operand ready.
    int a_planes; long long a_plane;
    int w_planes; long long w_plane;
    int out_planes; long long out_plane;
    int out2_planes; long long out2_plane;
    // ---- HALF residual stream (DiT, bf16 mode): res and / or out hold IEEE fp16 instead of fp32 (same strides, in elements).  The reference runs
    //      its flow decoder in fp16 end to end (infer_speech_model.py:103), i.e. with an fp16 residual stream; here only the stream is fp16 — the
    //      update gate * (acc + bias) + res is formed in fp32 and rounded once on the way out.
    int res_f16; int out_f16;
};
int launch_gemm(const GemmArgs& a, hipStream_t s);
int launch_gemm_x3(const GemmArgs& a, hipStream_t s);    // same convention; fp32 operands on the bf16 matrix cores (GemmArgs.x3)
int launch_conv_resident(const GemmArgs& a, hipStream_t s);   // same convention; bf16 k-tap convolutions over 64 channels per group, input rows resident in LDS (conv_resident.hip)
int launch_gemm_big(const GemmArgs& a, hipStream_t s);   // 1 = launched, 0 = not eligible (launch_gemm falls through), -1 = error

// ------------------------------------------------------------------------------------------------
// Skinny weight-streaming GEMM for the AR decode step:  out[m, n] = sum_k A[m, k] * W[n, k], M <= 128 per launch chunk.
// W is pre-packed in MFMA fragment order [N/16][K/32][64 lanes][8] so that every wave-level load is one
// contiguous 1 KiB (bf16) / 2 KiB (f32) burst.  A is row-major [M][lda].  Optional split-K writes fp32 partials.
// ------------------------------------------------------------------------------------------------
enum SkinnyEpi : int {
    SK_PARTIAL = 0,     // part[ks][m][n] = acc                                   (split-K, reduced by llm_reduce_norm)
    SK_STORE = 1,       // out[m][n] = acc + bias[n]           (f32 or dtype)
    SK_QKV_ROPE = 2,    // + bias, rotate-half RoPE, q -> qbuf, k/v -> KV cache     (llm step)
    SK_SWIGLU = 3,      // n-tiles alternate gate/up: out[m][n/2] = silu(gate) * up  (dtype)
    SK_RESID = 4,       // out[m][n] += acc + bias  (fp32 residual stream, in place; split_k == 1)
};
struct SkinnyArgs {
    int dtype;
    int M, N, K;                  // N % 16 == 0, K % 32 == 0
    const void* A; int lda;
    const void* W;                // packed
    int split_k;                  // >= 1
    int epi;
    const float* bias;
    float* part;                  // [split_k][M][N]
    void* out; int out_f32; int ldo;
    // SK_QKV_ROPE (rows are a dense [n_seq][kn] grid: m = si*kn + lt)
    int kn, q_heads, kv_heads;    // head_dim fixed at 64
    const int* slot; const int* pos0; const int* n_new;          // device [n_seq]
    const float* rope_cos; const float* rope_sin;                // [max_pos][32]
    void* qbuf;                   // [n_seq][kn][q_heads*64]              (dtype)
    void* kcache; void* vTcache;  // [slot][kv_heads][max_ctx][64] / [slot][kv_heads][64][max_ctx]   (dtype)
    int max_ctx;
    int kv_frag;                  // the caches are in fragment order (hvx_device.h: frag_index(pos, d, 2) / vfrag_index(pos, d) per (slot, kv head)); max_ctx % 32 == 0
    // layer-batched launches (MTP heads): blockIdx.z = head j, all pointers advance by these strides
    int nz; long long w_zs; long long a_zs; long long bias_zs; long long out_zs; long long part_zs;
    int n_valid;                  // SK_STORE: columns >= n_valid are not stored (0 = N)
    // fused RMSNorm (QKV / gate-up): A is the residual stream [M][lda] (dtype), W has the norm gain folded into its columns and the
    // accumulator is scaled by rsqrt(mean(x^2) + eps) per row before the epilogue
    int a_norm; float norm_eps;
    // SK_RESID: optional copy of the updated rows in `dtype` (the A operand of the next fused-norm GEMM)
    void* out2; int ldo2;
    int w_narrow;                 // SK_RESID: W is packed [N/4][K/128][64][8] (packing.pack_narrow4) for the 4-column workgroup form
    // w_narrow, one row tile (o_proj of a narrow decode grid): A is not read — its fragments are built from the key-split partials of the
    // attention launch in front (AttnArgs.part_o / part_ml with skip_combine), with the arithmetic of attn_combine_kernel, so the combine
    // launch between the two disappears.  Row m = si * kn + lt, column = q head * 64 + d; kn, q_heads, kv_heads, pos0, n_new as above.
    const float* att_o; const float* att_ml; const int* att_kvlen;
    int att_splits, att_chunk, att_rows_pad;
    // ---- wide decode grids (33..128 rows, bf16): activations in fragment order (hvx_device.h: frag_index) -----------------------
    int a_frag;                   // A is [ceil(M/16)][K/32][64][8] instead of row-major (launch_dec_gemm only)
    int out_frag;                 // SK_SWIGLU: out, SK_RESID: out2 are written in fragment order (their consumer is another launch_dec_gemm)
    // launch_dec_gemm, SK_SWIGLU: W holds e4m3 codes in double-step fragment order (packing.pack_frag_fp8; w_zs in codes) and w_scale the
    // power-of-two scale of every output column (row of W), [nz][N] with stride w_scale_zs
    int w_fp8; const float* w_scale; long long w_scale_zs;
};
int launch_skinny(const SkinnyArgs& a, hipStream_t s);
// The same GEMMs for 33..128 rows in the A-stationary / weight-ring form (gemm_dec.hip).  Returns 1 when the launch was taken, 0 when the
// shape is not one of its instantiations (the caller then uses launch_skinny on row-major activations), -1 on error.
int launch_dec_gemm(const SkinnyArgs& a, hipStream_t s);
bool dec_gemm_shape_ok(int M, int N, int K, int epi, int split_k);

// ------------------------------------------------------------------------------------------------
// Attention (flash-style online softmax, head_dim 64, one wave per (q-tile, key-split)).
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
    int dtype;
    int batch, heads;             // grid.z, grid.y  (heads = kv heads when GQA-packed)
    int n_rows;                   // query rows per (batch, head) = rows_hi * kn
    int kn;                       // r -> (r_hi, r_lo) = (r / kn, r % kn)
    const void* q; long long q_bs, q_hs, q_hi, q_lo;
    const void* k; long long k_bs, k_hs;        // [.][.][key][64]
    const void* vT; long long v_bs, v_hs; int v_ld;   // [.][.][64][v_ld]
    const int* kv_slot;           // optional: batch b reads cache slot kv_slot[b]
    const int* kv_len;            // optional device [batch]; else kv_len_const
    int kv_len_const;
    int causal; const int* pos0;  // key j visible to row r iff j <= pos0[b] + r_lo  (pos0 may be null -> 0)
    const int* n_valid_lo;        // optional: rows with r_lo >= n_valid_lo[b] are skipped (inactive)
    int chunk;                    // > 0: static chunk mask (cosyvoice/utils/mask.py:128-158): row r also needs j < (r_lo / chunk + 1) * chunk
    float scale;
    int q_log2;                   // q already multiplied by scale * log2(e) (fused QKV epilogue, GemmArgs.q_scale): scale is ignored
    void* out; long long o_bs, o_hs, o_hi, o_lo;      // dtype
    int n_splits; int split_chunk;                    // keys per split (multiple of 32) when n_splits > 1
    int n_sub;                                        // waves per workgroup of that form: 4 (0 = 4), or 8 (bf16, fragment-order caches, n_rows <= 16)
    int sub_chunk;                                    // > 0 (needs n_rows <= 32): one workgroup per split, its n_sub waves take sub_chunk keys
                                                      // each and merge their (m, l, o) in LDS, so the combine reads 4x fewer partials
    float* part_o; float* part_ml;                    // [batch][heads][n_splits][n_rows_pad][64], [...][n_rows_pad][2]
    int n_rows_pad;
    int skip_combine;                                 // n_splits > 1: leave the partials to the consumer (SkinnyArgs.att_o), `out` is not written
    int kv_frag;                                      // k / vT are the LLM's fragment-order caches (hvx_device.h: frag_index(key, d, 2), vfrag_index(key, d)
                                                      // per (batch, head) block of k_hs / v_hs elements); v_ld = the cache's context capacity
    int o_frag_kt;                                    // > 0 (LLM decode, 16-bit): out is the [batch * kn][heads * rows_hi * 64] activation matrix in fragment
                                                      // order (hvx_device.h: frag_index) with this many k-steps per row; the o_* strides are ignored
};
int launch_attention(const AttnArgs& a, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// Small fused kernels
// ------------------------------------------------------------------------------------------------
// x[m,:] += sum_s part[s][m][:] (+ bias);  y[m,:] = dtype( rmsnorm(x[m,:]) * g )          (any of part/g may be null)
//   rows are grouped in blocks of rows_per_z (MTP heads): block z uses part + z*part_zs, gain + z*gain_zs, bias + z*bias_zs.
//   do_norm == 0 -> y = dtype(x) (plain cast).
struct ReduceNormArgs {
    float* x; int ldx;
    const float* part; int split_k; long long part_stride; long long part_zs;
    const float* bias; long long bias_zs;
    const float* gain; long long gain_zs; float eps; int do_norm;
    void* y; int ldy; int dtype;
    int M, H, rows_per_z;
    int y_frag;                   // y (16-bit) is written in fragment order (hvx_device.h: frag_index, KT = H / 32)
    int y_frag_zrows;             // > 0: every z block is its own fragment-order matrix of this many (multiple of 16) rows: row = z * y_frag_zrows + m % rows_per_z
};
int launch_reduce_rmsnorm(const ReduceNormArgs& a, hipStream_t s);
// y[r, c] = dtype(act(x[r, c])) for a [rows][cols] f32 matrix (per-column alpha for Snake)
int launch_act_rows(const float* x, int ldx, void* y, int ldy, int dtype, int act, float param, const float* alpha, long long rows, int cols,
                    hipStream_t s);
int launch_act_rows_planes(const float* x, int ldx, void* y, int ldy, long long plane, int act, float param, const float* alpha, long long rows, int cols, hipStream_t s);
// gather rows: y[r,:] = x[idx[r],:] (idx[r] < 0 -> zeros)
int launch_heads_prologue(const float* x, int ldx, const int* idx, const float* gain_f, float eps_f, const float* gain_h, float eps_h, int K,
                          int S, int H, float* ylast, float* hx, void* ha, int dtype, hipStream_t s);
int launch_gather_rows_f32(const float* x, int ldx, const int* idx, float* y, int ldy, int rows, int H, hipStream_t s);
// embedding: x[r,:] = table[tok[r],:] as f32 (tok < 0 -> zeros); table dtype f32/bf16
int launch_embed(const void* table, int table_dtype, const int* tok, float* x, int ldx, int rows, int H, hipStream_t s);
// LLM input rows: tok >= 0 -> speech[tok]; tok <= -2 -> text[-tok-2]; tok == -1 -> zeros   (llm_multi_head_v3.py:941-952)
//   copy_frag: x_copy (16-bit) is written in fragment order (hvx_device.h: frag_index)
int launch_embed2(const void* speech, const void* text, int dtype, const int* tok, float* x, int ldx, void* x_copy, int rows, int H, hipStream_t s,
                  int copy_frag = 0);
// logits -> log_softmax (fp32, in place) over V columns, one block per row
int launch_log_softmax(float* x, int ld, int rows, int V, hipStream_t s);
// DiT: y = dtype( LN(x; eps, no affine) * (1 + scale[b]) + shift[b] ); x f32 [B][T][D]; shift/scale f32 [B][.] with stride mod_bs
int launch_layernorm_mod(const void* x, int x_f16, const float* shift, const float* scale, long long mod_bs, float eps, void* y, int dtype,
                         int B, int T, int D, hipStream_t s);
// DiT input: y[b,t,:] = dtype( cat[x[b,:,t], cond[b,:,t], mu[b,:,t], spk[b,:]] ), inputs f32 channel-major (B,80,T)
int launch_dit_concat(const float* x, const float* cond, const float* mu, const float* spk, void* y, int dtype, int B, int T, int mel,
                      hipStream_t s);
// sinusoidal time embedding (DiT/modules.py:71-83): y[b,:] = dtype([sin(1000 t e_i), cos(1000 t e_i)])
int launch_time_sinus(const float* t, void* y, int dtype, int B, int dim, hipStream_t s);
// CFG + Euler (flow_matching.py:116-120): x += dt * ((1+r) * v[0] - r * v[1]); v is [2][T][ldv] row-major f32, x is (80,T) f32
int launch_cfg_euler(float* x, const float* v, int ldv, long long v_bs, float dt, float rate, int T, int mel, hipStream_t s);
// generic strided f32 copy / cast helpers
int launch_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long long n, hipStream_t s);

// ---- Matcha / HiFi-GAN v1 family ----------------------------------------------------------------------
struct PackRowsArgs {
    const float* src[4]; int channels[4]; int broadcast[4];     // source k: (B, C_k, T) channel-major, or (B, C_k) broadcast over T
    float* dst; int ld; int T;                                   // dst [B][T][ld], columns beyond the sources zero-filled
};
int launch_pack_rows(const PackRowsArgs& a, int B, hipStream_t s);
size_t groupnorm_ws_bytes(int B, int T, int G);
// y = (act(GroupNorm_G(x * mask)) + tbias[b][c]) * mask, mask = [t < len[b]]; x, y f32 time-major [B][T][ld]; ws >= groupnorm_ws_bytes
int launch_groupnorm_act(const float* x, int ld, int B, int T, int C, int G, const int* len, const float* gamma, const float* beta, float eps,
                         const float* tbias, int act, float* y, int ldy, void* ws, hipStream_t s);
int launch_mask_rows(float* x, int ld, int B, int T, int C, const int* len, hipStream_t s);
int launch_euler_rows(float* x, const float* v, int ldv, float dt, int B, int T, int mel, hipStream_t s);
int launch_reflect_pad(const float* x, float* y, int L, int pad, int total, hipStream_t s);
int launch_spectral_magnitude(const float* spec, int ld, int frames, int bins, float* mag, int ld_mag, float eps, hipStream_t s);
int launch_spectral_subtract(float* spec, int ld, int frames, int bins, const float* bias, float strength, hipStream_t s);
int launch_feature_post(float* x, int ld, int rows, int cols, int post, hipStream_t s);   // 1: whisper dynamic range + affine, 2: subtract column means
int launch_overlap_add(const float* frames_buf, int ld, int frames, int n_fft, int hop, const float* wsq, float* y, int out_len, hipStream_t s);
int launch_resample_linear(const float* x, int rows, int t_in, float* y, int t_out, hipStream_t s);   // F.interpolate(mode='linear') along the last axis
int launch_transpose_f32(const float* src, float* dst, int rows, int cols, int ld_src, int ld_dst, hipStream_t s);   // dst[c][r] = src[r][c]
int launch_rows_to_dtype(const float* src, int ld_src, void* dst, int dst_dtype, int ld_dst, int rows, int cols, int cols_pad, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// Sampler (common.py:138-166 + llm_multi_head_v3.py:151-166, 890-900)
// ------------------------------------------------------------------------------------------------
struct SampleArgs {
    int n_seq, head_k, V, Vs;               // V = vocab (speech + stop ids), Vs = speech_token_size
    const float* logp; long long logp_ss, logp_hs;     // [n_seq][head_k][V]
    const int* hist; long long hist_ss; const int* hist_len;     // token history per seq (snapshot = first hist_len[s])
    const int* min_len;                     // per seq: head j ignores EOS while hist_len + j < min_len
    const int* active;                      // optional per seq (0 -> skip)
    int top_k; float top_p; int win_size; int rep_thresh;   // rep_thresh = ceil(win_size * tau_r) computed on the host in double
    const float* noise; long long noise_ss; int noise_len;  // Exp(1) stream per seq: ring of noise_len values, index = position % noise_len
    const long long* noise_limit;           // optional per seq: positions < limit are valid (absent: the plain window [0, noise_len))
    long long* cursor;                      // in/out per seq: next unread noise value
    int* out_ids;                           // [n_seq][head_k]; -1 = max_trials exhausted, -2 = noise exhausted (cursor unchanged)
    int max_trials;
};
int launch_ras_sample(const SampleArgs& a, hipStream_t s);

// Between two decode steps (llm_multi_head_v3.py:890-905 and the feedback of inference_wrapper): accept the sampled ids, append
// them to the utterance and the repetition window, decide whether the sequence stops, and write the next step's tok / ctrl.
struct AdvanceArgs {
    int n_seq, head_k, win_cap, max_out, speech_tokens;
    const int* ids;                         // [n_seq][head_k] from the sampler
    int* tok; int* ctrl;                    // next step's [n_seq][head_k] tokens and [5][n_seq] control
    int* hist; int* hist_len; int* min_adj; int* active;
    int* seq_state;                         // [n_seq][8]: pos, out_len, done, min_len, max_len, steps, err, -
    int* out_tokens;                        // [n_seq][max_out]
};
int launch_decode_advance(const AdvanceArgs& a, hipStream_t s);
int launch_decode_join(const AdvanceArgs& a, long long* cursor, int slot, int first_tok, int pos, int min_len, int max_len, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// HiFT source / STFT / iSTFT
// ------------------------------------------------------------------------------------------------
// frame-level harmonic phases (generator.py:233-260): phase[t][h] = 2*pi*480*cumsum_t( (f0[t]*(h+1)/sr) % 1 )
int launch_hift_phase(const float* f0, float* phase, int T, int H, float sr, int up, hipStream_t s);
// per-sample source (generator.py:289-317, 358-375): s[n] = tanh( sum_h w[h]*(sin(phase[t][h])*amp*uv + namp*table[n][h]) + b )
int launch_hift_source(const float* f0, const float* phase, const float* table, const float* w, const float* b, float* s_out,
                       int T, int H, int up, float amp, float sigma, float vthr, hipStream_t s);
// STFT n_fft=16 hop=4 hann, center/reflect (generator.py:491-497): spec[f][0..8]=re, [9..17]=im, row stride ld (>=18, rest zero)
int launch_hift_stft(const float* s_in, float* spec, int L, int ld, hipStream_t s);
// x[f][0..8] -> mag=exp clip 1e2, x[f][9..17] -> phase=sin; iSTFT; clamp (generator.py:702-710, 499-505)
int launch_hift_istft(const float* x, int ld, float* wav, int frames, float limit, hipStream_t s);
// reflection pad (1,0) on a time-major buffer: row0 = row2 (generator.py:686-687)
int launch_copy_row(float* buf, int ld, int dst_row, int src_row, int cols, hipStream_t s);

}  // namespace hvx
