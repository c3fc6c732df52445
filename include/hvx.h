/* hvx.h — C ABI of libhvx, the MI355X-native (gfx950) HydraVox speech-synthesis hot path.
 *
 * The reference (jingzhunxue/FlowMirror_HydraVox) has no native boundary: the path sits behind Python
 * objects (`llm.inference`, `flow.inference`, `hift.inference`, server/model_utils/infer_speech_model.py
 * :549-595).  Its one native-style precedent is the TensorRT estimator call of
 * server/model_utils/cosyvoice/flow/flow_matching.py:126-153 — raw device pointers in a fixed order
 * (x, mask, mu, t, spks, cond -> x), executed on the caller's stream — and this ABI follows that
 * template for every stage:
 *   - plain pointers and sizes, no torch types; every buffer is owned by the caller;
 *   - the callee never allocates device memory and never synchronises the stream;
 *   - all entry points return 0 on success, non-zero on failure; hvx_last_error() gives the message;
 *   - one handle per process (= per GPU), calls serialised by the single-threaded worker loop
 *     (server/worker.py:54), hence no internal locking and no global mutable state besides the error text.
 * Each entry point names the reference interface it replaces.
 */
#ifndef HVX_H
#define HVX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HVX_ABI_VERSION 4
#define HVX_F32 0
#define HVX_BF16 1

typedef void* hvx_stream;                 /* hipStream_t */

int hvx_abi_version(void);
const char* hvx_last_error(void);
/* number of compute units / whether a gfx950 device is current (0 = no usable device) */
int hvx_device_ok(void);

/* Sampled per-kernel timing for bench.py (diagnostics, not on the product path): every `period`-th launch of a kernel class
 * (0 decode GEMM, 1 tiled GEMM/conv, 2 attention, 3 sampler) is bracketed by hipEvents on its stream; period <= 0 disables.
 * `work` is the algorithmic bytes (class 0) or flops (classes 1, 2) of the launches. */
/* A HIP stream whose kernels run on a subset of the compute units (hipExtStreamCreateWithCUMask): CUs [first_cu, first_cu + n_cus) of the
 * driver's enumeration, which deals consecutive indices round-robin over the 8 XCDs, so a contiguous range is spread evenly over them.
 * The decode engine and the acoustic stage get disjoint ranges (pipeline.py: lm_cus): the ~150 short dependent launches of a decode step
 * then never queue behind 200-900 us workgroups of the DiT GEMMs / attention.  Destroy with hvx_stream_destroy.  Returns 0 or -1. */
int hvx_stream_create_cu_range(int32_t first_cu, int32_t n_cus, hvx_stream* out);
int hvx_stream_destroy(hvx_stream s);

/* Run-time options: the ONE switchboard of the library (csrc/hvx_options.h lists them; there are no environment variables).  An option picks between two forms of
 * the same contract or tunes a launch geometry; it is read when a launch is decided (a captured step graph keeps what it was captured with).  Unknown key, a value
 * out of range, or a lab option in a library built without -DHVX_LAB: -1 + hvx_last_error().  hvx_option_name enumerates them (NULL past the end).
 * hvx_build_flags: the extra compiler flags this library was built with ("" for the shipped one); hvx_is_lab_build: 1 when built with -DHVX_LAB (timing-only
 * kernels that do not store results may be reachable: flowmirror_hydravox_amd._lib refuses such a library unless it was named explicitly). */
int hvx_set_option(const char* key, int64_t value);
int hvx_get_option(const char* key, int64_t* value);
const char* hvx_option_name(int32_t index, int32_t* lab_only, int64_t* dflt);
const char* hvx_build_flags(void);
int hvx_is_lab_build(void);

int hvx_prof_enable(int32_t period);
int hvx_prof_read(int32_t kind, double* sampled_ms, double* sampled_work, int64_t* n_sampled, int64_t* n_launched, double* launched_work);

/* ---------------------------------------------------------------------------------------------------
 * Sampler — replaces cosyvoice/utils/common.py:138-166 (ras_sampling, nucleus_sampling, random_sampling)
 * and the EOS-rejection loop of cosyvoice/llm/llm_multi_head_v3.py:151-166, for all K heads of a step
 * against one history snapshot (:890-900).  `noise` is the per-sequence Exp(1) stream that
 * torch.multinomial(1) would have drawn from the CPU generator; `cursor` (in/out) the next unread value.
 * out_ids: >= 0 sampled id; -1 max_trials exhausted (the reference raises RuntimeError); -2 noise exhausted
 * (cursor left untouched; refill and call again).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_seq, head_k, vocab, speech_tokens;
    const float* logp; int64_t logp_seq_stride, logp_head_stride;
    const int32_t* hist; int64_t hist_seq_stride; const int32_t* hist_len;
    const int32_t* min_len; const int32_t* active;
    int32_t top_k; float top_p; int32_t win_size; int32_t rep_thresh;
    const float* noise; int64_t noise_seq_stride; int32_t noise_len;
    int64_t* cursor;
    int32_t* out_ids;
    int32_t max_trials;
    const int64_t* noise_limit;   /* optional [n_seq]: `noise` is then a ring addressed by absolute stream position % noise_len, valid below the limit */
} hvx_sample_args;
int hvx_ras_sample(const hvx_sample_args* a, hvx_stream s);

/* ---------------------------------------------------------------------------------------------------
 * Building-block operators (exported for the parity tests; the model entry points below are chains of these)
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t dtype, M, N, K, batch, groups;
    const void* A; int64_t a_bs; int32_t lda, a_gs, rows_in;
    int32_t cin_pad, conv_stride, conv_dil, pad_left, up;
    const void* W; int64_t w_gs;
    const float* bias;
    int32_t act; float act_param; const float* act_alpha;
    const float* gate; int64_t gate_bs;
    const float* res; int64_t res_bs; int32_t ldres, res_row_off;
    float scale;
    void* out; int32_t out_f32; int64_t out_bs; int32_t ldo, out_row_off, out_cols;
    void* out2; int32_t act2; float act2_param; const float* act2_alpha; int64_t out2_bs; int32_t ldo2, out2_row_off, out2_cols;
    int32_t x3;   /* fp32 only: allow the split-bf16 form (each operand as a bf16 pair, three bf16 MFMAs per step; ~1e-6 relative) */
    int32_t res_f16;   /* res holds IEEE fp16 instead of fp32 (same strides, in elements): the DiT's half residual stream, hvx_flow_set_half_stream */
    int32_t out_f16;   /* out holds IEEE fp16 (out_f32 must be 0) */
} hvx_gemm_args;
/* implicit-GEMM Conv1d / Linear on MFMA (torch.nn.functional.conv1d / linear call sites of the path) */
int hvx_op_gemm(const hvx_gemm_args* a, hvx_stream s);

typedef struct {
    int32_t dtype, batch, heads, t, t_pad;     /* q,k: [b][h][t_pad][64]; vT: [b][h][64][t_pad]; out: [b][t][h*64] */
    const void* q; const void* k; const void* vT; void* out;
    const int32_t* kv_len;                     /* optional [batch] */
    int32_t causal; float scale;
    int32_t n_splits, split_chunk; float* part_o; float* part_ml;   /* optional key splits (workspace) */
    int32_t chunk;                             /* > 0: static chunk mask, row i sees keys j < (i / chunk + 1) * chunk (cosyvoice/utils/mask.py:128-158) */
    int32_t q_log2;                            /* != 0: q already holds q * scale * log2(e) (the fused QKV epilogue's form); `scale` is ignored */
} hvx_attn_args;
/* F.scaled_dot_product_attention (cosyvoice/flow/DiT/modules.py:391) */
int hvx_op_attention(const hvx_attn_args* a, hvx_stream s);

/* F.interpolate(x, size=t_out, mode='linear') along the last axis of an f32 (rows, t_in) array: the `speed` knob of
 * inference_zero_shot / text_to_speech (infer_speech_model.py:583-588) and token2wav (cosyvoice/cli/model.py:424-426) */
int hvx_op_resample_linear(const float* x, int32_t rows, int32_t t_in, float* y, int32_t t_out, hvx_stream s);

/* packs a row-major [N][K] weight into the MFMA fragment order used by the decode GEMMs: [N/16][K/32][64][8] */
int hvx_op_skinny_gemm(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* Wpacked,
                       const float* bias, int32_t split_k, float* part_ws, float* out_f32, int32_t ldo, hvx_stream s);

/* ---------------------------------------------------------------------------------------------------
 * LLM — replaces CosyVoice3LM.inference_wrapper's per-step math (cosyvoice/llm/llm_multi_head_v3.py
 * :871-888): Qwen2 backbone over the NEW rows only (KV-cached; identical to the reference's full-prefix
 * recompute because the mask is causal), K MTP heads, llm_decoder, log_softmax.
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t dtype;                               /* HVX_BF16 (production) or HVX_F32 (parity mode) */
    int32_t hidden, layers, q_heads, kv_heads, inter;            /* head_dim == 64 */
    int32_t vocab, vocab_pad, speech_tokens, text_vocab;
    int32_t head_num, mtp_attn_dim, mtp_inter;
    float rms_eps, mtp_rms_eps;
    int32_t max_pos;
} hvx_llm_config;

/* weights[] order (device pointers; matrices in `dtype`, packed with hvx fragment order unless noted; vectors f32):
 *   0 rope_cos f32 [max_pos][32]    1 rope_sin f32 [max_pos][32]     2 final norm gain [H]
 *   3 llm_decoder packed [vocab_pad][H]   4 speech_embedding rows [vocab][H] (dtype)   5 text embedding rows [text_vocab][H] (dtype)
 *   per layer l, 9 entries from 6+9l: ln1 gain, Wqkv packed [(q+2kv)*64][H], bqkv, Wo [H][q*64] in the narrow order, ln2 gain,
 *                                     Wgate/up packed as alternating 16-row tiles [2*inter][H], Wdown [H][inter] in the narrow order
 *                                     ([N/4][K/128][64 lanes][8], packing.pack_narrow4; needs q*64 % 128 == 0 and inter % 128 == 0),
 *                                     then Wo and Wdown once more in the 16-column fragment order (used for grids of > 32 rows);
 *                                     the backbone's Wqkv / Wgate/up carry their RMSNorm gain folded in (W[n][k] * ln[k]): the
 *                                     kernels scale by 1/rms only, the ln1 / ln2 entries are kept for layout stability and not read
 *   then 7 entries stacked over the head_num MTP heads: ln1 [hn][H], Wv packed [hn][A][H], bv [hn][A], Wo packed [hn][H][A],
 *                                     ln2 [hn][H], Wgate/up packed [hn][2*mtp_inter][H], Wdown packed [hn][H][mtp_inter]        */
typedef struct hvx_llm hvx_llm;
int hvx_llm_create(const hvx_llm_config* cfg, const void* const* weights, int32_t n_weights, hvx_llm** out);
void hvx_llm_destroy(hvx_llm* h);
size_t hvx_llm_workspace_bytes(const hvx_llm* h, int32_t max_seq, int32_t max_rows, int32_t max_ctx);
size_t hvx_llm_kv_bytes(const hvx_llm* h, int32_t n_slots, int32_t max_ctx);
int hvx_llm_bind(hvx_llm* h, void* workspace, size_t ws_bytes, int32_t max_seq, int32_t max_rows, void* kv, size_t kv_bytes,
                 int32_t n_slots, int32_t max_ctx, hvx_stream s);
/* One pass over a dense [n_seq][kn] grid of new rows.  Device int32 control arrays:
 *   tok[n_seq*kn]  >= 0: speech_embedding row; <= -2: text-embedding row (-tok-2); -1: padding row
 *   ctrl[5][n_seq] = slot, pos0 (KV length before this call), n_new (valid rows), kv_len (= pos0+n_new), last_row (grid row of the
 *                    last valid row, -1 if none)
 * head_k > 0: writes logp[n_seq][head_k][vocab] (fp32 log-probabilities of the K heads on each sequence's last new row). */
int hvx_llm_forward(hvx_llm* h, hvx_stream s, int32_t n_seq, int32_t kn, const int32_t* tok, const int32_t* ctrl,
                    int32_t head_k, float* logp);
/* replay the decode-step launches (head_k > 0) as one cached hipGraph per (grid, control/logp addresses, stream) */
int hvx_llm_use_graph(hvx_llm* h, int32_t enable);
/* fp8 copy of the MTP heads' gate / up projection (fp8 head weights; the largest weight stream of a decode step, 79 MB per head in
 * bf16).  codes: [head_num][2 * mtp_inter * hidden] e4m3 (OCP) codes in the double-step fragment order of packing.pack_frag_fp8 over the same
 * interleaved (gate, up) tile order as weights[mtp gate/up]; scales: [head_num][2 * mtp_inter] fp32 powers of two, one per output column.  The bf16
 * tensor passed to hvx_llm_create must hold exactly code * scale (packing.quantize_e4m3_pow2 makes both): wide bf16 decode grids (33..256
 * sequences) then stream the codes — half the bytes, the same products in the same order, bit-identical log-probabilities — and every other
 * grid keeps reading the bf16 tensor.  NULL codes switch it off.  Device pointers; the caller keeps them alive.  Cached decode graphs are dropped. */
int hvx_llm_set_head_mlp_fp8(hvx_llm* h, const void* codes, const float* scales);
/* Device-resident decode loop — replaces the per-step host logic of CosyVoice3LM.inference_wrapper (llm_multi_head_v3.py:871-905):
 * one call enqueues `n_steps` repetitions of { forward over the [n_seq][head_k] grid, RAS sampling of the K heads, advance }, where
 * `advance` does on the device what the reference's Python loop does between two steps: feed the accepted tokens back (tok / ctrl
 * of the next step), append them to the utterance, slide the repetition window, stop a sequence on a stop id / max_len / an empty
 * group.  Nothing is read back between steps; the caller looks at `seq_state` whenever it wants (e.g. every 8 steps).
 *   tok / ctrl                  as in hvx_llm_forward (kn = head_k); rewritten by every step
 *   hist [n_seq][win_cap]       ring of the last tokens (slot = out_len % win_cap), hist_len = min(out_len, win_cap)
 *   min_adj [n_seq]             min_len - (out_len - hist_len): head j ignores stop ids while hist_len + j < min_adj
 *   active [n_seq]              0 once a sequence is finished
 *   seq_state [n_seq][8]        pos, out_len, done, min_len, max_len, steps, err (1 = sampler max_trials exhausted, 2 = the sequence is
 *                               waiting for noise: its steps are void, nothing advances, until the ring is topped up and the flag cleared), reserved
 *   out_tokens [n_seq][max_out] the utterance so far (out_len valid entries)
 *   ids [n_seq][head_k]         scratch: the ids sampled by the last step
 * Sampler fields as in hvx_sample_args.  The step is replayed from a cached hipGraph keyed on the argument block. */
typedef struct {
    int32_t n_seq, head_k, win_cap, max_out;
    int32_t* tok; int32_t* ctrl; int32_t* hist; int32_t* hist_len; int32_t* min_adj; int32_t* active;
    int32_t* seq_state; int32_t* out_tokens; int32_t* ids;
    float* logp;
    int32_t top_k; float top_p; int32_t win_size; int32_t rep_thresh; int32_t max_trials;
    const float* noise; int64_t noise_seq_stride; int32_t noise_len; int64_t* cursor;
    const int64_t* noise_limit;   /* [n_seq] absolute positions up to which the noise ring is filled (the caller tops it up while steps run) */
} hvx_decode_args;
int hvx_llm_decode_steps(hvx_llm* h, hvx_stream s, const hvx_decode_args* a, int32_t n_steps);
/* Continuous batching — the batching scheduler SURVEY.md §8(f) N1 asks for in place of the reference's one-request-at-a-time worker loop
 * (server/worker.py:54-102): between two hvx_llm_decode_steps calls on the same stream a new sequence takes over slot `slot` of the grid
 * described by `a` (a finished sequence's slot, or one that was never used).  The caller has prefilled all but the last prefix row into
 * that slot's KV cache (hvx_llm_forward with ctrl = {slot, 0, n, n, n - 1}, head_k = 0) and has put the sequence's noise at position 0 of
 * the slot's ring; this call writes the slot's decode state: first_tok = the last prefix row, pos = rows already cached, the length
 * limits, an empty repetition window, cursor 0, active. */
int hvx_llm_decode_join(hvx_llm* h, hvx_stream s, const hvx_decode_args* a, int32_t slot, int32_t first_tok, int32_t pos,
                        int32_t min_len, int32_t max_len);
/* debugging / parity: copy the post-final-norm hidden of the last rows of the previous forward (fp32 [n_seq][H]) */
int hvx_llm_last_hidden(hvx_llm* h, hvx_stream s, int32_t n_seq, float* out);

/* ---------------------------------------------------------------------------------------------------
 * Flow — replaces CausalMaskedDiffWithDiT.inference (cosyvoice/flow/flow.py:367-430), PreLookaheadLayer
 * (cosyvoice/transformer/upsample_encoder.py:82-103), CausalConditionalCFM.forward / solve_euler
 * (cosyvoice/flow/flow_matching.py:204-228, 71-124) and DiT.forward (cosyvoice/flow/DiT/dit.py:145-176).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t dtype;
    int32_t vocab, mel, spk_dim, pla_channels, pla_len;
    int32_t dim, depth, heads, ff, conv_kernel, conv_groups, time_freq_dim;
    int32_t max_t;                                /* rows of the rope tables */
    float cfg_rate;
} hvx_flow_config;
/* weights[] order (matrices `dtype` row-major [N][K] with conv kernels as [Cout][tap][Cin_pad]; vectors f32):
 *   0 rope_cos f32 [max_t][32]  1 rope_sin   2 input_embedding f32 [vocab][mel]   3 spk affine W f32 [mel][spk_dim]  4 spk affine b
 *   5 pla conv1 W [C][4][96]  6 b   7 pla conv2 W [mel][3][C]  8 b
 *   9 time_mlp.0 W [D][256] 10 b  11 time_mlp.2 W [D][D] 12 b   13 input proj W [D][320] 14 b
 *   15 conv_pos 1 W [groups][Cg][k*Cg] 16 b [D]  17 conv_pos 2 W  18 b
 *   per block, 10 entries from 19+10i: adaLN W [6D][D], b; Wqkv [3D][D], bqkv; Wout [D][D], b; ff1 W [ff][D], b; ff2 W [D][ff], b
 *   tail: norm_out W [2D][D], b; proj_out W [mel][D], b                                                             */
typedef struct hvx_flow hvx_flow;
int hvx_flow_create(const hvx_flow_config* cfg, const void* const* weights, int32_t n_weights, hvx_flow** out);
void hvx_flow_destroy(hvx_flow* h);
size_t hvx_flow_workspace_bytes(const hvx_flow* h, int32_t batch, int32_t t);
/* token (int32 [n], prompt already concatenated) + speaker embedding (f32 [spk_dim]) -> mu f32 (mel, 2n) channel-major, spk f32 [mel] */
int hvx_flow_encode(hvx_flow* h, hvx_stream s, void* ws, size_t ws_bytes, const int32_t* token, int32_t n, const float* embedding,
                    float* mu, float* spk);
/* pre-lookahead only (parity): x f32 [n][mel] -> y f32 [n][mel] */
int hvx_flow_prelookahead(hvx_flow* h, hvx_stream s, void* ws, size_t ws_bytes, const float* x, int32_t n, float* y);
/* Chunked synthesis (finalize=False, flow.py:401-404; upsample_encoder.py:90-95): the last pla_len rows / tokens are the look-ahead
 * context of the rows before them instead of zero padding, and produce no output: y f32 [n - pla_len][mel], mu (mel, 2 (n - pla_len)).
 * finalize != 0 is hvx_flow_encode. */
int hvx_flow_prelookahead_context(hvx_flow* h, hvx_stream s, void* ws, size_t ws_bytes, const float* x, int32_t n, float* y);
int hvx_flow_encode_chunk(hvx_flow* h, hvx_stream s, void* ws, size_t ws_bytes, const int32_t* token, int32_t n, const float* embedding,
                          int32_t finalize, float* mu, float* spk);
/* estimator, TensorRT argument order (flow_matching.py:130-153): x, mu, cond f32 (B, mel, T); mask -> kv_len int32 [B] (NULL: all T);
 * t f32 [B]; spks f32 (B, mel); out f32 (B, mel, T) */
int hvx_cfm_estimator(hvx_flow* h, hvx_stream s, void* ws, size_t ws_bytes, int32_t batch, int32_t t_len, const float* x,
                      const int32_t* kv_len, const float* mu, const float* t, const float* spks, const float* cond, float* out);
/* streaming=True (dit.py:163-164): attention row i additionally sees only keys j < (i / static_chunk_size + 1) * static_chunk_size
 * (cosyvoice/utils/mask.py:128-158, 223-230); static_chunk_size == 0 is hvx_cfm_estimator. */
int hvx_cfm_estimator_streaming(hvx_flow* h, hvx_stream s, void* ws, size_t ws_bytes, int32_t batch, int32_t t_len, const float* x,
                                const int32_t* kv_len, const float* mu, const float* t, const float* spks, const float* cond,
                                int32_t static_chunk_size, float* out);
/* bf16 mode only: keep the residual stream of the DiT blocks in IEEE fp16 instead of fp32 (on != 0).  The reference runs its flow decoder in
 * fp16 end to end (infer_speech_model.py:103 `flow.eval().cuda().half()`), so an fp16 stream is the reference's own arithmetic; every update
 * gate * (Linear + bias) + x is still formed in fp32 and rounded once (saturating at +-65504).  Halves the HBM traffic of the two residual Linears and of the adaLN
 * passes of every block.  Off by default at the C level; the Python host turns it on for bf16 handles (HvxFlow(half_stream=True)). */
int hvx_flow_set_half_stream(hvx_flow* h, int32_t on);
/* bf16 mode only: the QKV, FF1 and FF2 Linears of every DiT block take IEEE fp16 operands — the reference's deployed dtype
 * (infer_speech_model.py:103) — and the activations that feed them (adaLN outputs, FF hidden) are stored as fp16.  The caller must have passed THOSE
 * weight matrices (weights[19 + 10 i + {2, 6, 8}]) as fp16 instead of bf16; q / k / v, the attention and the Linear behind it stay bf16 (its input is
 * the attention's bf16 output, exact in fp16: fp16 weights there measured no different).
 * Brings the bf16 mode's distance from the fp32 reference (4.7e-3 of the output scale through 22 blocks) to the reference's own fp16 distance (1.4-1.7e-3). */
int hvx_flow_set_f16_linears(hvx_flow* h, int32_t on);
/* bf16 mode only: the SMALL Linears of the estimator — time MLP (weights[9], [11]), every adaLN modulation Linear (weights[19 + 10 i + 0], and
 * weights[19 + 10 depth + 0]), the input projection (weights[13]) and the output projection (weights[19 + 10 depth + 2]) — run in fp32; the caller must have
 * passed THOSE matrices as fp32.  Together with hvx_flow_set_f16_linears and the fp16 stream this is the mode that sits at the reference's own fp16 distance
 * from fp32.  Needs ff >= 2 dim (the fp32 adaLN output of the last layer borrows the FF hidden buffer). */
int hvx_flow_set_f32_small(hvx_flow* h, int32_t on);
/* optional persistent device buffer in which hvx_cfm_solve keeps the adaLN modulation vectors of each distinct step time t
 * (they depend on t and the weights only); pass NULL to disable.  Must be re-set after the weights change. */
int hvx_flow_set_mod_cache(hvx_flow* h, void* buf, size_t bytes);
/* full Euler solve with batch-2 classifier-free guidance: x (mel,T) f32 in (noise) / out (mel); t_steps[n], dt_steps[n] from the host
 * (computed exactly as the reference accumulates them) */
int hvx_cfm_solve(hvx_flow* h, hvx_stream s, void* ws, size_t ws_bytes, int32_t t_len, float* x, const float* mu, const float* spks,
                  const float* cond, int32_t n_steps, const float* t_steps, const float* dt_steps);
/* the same with the static chunk mask of streaming=True (flow_matching.py:204-228 -> dit.py:163-164) */
int hvx_cfm_solve_streaming(hvx_flow* h, hvx_stream s, void* ws, size_t ws_bytes, int32_t t_len, float* x, const float* mu, const float* spks,
                            const float* cond, int32_t n_steps, const float* t_steps, const float* dt_steps, int32_t static_chunk_size);

/* Padded multi-utterance solve (length-bucketed acoustic batches; the reference solves one utterance per call,
 * cosyvoice/flow/flow.py:387, with the batch-2 CFG layout of flow_matching.py:95-108): n utterances padded to T frames, t_len device
 * int32 [n] valid frames of each (NULL: all T).  x (n, mel, T) holds the noise on entry and the mels on return; mu, cond (n, mel, T);
 * spks (n, mel).  The estimator runs on 2n batch entries (n conditional, n unconditional) with key-padding masks from t_len; frames at or
 * beyond an utterance's t_len hold unspecified values on return.  Workspace: hvx_flow_workspace_bytes(h, 2n, T). */
int hvx_cfm_solve_batch(hvx_flow* h, hvx_stream s, void* ws, size_t ws_bytes, int32_t n, int32_t t_max, const int32_t* t_len, float* x,
                        const float* mu, const float* spks, const float* cond, int32_t n_steps, const float* t_steps, const float* dt_steps,
                        int32_t static_chunk_size);

/* ---------------------------------------------------------------------------------------------------
 * HiFT — replaces CausalHiFTGenerator.inference / decode (cosyvoice/hifigan/generator.py:713-726, 672-711),
 * CausalConvRNNF0Predictor.forward (cosyvoice/hifigan/f0_predictor.py:95-103), SourceModuleHnNSF / SineGen2
 * (generator.py:358-375, 233-317).  fp32 throughout, like the reference (infer_speech_model.py:104).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t mel, base_channels, nb_harmonics, f0_channels;
    int32_t n_up; int32_t up_rates[4]; int32_t up_kernels[4];
    int32_t n_rb; int32_t rb_kernels[4]; int32_t rb_dils[4][3];
    int32_t src_rb_kernels[4]; int32_t src_rb_dils[4][3];
    int32_t n_fft, hop, conv_pre_kernel, conv_post_kernel;
    float sampling_rate, nsf_alpha, nsf_sigma, voiced_threshold, lrelu_slope, audio_limit;
    int32_t exact_fp32;     /* 0: decode convolutions as 3 bf16 MFMAs on (hi, lo) operand pairs (~16 mantissa bits: 1.5e-4 of the reference at 5632 frames);
                             * 1: on the exact fp32 MFMA forms (2.8e-5), the arithmetic of the reference's fp32 vocoder.  The F0 predictor is exact fp32 either way. */
} hvx_hift_config;
/* weights[]: f32, weight-norm folded, conv kernels as [Cout][tap][Cin_pad32]; consumed in the order documented in
 * flowmirror_hydravox_amd/hift.py::pack_hift_weights (f0 predictor, source linear, conv_pre, per stage: up, source_down,
 * source resblock, 3 resblocks; conv_post). */
typedef struct hvx_hift hvx_hift;
int hvx_hift_create(const hvx_hift_config* cfg, const void* const* weights, int32_t n_weights, hvx_hift** out);
void hvx_hift_destroy(hvx_hift* h);
/* Optional: per weight pointer of hvx_hift_create (same order, n = n_weights) either NULL or the convolution weight as a (hi, lo) bf16 plane
 * pair [2][Cout][taps * Cin_pad] (hi = bf16(w), lo = bf16(w - hi)).  With planes for every ResBlock / up-sampling / conv_post weight the decode
 * keeps its activations as plane pairs too and the split-bf16 convolutions read both operands by LDS-DMA (csrc/gemm_x3.hip: gemm_x3p_kernel);
 * values are the same as without.  The caller keeps the planes alive as long as the handle. */
int hvx_hift_set_weight_planes(hvx_hift* h, const void* const* planes, int32_t n);
size_t hvx_hift_workspace_bytes(const hvx_hift* h, int32_t t);
/* mel f32 (mel, T) channel-major -> f0 f32 [T] */
int hvx_hift_f0(hvx_hift* h, hvx_stream s, void* ws, size_t ws_bytes, const float* mel, int32_t t, float* f0);
/* f0 f32 [T] + fixed noise table f32 [>= T*up][H] -> source f32 [T*up] */
int hvx_hift_source(hvx_hift* h, hvx_stream s, void* ws, size_t ws_bytes, const float* f0, int32_t t, const float* sine_table,
                    float* source);
/* mel (mel, T) + source [T*up] -> wav f32 [T*up] */
int hvx_hift_decode(hvx_hift* h, hvx_stream s, void* ws, size_t ws_bytes, const float* mel, const float* source, int32_t t, float* wav);
/* finalize=False (generator.py:672-711): the last look_right of the t_in mel frames are real right context of conv_pre and the source STFT
 * is cut to match: mel (mel, t_in) + source [t_in*up] -> wav f32 [(t_in - look_right)*up] (the caller drops its last up*... hop samples,
 * generator.py:708-709).  Workspace: hvx_hift_workspace_bytes(h, t_in). */
int hvx_hift_decode_chunk(hvx_hift* h, hvx_stream s, void* ws, size_t ws_bytes, const float* mel, const float* source, int32_t t_in,
                          int32_t look_right, float* wav);

/* ---------------------------------------------------------------------------------------------------
 * Matcha-TTS family, fp32:
 *   hvx_matcha_*   — matcha/models/components/decoder.py:363-443 Decoder.forward (and its CosyVoice variant
 *                    cosyvoice/flow/decoder.py:210-291 when cv_variant = 1: `cond` input, key-padding masks) and
 *                    flow_matching.py:32-85 BASECFM.solve_euler
 *   hvx_hifigan_*  — matcha/hifigan/models.py:181-197 Generator.forward (config.py v1)
 *   hvx_denoise    — matcha/hifigan/denoiser.py:57-64 Denoiser.forward
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t in_channels, out_channels;          /* packed [x | mu | spks | cond] width (multiple of 32); mel bins */
    int32_t n_stages; int32_t channels[4];      /* U-Net widths (multiples of 64) */
    int32_t n_blocks, n_mid, heads, ff_mult;    /* transformer blocks per stage, mid stages, heads of 64, FeedForward multiplier */
    int32_t cv_variant;                         /* 0: Matcha (full-length masks only), 1: CosyVoice ConditionalDecoder (key-padding masks) */
    int32_t max_t;
} hvx_matcha_config;
/* weights[] (all f32; Linear [N][K], Conv1d [Cout][tap][Cin_pad32]):
 *   time_mlp: W1 [TE][Cin], b1, W2 [TE][TE], b2                                  (TE = 4 * channels[0])
 *   units in execution order — n_stages down, n_mid mid, n_stages up — each:
 *     resnet: mlp W [C][TE], b; block1 conv W, b, GroupNorm gamma, beta; block2 conv W, b, gamma, beta; res_conv W [C][Cin_pad], b
 *     n_blocks x transformer: norm1 (gamma - 1), beta; Wqkv [3*heads*64][C]; Wo [C][heads*64], bo; norm3 (gamma - 1), beta;
 *                             ff W1 [ff_mult*C][C], b1, snake table [2][ff_mult*C] = exp(alpha) | exp(beta); ff W2 [C][ff_mult*C], b2
 *     down / up units only: resampler W, b — Conv1d k3 (stride 2, or 1 on the last stage) / ConvTranspose1d k4 s2 p1 as its two
 *                           phase matrices [2][C][2 taps][C] (packing.convtranspose_phases), Conv1d k3 on the last up stage
 *   final_block conv W, b, GroupNorm gamma, beta; final_proj W [mel][C], b */
typedef struct hvx_matcha hvx_matcha;
int hvx_matcha_create(const hvx_matcha_config* cfg, const void* const* weights, int32_t n_weights, hvx_matcha** out);
void hvx_matcha_destroy(hvx_matcha* h);
size_t hvx_matcha_workspace_bytes(const hvx_matcha* h, int32_t batch, int32_t t);
/* x, mu, cond: f32 (B, mel, T); spks f32 (B, spk_dim) or null; lens: device int32 [n_stages][B] valid lengths per U-Net level or null;
 * t: device f32 [B]; out: f32 (B, mel, T) */
int hvx_matcha_estimator(hvx_matcha* h, hvx_stream s, void* ws, size_t ws_bytes, int32_t batch, int32_t t_len, const float* x, const float* mu,
                         const float* spks, int32_t spk_dim, const float* cond, const int32_t* lens, const float* t, float* out);
/* x holds z on entry and the sample on return; ts / dts: host arrays, the fp32 values solve_euler's loop visits */
int hvx_matcha_solve(hvx_matcha* h, hvx_stream s, void* ws, size_t ws_bytes, int32_t batch, int32_t t_len, float* x, const float* mu,
                     const float* spks, int32_t spk_dim, const float* cond, const int32_t* lens, int32_t n_steps, const float* ts, const float* dts);

typedef struct {
    int32_t mel, initial_channel;
    int32_t n_up; int32_t up_rates[4]; int32_t up_kernels[4];
    int32_t n_rb; int32_t rb_kernels[4]; int32_t rb_dils[4][3];
    int32_t exact_fp32;     /* as hvx_hift_config.exact_fp32 */
} hvx_hifigan_config;
/* weights[] (f32, weight-norm folded): conv_pre W [C0][7][mel_pad32], b; per upsample: ConvTranspose phases [rate][C/2][k/rate][C], b,
 * then n_rb x 3 x (convs1 W, b, convs2 W, b); conv_post W [1][7][C], b */
typedef struct hvx_hifigan hvx_hifigan;
int hvx_hifigan_create(const hvx_hifigan_config* cfg, const void* const* weights, int32_t n_weights, hvx_hifigan** out);
void hvx_hifigan_destroy(hvx_hifigan* h);
size_t hvx_hifigan_workspace_bytes(const hvx_hifigan* h, int32_t t);
int hvx_hifigan_forward(hvx_hifigan* h, hvx_stream s, void* ws, size_t ws_bytes, const float* mel, int32_t T, float* wav);

/* mel_spectrogram (matcha/utils/audio.py:45-82), the prompt-feature extractor the zero-shot frontend calls (cosyvoice/cli/frontend.py:119;
 * SURVEY.md §8(f) N2): audio f32 [L] -> out f32 (n_mels, frames), frames = (L + 2*((n_fft - hop)/2) - n_fft) / hop + 1.
 * stft_basis [2*(n_fft/2+1)][n_fft] (hann window folded in), mel_basis [n_mels][pad32(n_fft/2+1)] */
size_t hvx_mel_workspace_bytes(int32_t L, int32_t n_fft, int32_t hop, int32_t n_mels);
int hvx_mel_spectrogram(hvx_stream s, void* ws, size_t ws_bytes, const float* audio, int32_t L, int32_t n_fft, int32_t hop,
                        const float* stft_basis, const float* mel_basis, int32_t n_mels, float* out);
/* Framed spectral features of the speech frontends — one description for matcha's mel_spectrogram, whisper's
 * log_mel_spectrogram (speech-tokenizer input, cosyvoice/cli/frontend.py:95) and kaldi's fbank (CAM++ input, frontend.py:104-108):
 * [reflect-pad ->] frames of `frame_len` samples every `hop` -> GEMM with `basis` [2*bins][frame_len] (window and whatever else is
 * linear in the frame folded in by the caller: DC removal, pre-emphasis, zero padding of the FFT) -> |X| or |X|^2 -> GEMM with
 * `mel_basis` [n_mels][pad32(bins)] -> ln(max(., log_floor)) * log_scale -> post-processing.  frame_len and hop are multiples of 32
 * (a 400-sample frame is given as 416 with 16 zero basis columns). */
typedef struct {
    int32_t frame_len, hop;
    int32_t reflect_pad;        /* samples mirrored onto both ends before framing (torch.stft center=True: n_fft/2); 0: frames start at sample 0 */
    int32_t n_frames;
    int32_t bins;
    int32_t power; float mag_eps;               /* power != 0: |X|^2; else sqrt(|X|^2 + mag_eps) */
    int32_t n_mels; float log_floor, log_scale;
    int32_t post;               /* 0 none; 1 whisper: (max(x, max(x) - 8) + 4) / 4; 2 subtract every mel bin's mean over the frames */
    int32_t time_major;         /* output (n_frames, n_mels) instead of (n_mels, n_frames) */
} hvx_feature_config;
size_t hvx_frame_features_workspace_bytes(int32_t L, const hvx_feature_config* c);
int hvx_frame_features(hvx_stream s, void* ws, size_t ws_bytes, const float* audio, int32_t L, const hvx_feature_config* c, const float* basis,
                       const float* mel_basis, float* out);
size_t hvx_denoise_workspace_bytes(int32_t L, int32_t n_fft, int32_t hop);
/* |torch.stft(audio, n_fft, hop, window=hann, center=True)|: mag f32 [1 + L/hop][n_fft/2 + 1] */
int hvx_stft_magnitude(hvx_stream s, void* ws, size_t ws_bytes, const float* audio, int32_t L, int32_t n_fft, int32_t hop, const float* stft_basis,
                       float* mag);
int hvx_denoise(hvx_stream s, void* ws, size_t ws_bytes, const float* audio, int32_t L, int32_t n_fft, int32_t hop, const float* stft_basis,
                const float* istft_basis, const float* wsq, const float* bias, float strength, float* out);

/* ---- ONNX graph operators ---------------------------------------------------------------------------------------------
 * The zero-shot frontend of the reference runs `speech_tokenizer_v3.onnx` and `campplus.onnx` through onnxruntime on the prompt audio
 * (server/model_utils/cosyvoice/cli/frontend.py:92-115).  flowmirror_hydravox_amd/onnx_graph.py executes such graphs on the device: Conv /
 * MatMul / Gemm through hvx_op_gemm (exact fp32 MFMA forms), everything else through the fp32 kernels below.  Tensors are dense fp32;
 * integer (shape) tensors stay on the host. */
enum {  /* hvx_nd_elementwise operators: unary on a | binary on (a, b) | select */
    HVX_EW_COPY = 0, HVX_EW_RELU, HVX_EW_SIGMOID, HVX_EW_TANH, HVX_EW_ERF, HVX_EW_SQRT, HVX_EW_EXP, HVX_EW_LOG, HVX_EW_NEG, HVX_EW_ABS, HVX_EW_ROUND,
    HVX_EW_FLOOR, HVX_EW_CEIL, HVX_EW_RECIP, HVX_EW_CLIP /* [p0, p1] */, HVX_EW_LEAKY_RELU /* slope p0 */, HVX_EW_SOFTPLUS, HVX_EW_SIN, HVX_EW_COS,
    HVX_EW_ADD = 32, HVX_EW_SUB, HVX_EW_MUL, HVX_EW_DIV, HVX_EW_POW, HVX_EW_MAX, HVX_EW_MIN, HVX_EW_EQUAL, HVX_EW_LESS, HVX_EW_GREATER,
    HVX_EW_WHERE = 48   /* out = a != 0 ? b : c */
};
enum { HVX_RED_SUM = 0, HVX_RED_MEAN = 1, HVX_RED_MAX = 2, HVX_RED_MIN = 3, HVX_RED_SUMSQ = 4 };
typedef struct {
    int32_t ndim;                         /* 1..6 */
    int32_t shape[6];                     /* extents of the (contiguous, row-major) output */
    int64_t stride_a[6], stride_b[6], stride_c[6];   /* element strides of the operands along the output axes; 0 broadcasts */
} hvx_nd;
/* out[i] = op(a[..], b[..], c[..]) over the output index space of d: the arithmetic of an ONNX graph (Add, Mul, Relu, Sigmoid, Erf, Clip, Where ...)
 * and, with HVX_EW_COPY, its data movement (Transpose, Slice, Expand and Concat pieces are strided copies) */
int hvx_nd_elementwise(int32_t op, const hvx_nd* d, const float* a, const float* b, const float* c, float p0, float p1, float* out, hvx_stream s);
/* out[r] = reduce(x[r][0..cols)) (ReduceSum / ReduceMean / ReduceMax / ReduceMin / sum of squares over the last axis of a [rows][cols] view) */
int hvx_rows_reduce(int32_t op, const float* x, int64_t rows, int64_t cols, float* out, hvx_stream s);
/* out[r][:] = softmax(x[r][:]) (ONNX Softmax, axis = last) */
int hvx_rows_softmax(const float* x, int64_t rows, int64_t cols, float* out, hvx_stream s);
/* ONNX AveragePool over the last axis of [rows][t_in] -> [rows][t_out] (the caller computes t_out, with ceil_mode if the node asks for it) */
int hvx_avgpool_rows(const float* x, int64_t rows, int32_t t_in, int32_t kernel, int32_t stride, int32_t pad, int32_t count_include_pad, float* y,
                     int32_t t_out, hvx_stream s);
/* ONNX Conv, 2-D, group 1, dilation 1: x [B][Cin][H][W], w [Cout][Cin][kh][kw], bias [Cout] or NULL -> y [B][Cout][Ho][Wo] (the CAM++ front module) */
int hvx_conv2d(const float* x, const float* w, const float* bias, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t kh, int32_t kw,
               int32_t sh, int32_t sw, int32_t ph, int32_t pw, float* y, hvx_stream s);

#ifdef __cplusplus
}
#endif
#endif /* HVX_H */
