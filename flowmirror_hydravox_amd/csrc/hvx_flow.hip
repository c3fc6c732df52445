// hvx_flow.hip — flow-matching mel decoder: token encoder, DiT estimator, CFG Euler solver (include/hvx.h: hvx_flow_*, hvx_cfm_*).
//
// Restates (file:line under server/model_utils/cosyvoice/):
//   flow/flow.py:389-405            speaker affine, token embedding, pre-lookahead, x2 repeat
//   transformer/upsample_encoder.py:82-103   PreLookaheadLayer
//   flow/DiT/dit.py:145-176         DiT.forward             flow/DiT/modules.py:516-530  DiTBlock
//   flow/DiT/modules.py:230-244     AdaLayerNormZero        :349-407 AttnProcessor   :271-282 FeedForward
//   flow/DiT/modules.py:115-144     CausalConvPositionEmbedding      :606-616 TimestepEmbedding
//   flow/flow_matching.py:71-124    solve_euler (batch-2 CFG)
// Data layout in HBM: activations are time-major rows [B][T][C] (C contiguous) so that every Linear and every
// Conv1d is one implicit-GEMM launch; GEMM operands are `dtype`; the DiT's residual stream is fp32 in fp32 mode and IEEE fp16 in bf16 mode
// (hvx_flow_set_half_stream; the reference runs the whole decoder in fp16, infer_speech_model.py:103 — the updates are still formed in fp32
// and rounded once per residual add); q/k are written
// head-major and V already transposed by the QKV epilogue, which is what the attention kernel consumes.
#include <string.h>

#include <vector>

#include "hvx.h"
#include "hvx_device.h"
#include "hvx_kernels.h"

using namespace hvx;

struct hvx_flow {
    hvx_flow_config c;
    std::vector<const void*> w;
    // adaLN modulation cache for the CFG-2 solver: the Euler t-grid is a constant of the model (flow_matching.py:225-227),
    // so the 22 x 6D + 2D modulation vectors of each step are computed once and reused by every later utterance.
    // A slot is registered only AFTER the launches that fill it have been enqueued without error (a failed solve must not leave a
    // "valid" slot of garbage behind), and carries the event recorded behind those launches: a hit on another stream waits for it.
    float* mod_cache = nullptr;
    int mod_slots = 0;
    bool half_stream = false;          // residual stream of the DiT blocks stored as fp16 (bf16 mode only)
    bool f32_small = false;            // time MLP, adaLN modulation Linears, input and output projection in fp32 (their weights were packed as fp32): bf16 mode only
    bool f16_linears = false;          // QKV, FF1 and FF2 of every DiT block take IEEE fp16 operands (their weights were packed as fp16): bf16 mode only
    struct ModSlot {
        float t;
        hipStream_t s;
        hipEvent_t ev;
    };
    std::vector<ModSlot> mod_t;
    void drop_mods() {
        for (auto& m : mod_t)
            if (m.ev) hipEventDestroy(m.ev);
        mod_t.clear();
    }
    size_t mod_slot_floats() const { return (size_t)c.depth * 2 * 6 * c.dim + (size_t)2 * 2 * c.dim; }
};

namespace {

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Carve {
    char* base;
    size_t off = 0;
    explicit Carve(char* b) : base(b) {}
    template <class T> T* take(size_t bytes) {
        T* p = reinterpret_cast<T*>(base ? base + off : nullptr);
        off += align_up(bytes);
        return p;
    }
};

struct EstBufs {
    void *tsin, *th, *tsilu, *hin, *x0t, *c1, *n, *q, *k, *vT, *att, *ffh;
    float *mods, *fmod, *x, *outrow;
    // solver extras
    float *x_in, *mu_in, *spk_in, *cond_in, *t_in;
    int* kv_in;
    // adaLN modulation vectors: [depth][mod_rows][6D] + final [mod_rows][2D]; mod_bs = 6D (a row per batch entry) or 0 (one row for all:
    // the solver, where every batch entry sits at the same step time t)
    long long mod_bs = -1;
    int mod_rows = 0;
    // encoder extras
    float *e_h0, *e_out;
    void *e_h0t, *e_c1;
    int t_pad;
};

size_t carve_est(const hvx_flow_config& c, char* base, int B, int T, EstBufs& b) {
    const size_t es = dtype_size(c.dtype);
    const int D = c.dim, H = c.heads;
    const int Tp = (T + 63) & ~63;                 // the LDS-staged attention walks 64-key tiles
    b.t_pad = Tp;
    Carve cv(base);
    b.tsin = cv.take<void>((size_t)B * c.time_freq_dim * 4);      // (4-byte elements: these and hin are fp32 when the small Linears run in fp32)
    b.th = cv.take<void>((size_t)B * D * 4);
    b.tsilu = cv.take<void>((size_t)B * D * 4);
    b.mods = cv.take<float>((size_t)c.depth * B * 6 * D * 4);
    b.fmod = cv.take<float>((size_t)B * 2 * D * 4);
    b.hin = cv.take<void>((size_t)B * T * 4 * c.mel * 4);
    b.x0t = cv.take<void>((size_t)B * T * D * es);
    b.c1 = cv.take<void>((size_t)B * T * D * es);
    b.x = cv.take<float>((size_t)B * T * D * 4);
    b.n = cv.take<void>((size_t)B * T * D * es);
    b.q = cv.take<void>((size_t)B * H * Tp * 64 * es);
    b.k = cv.take<void>((size_t)B * H * Tp * 64 * es);
    b.vT = cv.take<void>((size_t)B * H * Tp * 64 * es);
    b.att = cv.take<void>((size_t)B * T * D * es);
    b.ffh = cv.take<void>((size_t)B * T * c.ff * es);
    b.outrow = cv.take<float>((size_t)B * T * c.mel * 4);
    const int Bs = B < 2 ? 2 : B;
    b.x_in = cv.take<float>((size_t)Bs * c.mel * T * 4);
    b.mu_in = cv.take<float>((size_t)Bs * c.mel * T * 4);
    b.spk_in = cv.take<float>((size_t)Bs * c.mel * 4);
    b.cond_in = cv.take<float>((size_t)Bs * c.mel * T * 4);
    b.t_in = cv.take<float>(64);
    b.kv_in = cv.take<int>((size_t)Bs * 4);
    return cv.off;
}

size_t carve_enc(const hvx_flow_config& c, char* base, int n, EstBufs& b) {
    const size_t es = dtype_size(c.dtype);
    const int melp = (c.mel + 31) & ~31;
    Carve cv(base);
    b.e_h0 = cv.take<float>((size_t)n * c.mel * 4);
    b.e_h0t = cv.take<void>((size_t)n * melp * es);
    b.e_c1 = cv.take<void>((size_t)n * c.pla_channels * es);
    b.e_out = cv.take<float>((size_t)n * c.mel * 4);
    return cv.off;
}

GemmArgs linear(int dtype, int M, int N, int K, const void* A, int lda, const void* W, const float* bias) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.dtype = dtype; g.M = M; g.N = N; g.K = K; g.batch = 1; g.groups = 1;
    g.A = A; g.lda = lda; g.rows_in = M; g.cin_pad = K; g.conv_stride = 1; g.conv_dil = 1; g.pad_left = 0; g.up = 1;
    g.W = W; g.epi = EPI_GENERIC; g.bias = bias; g.scale = 1.0f;
    return g;
}

// ---- tiny kernels local to the flow ------------------------------------------------------------------------
__global__ void spk_affine_kernel(const float* emb, const float* W, const float* b, float* out, int in_dim, int out_dim) {
    __shared__ float red[4];
    __shared__ float xn[1024];
    float ss = 0.0f;
    for (int i = threadIdx.x; i < in_dim; i += 256) ss += emb[i] * emb[i];
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);      // F.normalize eps
    for (int i = threadIdx.x; i < in_dim; i += 256) xn[i] = emb[i] / nrm;
    __syncthreads();
    for (int o = threadIdx.x; o < out_dim; o += 256) {
        float acc = 0.0f;
        for (int i = 0; i < in_dim; ++i) acc += W[(long long)o * in_dim + i] * xn[i];
        out[o] = acc + b[o];
    }
}

__global__ void token_embed_kernel(const float* table, const int* tok, float* out, int n, int mel) {
    const int i = blockIdx.x;
    int t = tok[i];
    t = t < 0 ? 0 : t;                                   // torch.clamp(token, min=0)  (flow.py:398)
    for (int c = threadIdx.x; c < mel; c += blockDim.x) out[(long long)i * mel + c] = table[(long long)t * mel + c];
}

__global__ void mu_expand_kernel(const float* h, float* mu, int n, int mel, int ratio) {
    // mu[c][ratio*i + r] = h[i][c]   (repeat_interleave along time, then transpose; flow.py:405, 421)
    const int T = n * ratio;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (t < T) mu[(long long)c * T + t] = h[(long long)(t / ratio) * mel + c];
}

__global__ void fill_kernel(float* p, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

#define HVX_CHECK(x) do { if (x) return -1; } while (0)
#define HIP_OK(x) do { if ((x) != hipSuccess) return set_error("hip call failed: %s", #x), -1; } while (0)

// n rows in, n_out rows out: n_out == n zero-pads the look-ahead on the right (finalize), n_out == n - pla_len takes the last pla_len
// rows of x as the look-ahead context (upsample_encoder.py:90-95)
int prelookahead(const hvx_flow* h, hipStream_t s, EstBufs& b, const float* x, int n, int n_out, float* y) {
    const hvx_flow_config& c = h->c;
    const int melp = (c.mel + 31) & ~31;
    const void* const* w = h->w.data();
    HVX_CHECK(launch_rows_to_dtype(x, c.mel, b.e_h0t, c.dtype, melp, n, c.mel, melp, s));
    // conv1: Conv1d(mel -> C, k = len+1), input right-padded by `len` zeros, LeakyReLU(0.01)
    GemmArgs g = linear(c.dtype, n_out, c.pla_channels, (c.pla_len + 1) * melp, b.e_h0t, melp, w[5], (const float*)w[6]);
    g.cin_pad = melp; g.rows_in = n;
    g.act = ACT_LRELU; g.act_param = 0.01f;
    g.out = b.e_c1; g.out_f32 = 0; g.ldo = c.pla_channels; g.out_cols = c.pla_channels;
    HVX_CHECK(launch_gemm(g, s));
    // conv2: left pad 2, Conv1d(C -> mel, k = 3), + residual
    g = linear(c.dtype, n_out, c.mel, 3 * c.pla_channels, b.e_c1, c.pla_channels, w[7], (const float*)w[8]);
    g.cin_pad = c.pla_channels; g.rows_in = n_out; g.pad_left = 2;
    g.res = x; g.ldres = c.mel;
    g.out = y; g.out_f32 = 1; g.ldo = c.mel; g.out_cols = c.mel;
    HVX_CHECK(launch_gemm(g, s));
    return 0;
}

// estimator on time-major rows; leaves v in b.outrow [B][T][mel]
int estimator_core(const hvx_flow* h, hipStream_t s, EstBufs& b, int B, int T, const float* x, const int* kv_len, const float* mu,
                   const float* t, const float* spks, const float* cond, bool skip_mods = false, int chunk = 0) {
    const hvx_flow_config& c = h->c;
    const int dt = c.dtype, D = c.dim, H = c.heads, Tp = b.t_pad, mel = c.mel;
    // rows of modulation vectors that are computed (from t[0 .. MB)) and how the batch entries index them
    const int MB = b.mod_rows > 0 ? b.mod_rows : B;
    const long long mbs = b.mod_bs >= 0 ? b.mod_bs : 6LL * D, fbs = b.mod_bs >= 0 ? (b.mod_bs ? 2LL * D : 0) : 2LL * D;
    const size_t es = dtype_size(dt);
    const void* const* w = h->w.data();
    if (T > c.max_t) return set_error("estimator: T=%d exceeds max_t=%d", T, c.max_t), -1;

    // ---- time embedding -> SiLU(t_emb) -> all adaLN modulation vectors --------------------------------------------
    GemmArgs g;
    const void* const* tw = w + 19 + 10 * c.depth;
    // operand type of the SMALL Linears (time MLP, adaLN modulation, input / output projection): exact fp32 when the handle was told so — with fp16 block
    // Linears these carry the bf16 mode's remaining distance from fp32 (tools/dit_rounding_study.py); a few ms per solve
    const int sdt = (h->f32_small && dt == DT_BF16) ? DT_F32 : dt;
    if (!skip_mods) {
    HVX_CHECK(launch_time_sinus(t, b.tsin, sdt, MB, c.time_freq_dim, s));
    g = linear(sdt, MB, D, c.time_freq_dim, b.tsin, c.time_freq_dim, w[9], (const float*)w[10]);
    g.act = ACT_SILU; g.out = b.th; g.ldo = D; g.out_cols = D;
    HVX_CHECK(launch_gemm(g, s));
    g = linear(sdt, MB, D, D, b.th, D, w[11], (const float*)w[12]);
    g.out2 = b.tsilu; g.act2 = ACT_SILU; g.ldo2 = D; g.out2_cols = D;
    HVX_CHECK(launch_gemm(g, s));
    for (int i = 0; i < c.depth; ++i) {
        const void* const* bw = w + 19 + 10 * i;
        g = linear(sdt, MB, 6 * D, D, b.tsilu, D, bw[0], (const float*)bw[1]);
        g.out = b.mods + (size_t)i * MB * 6 * D; g.out_f32 = 1; g.ldo = 6 * D; g.out_cols = 6 * D;
        HVX_CHECK(launch_gemm(g, s));
    }
    g = linear(sdt, MB, 2 * D, D, b.tsilu, D, tw[0], (const float*)tw[1]);
    g.out = b.fmod; g.out_f32 = 1; g.ldo = 2 * D; g.out_cols = 2 * D;
    HVX_CHECK(launch_gemm(g, s));
    }

    // ---- input embedding: Linear(cat[x, cond, mu, spks]) + causal grouped conv position embedding ------------------
    HVX_CHECK(launch_dit_concat(x, cond, mu, spks, b.hin, sdt, B, T, mel, s));
    const int IN = 4 * mel;
    g = linear(sdt, T, D, IN, b.hin, IN, w[13], (const float*)w[14]);
    g.batch = B; g.a_bs = (long long)T * IN;
    g.out = b.x; g.out_f32 = 1; g.out_bs = (long long)T * D; g.ldo = D; g.out_cols = D;           // x0 (fp32, residual of the conv branch)
    if (sdt == dt) {
        g.out2 = b.x0t; g.act2 = ACT_NONE; g.out2_bs = (long long)T * D; g.ldo2 = D; g.out2_cols = D;
        HVX_CHECK(launch_gemm(g, s));
    } else {
        // fp32 operands on the bf16 matrix cores as (hi, lo) pairs (1e-6, a third of the exact form's time) where ONE batch entry already fills the
        // split form's tile grid: the choice must not depend on the batch, or an utterance's mel would depend on its neighbours in the solve
        g.x3 = ((long long)((T + 127) / 128) * ((D + 127) / 128) >= 256) ? 1 : 0;
        HVX_CHECK(launch_gemm(g, s));
        HVX_CHECK(launch_rows_to_dtype(b.x, D, b.x0t, dt, D, B * T, D, D, s));     // the position-embedding convolutions read x0 in `dtype`
    }
    const int Cg = D / c.conv_groups, kc = c.conv_kernel;
    // the residual stream of the blocks: fp32 in b.x, or fp16 in the memory of x0t (the bf16 copy of x0 is dead once the first conv has read it)
    const int hs = (h->half_stream && dt == DT_BF16) ? 1 : 0;
    void* const xs = hs ? b.x0t : static_cast<void*>(b.x);
    for (int pass = 0; pass < 2; ++pass) {
        g = linear(dt, T, Cg, kc * Cg, pass == 0 ? b.x0t : b.c1, D, w[15 + 2 * pass], (const float*)w[16 + 2 * pass]);
        g.batch = B; g.groups = c.conv_groups; g.a_bs = (long long)T * D; g.a_gs = Cg; g.rows_in = T; g.cin_pad = Cg; g.pad_left = kc - 1;
        g.w_gs = (long long)Cg * kc * Cg;
        g.act = ACT_MISH;
        if (pass == 0) {
            g.out = b.c1; g.out_f32 = 0; g.out_bs = (long long)T * D; g.ldo = D; g.out_cols = D;
        } else {
            g.res = b.x; g.res_bs = (long long)T * D; g.ldres = D;                                // conv_pos_embed(x) + x  (dit.py:97)
            g.out = xs; g.out_f32 = hs ? 0 : 1; g.out_f16 = hs; g.out_bs = (long long)T * D; g.ldo = D; g.out_cols = D;
        }
        HVX_CHECK(launch_gemm(g, s));
    }

    // ---- DiT blocks ---------------------------------------------------------------------------------------------------
    // operand type of the QKV / FF1 / FF2 Linears of a block (and of the activations that feed them: adaLN outputs, FF hidden); the attention and the
    // Linear behind it stay bf16
    const int ldt = (h->f16_linears && dt == DT_BF16) ? DT_F16 : dt;
    HIP_OK(hipMemsetAsync(b.vT, 0, (size_t)B * H * Tp * 64 * es, s));       // padded key columns must be finite
    for (int i = 0; i < c.depth; ++i) {
        const void* const* bw = w + 19 + 10 * i;
        const float* mod = b.mods + (size_t)i * MB * 6 * D;
        HVX_CHECK(launch_layernorm_mod(xs, hs, mod, mod + D, mbs, 1e-6f, b.n, ldt, B, T, D, s));
        g = linear(ldt, T, 3 * D, D, b.n, D, bw[2], (const float*)bw[3]);
        g.batch = B; g.a_bs = (long long)T * D; g.epi = EPI_QKV_DIT;
        g.q = b.q; g.k = b.k; g.vT = b.vT; g.heads = H; g.t_pad = Tp; g.rope_cos = (const float*)w[0]; g.rope_sin = (const float*)w[1];
        g.q_scale = 0.125f * 1.4426950408889634f;              // dim_head^-0.5 (modules.py:391, SDPA default scale) in log2 units, rounded once with q
        HVX_CHECK(launch_gemm(g, s));
        AttnArgs at;
        memset(&at, 0, sizeof(at));
        at.dtype = dt; at.batch = B; at.heads = H; at.n_rows = T; at.kn = T;
        at.q = b.q; at.q_bs = (long long)H * Tp * 64; at.q_hs = (long long)Tp * 64; at.q_lo = 64;
        at.k = b.k; at.k_bs = at.q_bs; at.k_hs = at.q_hs;
        at.vT = b.vT; at.v_bs = at.q_bs; at.v_hs = (long long)64 * Tp; at.v_ld = Tp;
        at.kv_len = kv_len; at.kv_len_const = T; at.causal = 0; at.chunk = chunk; at.scale = 0.125f; at.q_log2 = 1;
        at.out = b.att; at.o_bs = (long long)T * D; at.o_hs = 64; at.o_lo = D; at.n_splits = 1;
        HVX_CHECK(launch_attention(at, s));
        g = linear(dt, T, D, D, b.att, D, bw[4], (const float*)bw[5]);      // (bf16 operands: the attention writes bf16, which is exact in fp16 — fp16 weights here change nothing)
        g.batch = B; g.a_bs = (long long)T * D;
        g.gate = mod + 2 * D; g.gate_bs = mbs; g.res = static_cast<const float*>(xs); g.res_f16 = hs; g.res_bs = (long long)T * D; g.ldres = D;
        g.out = xs; g.out_f32 = hs ? 0 : 1; g.out_f16 = hs; g.out_bs = (long long)T * D; g.ldo = D; g.out_cols = D;
        HVX_CHECK(launch_gemm(g, s));
        HVX_CHECK(launch_layernorm_mod(xs, hs, mod + 3 * D, mod + 4 * D, mbs, 1e-6f, b.n, ldt, B, T, D, s));
        g = linear(ldt, T, c.ff, D, b.n, D, bw[6], (const float*)bw[7]);
        g.batch = B; g.a_bs = (long long)T * D; g.act = ACT_GELU_TANH;
        g.out = b.ffh; g.out_f32 = 0; g.out_bs = (long long)T * c.ff; g.ldo = c.ff; g.out_cols = c.ff;
        HVX_CHECK(launch_gemm(g, s));
        g = linear(ldt, T, D, c.ff, b.ffh, c.ff, bw[8], (const float*)bw[9]);
        g.batch = B; g.a_bs = (long long)T * c.ff;
        g.gate = mod + 5 * D; g.gate_bs = mbs; g.res = static_cast<const float*>(xs); g.res_f16 = hs; g.res_bs = (long long)T * D; g.ldres = D;
        g.out = xs; g.out_f32 = hs ? 0 : 1; g.out_f16 = hs; g.out_bs = (long long)T * D; g.ldo = D; g.out_cols = D;
        HVX_CHECK(launch_gemm(g, s));
    }
    // ---- final adaLN (scale first, then shift: modules.py:262) + projection -------------------------------------------
    // (when the output projection runs in fp32 the last adaLN output is fp32 too: it goes into the FF hidden buffer, dead after the last block and at
    // least as large — ff x 2 bytes >= dim x 4 bytes for every ff_mult >= 2)
    const bool f32_out = sdt != dt;
    if (f32_out && (size_t)c.ff * es < (size_t)D * 4) return set_error("estimator: f32_small needs ff >= 2 dim"), -1;
    void* const nl = f32_out ? b.ffh : b.n;
    HVX_CHECK(launch_layernorm_mod(xs, hs, b.fmod + D, b.fmod, fbs, 1e-6f, nl, f32_out ? (int)DT_F32 : dt, B, T, D, s));
    g = linear(f32_out ? (int)DT_F32 : dt, T, mel, D, nl, D, tw[2], (const float*)tw[3]);
    g.x3 = 0;                                                  // (N = mel: exact fp32 form whatever the batch — same reason)
    g.batch = B; g.a_bs = (long long)T * D;
    g.out = b.outrow; g.out_f32 = 1; g.out_bs = (long long)T * mel; g.ldo = mel; g.out_cols = mel;
    HVX_CHECK(launch_gemm(g, s));
    return 0;
}

}  // namespace

extern "C" {

int hvx_flow_create(const hvx_flow_config* cfg, const void* const* weights, int32_t n_weights, hvx_flow** out) {
    if (!cfg || !weights || !out) return set_error("hvx_flow_create: null argument"), -1;
    const int expect = 19 + 10 * cfg->depth + 4;
    if (n_weights != expect) return set_error("hvx_flow_create: expected %d weight pointers, got %d", expect, n_weights), -1;
    if (cfg->dim != cfg->heads * 64 || cfg->dim % cfg->conv_groups || (cfg->dim / cfg->conv_groups) % 32 || cfg->pla_channels % 32 ||
        cfg->ff % 32 || (4 * cfg->mel) % 32 || cfg->time_freq_dim % 32 || cfg->spk_dim > 1024)
        return set_error("hvx_flow_create: unsupported dimensions"), -1;
    for (int i = 0; i < n_weights; ++i)
        if (!weights[i]) return set_error("hvx_flow_create: weight %d is null", i), -1;
    hvx_flow* h = new hvx_flow();
    h->c = *cfg;
    h->w.assign(weights, weights + n_weights);
    *out = h;
    return 0;
}
void hvx_flow_destroy(hvx_flow* h) {
    if (h) h->drop_mods();
    delete h;
}

size_t hvx_flow_workspace_bytes(const hvx_flow* h, int32_t batch, int32_t t) {
    EstBufs b;
    const size_t a = carve_est(h->c, nullptr, batch, t, b);
    const size_t e = carve_enc(h->c, nullptr, t, b);
    return (a > e ? a : e) + 256;
}

int hvx_flow_prelookahead(hvx_flow* h, hvx_stream s, void* ws, size_t ws_bytes, const float* x, int32_t n, float* y) {
    EstBufs b;
    if (carve_enc(h->c, (char*)ws, n, b) > ws_bytes) return set_error("hvx_flow_prelookahead: workspace too small"), -1;
    return prelookahead(h, (hipStream_t)s, b, x, n, n, y);
}

int hvx_flow_prelookahead_context(hvx_flow* h, hvx_stream s, void* ws, size_t ws_bytes, const float* x, int32_t n, float* y) {
    EstBufs b;
    if (n <= h->c.pla_len) return set_error("hvx_flow_prelookahead_context: needs more than %d rows", h->c.pla_len), -1;
    if (carve_enc(h->c, (char*)ws, n, b) > ws_bytes) return set_error("hvx_flow_prelookahead_context: workspace too small"), -1;
    return prelookahead(h, (hipStream_t)s, b, x, n, n - h->c.pla_len, y);
}

static int encode_impl(hvx_flow* h, hvx_stream stream, void* ws, size_t ws_bytes, const int32_t* token, int32_t n, const float* embedding,
                       int finalize, float* mu, float* spk) {
    hipStream_t s = (hipStream_t)stream;
    const int n_out = finalize ? n : n - h->c.pla_len;
    if (n_out <= 0) return set_error("hvx_flow_encode: %d tokens leave nothing after the look-ahead", n), -1;
    const hvx_flow_config& c = h->c;
    EstBufs b;
    if (carve_enc(c, (char*)ws, n, b) > ws_bytes) return set_error("hvx_flow_encode: workspace too small"), -1;
    hipLaunchKernelGGL(spk_affine_kernel, dim3(1), dim3(256), 0, s, embedding, (const float*)h->w[3], (const float*)h->w[4], spk, c.spk_dim, c.mel);
    hipLaunchKernelGGL(token_embed_kernel, dim3(n), dim3(128), 0, s, (const float*)h->w[2], token, b.e_h0, n, c.mel);
    HVX_CHECK(prelookahead(h, s, b, b.e_h0, n, n_out, b.e_out));
    const int T = 2 * n_out;
    hipLaunchKernelGGL(mu_expand_kernel, dim3((T + 255) / 256, c.mel), dim3(256), 0, s, b.e_out, mu, n_out, c.mel, 2);
    HIP_OK(hipGetLastError());
    return 0;
}

int hvx_flow_encode(hvx_flow* h, hvx_stream stream, void* ws, size_t ws_bytes, const int32_t* token, int32_t n, const float* embedding,
                    float* mu, float* spk) {
    return encode_impl(h, stream, ws, ws_bytes, token, n, embedding, 1, mu, spk);
}
int hvx_flow_encode_chunk(hvx_flow* h, hvx_stream stream, void* ws, size_t ws_bytes, const int32_t* token, int32_t n, const float* embedding,
                          int32_t finalize, float* mu, float* spk) {
    return encode_impl(h, stream, ws, ws_bytes, token, n, embedding, finalize, mu, spk);
}

int hvx_cfm_estimator_streaming(hvx_flow* h, hvx_stream stream, void* ws, size_t ws_bytes, int32_t batch, int32_t t_len, const float* x,
                                const int32_t* kv_len, const float* mu, const float* t, const float* spks, const float* cond,
                                int32_t static_chunk_size, float* out) {
    hipStream_t s = (hipStream_t)stream;
    EstBufs b;
    if (static_chunk_size < 0) return set_error("hvx_cfm_estimator: negative chunk size"), -1;
    if (carve_est(h->c, (char*)ws, batch, t_len, b) > ws_bytes) return set_error("hvx_cfm_estimator: workspace too small"), -1;
    HVX_CHECK(estimator_core(h, s, b, batch, t_len, x, kv_len, mu, t, spks, cond, false, static_chunk_size));
    for (int bi = 0; bi < batch; ++bi)
        HVX_CHECK(launch_transpose_f32(b.outrow + (size_t)bi * t_len * h->c.mel, out + (size_t)bi * t_len * h->c.mel, t_len, h->c.mel, h->c.mel, t_len, s));
    return 0;
}

int hvx_cfm_estimator(hvx_flow* h, hvx_stream stream, void* ws, size_t ws_bytes, int32_t batch, int32_t t_len, const float* x,
                      const int32_t* kv_len, const float* mu, const float* t, const float* spks, const float* cond, float* out) {
    return hvx_cfm_estimator_streaming(h, stream, ws, ws_bytes, batch, t_len, x, kv_len, mu, t, spks, cond, 0, out);
}

int hvx_flow_set_half_stream(hvx_flow* h, int32_t on) {
    if (!h) return set_error("hvx_flow_set_half_stream: null handle"), -1;
    if (on && h->c.dtype != DT_BF16) return set_error("hvx_flow_set_half_stream: the fp16 residual stream belongs to the bf16 mode"), -1;
    h->half_stream = on != 0;
    return 0;
}

int hvx_flow_set_f32_small(hvx_flow* h, int32_t on) {
    if (!h) return set_error("hvx_flow_set_f32_small: null handle"), -1;
    if (on && h->c.dtype != DT_BF16) return set_error("hvx_flow_set_f32_small: belongs to the bf16 mode"), -1;
    h->f32_small = on != 0;
    h->drop_mods();                                     // cached modulation vectors were computed in the other arithmetic
    return 0;
}

int hvx_flow_set_f16_linears(hvx_flow* h, int32_t on) {
    if (!h) return set_error("hvx_flow_set_f16_linears: null handle"), -1;
    if (on && h->c.dtype != DT_BF16) return set_error("hvx_flow_set_f16_linears: fp16 Linear operands belong to the bf16 mode"), -1;
    if (on && ((h->c.dim & 63) || h->c.dim < 256 || (h->c.ff & 63))) return set_error("hvx_flow_set_f16_linears: needs dim >= 256 and dim, ff multiples of 64"), -1;
    h->f16_linears = on != 0;
    return 0;
}

int hvx_flow_set_mod_cache(hvx_flow* h, void* buf, size_t bytes) {
    if (!h) return set_error("hvx_flow_set_mod_cache: null handle"), -1;
    h->mod_cache = (float*)buf;
    h->mod_slots = buf ? (int)(bytes / (h->mod_slot_floats() * 4)) : 0;
    h->drop_mods();
    return 0;
}

// x[u][ch][t] += dt * ((1 + rate) * v[u][t][ch] - rate * v[n + u][t][ch])   (flow_matching.py:116-120), all utterances of a batch
__global__ void cfg_euler_batch_kernel(float* x, const float* v, int n, int T, int mel, float dt, float rate) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y, u = blockIdx.z;
    if (t >= T) return;
    const float vc = v[((long long)u * T + t) * mel + ch], vu = v[((long long)(n + u) * T + t) * mel + ch];
    float* px = x + ((long long)u * mel + ch) * T + t;
    *px = *px + dt * ((1.0f + rate) * vc - rate * vu);
}

__global__ void kv_len_pair_kernel(const int* t_len, int* kv, int n, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * n) kv[i] = t_len ? t_len[i % n] : T;
}

int hvx_cfm_solve_batch(hvx_flow* h, hvx_stream stream, void* ws, size_t ws_bytes, int32_t n, int32_t T, const int32_t* t_len, float* x,
                        const float* mu, const float* spks, const float* cond, int32_t n_steps, const float* t_steps, const float* dt_steps,
                        int32_t static_chunk_size) {
    hipStream_t s = (hipStream_t)stream;
    if (!h) return set_error("hvx_cfm_solve: null handle"), -1;
    const hvx_flow_config& c = h->c;
    // everything that can be refused is refused before any state (workspace, modulation cache) is touched
    if (static_chunk_size < 0) return set_error("hvx_cfm_solve: negative chunk size"), -1;
    if (n < 1 || T <= 0 || T > c.max_t) return set_error("hvx_cfm_solve: %d utterances of T=%d outside (0, max_t=%d]", n, T, c.max_t), -1;
    if (n_steps <= 0 || !x || !mu || !spks || !cond || !t_steps || !dt_steps) return set_error("hvx_cfm_solve: bad arguments"), -1;
    EstBufs b;
    const int B = 2 * n;
    if (carve_est(c, (char*)ws, B, T, b) > ws_bytes) return set_error("hvx_cfm_solve: workspace too small"), -1;
    b.mod_rows = 2;                    // every batch entry sits at the same step time: one modulation row serves all (two are kept: the cache layout)
    b.mod_bs = 0;
    const size_t plane = (size_t)c.mel * T * 4;
    // rows [0, n): conditional, rows [n, 2n): unconditional (zeros)  (flow_matching.py:95-108)
    HIP_OK(hipMemsetAsync(b.mu_in, 0, B * plane, s));
    HIP_OK(hipMemsetAsync(b.cond_in, 0, B * plane, s));
    HIP_OK(hipMemsetAsync(b.spk_in, 0, (size_t)B * c.mel * 4, s));
    HIP_OK(hipMemcpyAsync(b.mu_in, mu, n * plane, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(b.cond_in, cond, n * plane, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(b.spk_in, spks, (size_t)n * c.mel * 4, hipMemcpyDeviceToDevice, s));
    const int* kv = nullptr;
    if (t_len) {
        hipLaunchKernelGGL(kv_len_pair_kernel, dim3((B + 63) / 64), dim3(64), 0, s, t_len, b.kv_in, n, T);
        kv = b.kv_in;
    }
    for (int st = 0; st < n_steps; ++st) {
        HIP_OK(hipMemcpyAsync(b.x_in, x, n * plane, hipMemcpyDeviceToDevice, s));
        HIP_OK(hipMemcpyAsync((char*)b.x_in + n * plane, x, n * plane, hipMemcpyDeviceToDevice, s));
        bool hit = false;
        int fill_slot = -1;
        float* ws_mods = b.mods;
        float* ws_fmod = b.fmod;
        if (h->mod_cache) {
            int slot = -1;
            for (size_t e = 0; e < h->mod_t.size(); ++e)
                if (memcmp(&h->mod_t[e].t, &t_steps[st], 4) == 0) slot = (int)e;
            hit = slot >= 0;
            if (hit && h->mod_t[slot].s != s) HIP_OK(hipStreamWaitEvent(s, h->mod_t[slot].ev, 0));     // filled on another stream
            if (!hit && (int)h->mod_t.size() < h->mod_slots) slot = fill_slot = (int)h->mod_t.size();
            if (slot >= 0) {
                b.mods = h->mod_cache + (size_t)slot * h->mod_slot_floats();
                b.fmod = b.mods + (size_t)c.depth * 2 * 6 * c.dim;
            }
        }
        if (!hit) hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, s, b.t_in, 2, t_steps[st]);
        const int rc = estimator_core(h, s, b, B, T, b.x_in, kv, b.mu_in, b.t_in, b.spk_in, b.cond_in, hit, static_chunk_size);
        b.mods = ws_mods;
        b.fmod = ws_fmod;
        if (rc) return -1;
        if (fill_slot >= 0) {                      // the slot's contents are enqueued: register it, with the event a reader on another stream waits for
            hvx_flow::ModSlot m{t_steps[st], s, nullptr};
            if (hipEventCreateWithFlags(&m.ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(m.ev, s) != hipSuccess) {
                if (m.ev) hipEventDestroy(m.ev);
                return set_error("hvx_cfm_solve: event for the modulation cache failed"), -1;
            }
            h->mod_t.push_back(m);
        }
        hipLaunchKernelGGL(cfg_euler_batch_kernel, dim3((T + 255) / 256, c.mel, n), dim3(256), 0, s, x, b.outrow, n, T, c.mel, dt_steps[st], c.cfg_rate);
        HIP_OK(hipGetLastError());
    }
    return 0;
}

int hvx_cfm_solve_streaming(hvx_flow* h, hvx_stream stream, void* ws, size_t ws_bytes, int32_t T, float* x, const float* mu, const float* spks,
                            const float* cond, int32_t n_steps, const float* t_steps, const float* dt_steps, int32_t static_chunk_size) {
    return hvx_cfm_solve_batch(h, stream, ws, ws_bytes, 1, T, nullptr, x, mu, spks, cond, n_steps, t_steps, dt_steps, static_chunk_size);
}

int hvx_cfm_solve(hvx_flow* h, hvx_stream stream, void* ws, size_t ws_bytes, int32_t T, float* x, const float* mu, const float* spks,
                  const float* cond, int32_t n_steps, const float* t_steps, const float* dt_steps) {
    return hvx_cfm_solve_batch(h, stream, ws, ws_bytes, 1, T, nullptr, x, mu, spks, cond, n_steps, t_steps, dt_steps, 0);
}

}  // extern "C"
