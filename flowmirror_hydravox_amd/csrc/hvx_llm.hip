// hvx_llm.hip — KV-cached multi-head AR step of the speech-token LM (include/hvx.h: hvx_llm_*).
//
// Restates the per-step math of CosyVoice3LM.inference_wrapper
// (server/model_utils/cosyvoice/llm/llm_multi_head_v3.py:871-888): Qwen2 backbone (HF Qwen2: RMSNorm,
// q/k/v Linear+bias, rotate-half RoPE, GQA attention, o Linear, SwiGLU MLP), final RMSNorm, the K degenerate
// MTP decoder layers (length-1 sequence => h1 = y + Wo(Wv n1 + bv)), llm_decoder, log_softmax.
// The reference recomputes the full prefix every step (cache=None, :873-882); with a causal mask that equals
// this KV-cached evaluation of the new rows only.
//
// Per layer, five GEMM / attention launches (+ the split combine): QKV (RMSNorm folded in, +bias +RoPE +KV append) | GQA-packed
// attention | o_proj (+residual, in place) | gate/up (RMSNorm folded in, +SwiGLU) | down (+residual).  The residual stream x stays
// fp32 (bf16 mode keeps a bf16 copy beside it for the fused-norm GEMMs); GEMM operands are `dtype` (bf16 production / f32 parity).
// hvx_llm_decode_steps replays {forward, RAS sampler, advance} as one hipGraph per step: the decode loop lives on the device.
#include <stdlib.h>
#include <string.h>

#include <map>
#include <tuple>
#include <vector>

#include "hvx.h"
#include "hvx_kernels.h"

using namespace hvx;

struct GraphKey {
    int n_seq, kn, head_k;
    const void *tok, *ctrl, *logp;
    hipStream_t s;
    bool operator<(const GraphKey& o) const {
        return std::tie(n_seq, kn, head_k, tok, ctrl, logp, s) < std::tie(o.n_seq, o.kn, o.head_k, o.tok, o.ctrl, o.logp, o.s);
    }
};

struct hvx_llm {
    hvx_llm_config c;
    std::vector<const void*> w;
    bool use_graph = false;
    const void* head_mlp_codes = nullptr;    // hvx_llm_set_head_mlp_fp8: e4m3 copy of the heads' gate / up projection + per-column scales
    const float* head_mlp_scales = nullptr;
    std::map<GraphKey, hipGraphExec_t> graphs;
    struct StepGraph {
        hvx_decode_args a;
        hipStream_t s;
        hipGraphExec_t ex;
    };
    std::vector<StepGraph> step_graphs;      // whole decode steps (forward + sampler + advance), keyed on the argument block
    void drop_graphs() {
        for (auto& kv : graphs) hipGraphExecDestroy(kv.second);
        graphs.clear();
        for (auto& g : step_graphs) hipGraphExecDestroy(g.ex);
        step_graphs.clear();
    }
    // bound buffers
    char* ws = nullptr;
    size_t ws_bytes = 0;
    int max_seq = 0, max_rows = 0, n_slots = 0, max_ctx = 0;
    void* kcache = nullptr;
    void* vTcache = nullptr;
    // workspace carve
    float* x = nullptr;          // [R][H]
    void* a = nullptr;           // [R][H]      dtype: copy of x for the fused-norm GEMMs (bf16 mode)
    void* qbuf = nullptr;        // [R][q*64]   dtype
    void* attn = nullptr;        // [R][q*64]   dtype
    void* hmlp = nullptr;        // [R][inter]  dtype
    float* part = nullptr;       // split-K partials (shared by backbone and heads)
    float* att_o = nullptr;      // attention split partials
    float* att_ml = nullptr;
    float* ylast = nullptr;      // [S][H] f32 (post final norm)
    float* hx = nullptr;         // [hn][S][H] f32
    void* ha = nullptr;          // [hn][S][H]  dtype
    void* hv = nullptr;          // [hn][S][A]  dtype
    void* hm = nullptr;          // [hn][S][mtp_inter] dtype
    float* logits = nullptr;     // [S][hn][vocab_pad] f32
    int att_splits = 1, att_chunk = 0, att_rows_pad = 0;
};

namespace {

constexpr int MAX_SPLIT = 16;
// keys per decode-attention split (one workgroup); its 4 waves take a quarter each and merge in LDS.  256 on the whole chip; a decode engine confined
// to a few compute units (hvx_stream_create_cu_range) wants fewer, longer workgroups (option att_chunk, a multiple of 128; hvx_set_option refuses anything else)
static int att_chunk_env_value() {
    const int v = (int)opt(OPT_ATT_CHUNK);              // (validated by hvx_set_option: 0 or a multiple of 128)
    return v;
}
static bool att_chunk_from_env() { return att_chunk_env_value() != 0; }
static int att_chunk_keys() { return att_chunk_from_env() ? att_chunk_env_value() : 256; }

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

int pick_split(int N, int K, int nz) {
    // workgroups = 16-column tiles x cross-block K splits; inside a workgroup the 4 waves split K again.  Aim at >= ~1.5
    // workgroups per CU while leaving every wave at least 2 k-steps.
    const int groups = (N / 16) * nz;
    const int kt = K / 32;
    int s = 384 / (groups > 0 ? groups : 1);
    if (s > kt / 8) s = kt / 8;
    if (s > MAX_SPLIT) s = MAX_SPLIT;
    if (s < 1) s = 1;
    return s;
}

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

size_t carve(hvx_llm* h, char* base, int S, int R, int max_ctx) {
    const hvx_llm_config& c = h->c;
    const size_t es = dtype_size(c.dtype);
    const int H = c.hidden, Q = c.q_heads * 64, A = c.mtp_attn_dim, hn = c.head_num;
    Carve cv(base);
    const size_t R16 = ((size_t)R + 15) / 16 * 16;                    // the fragment-order activation matrices of a wide decode grid come in 16-row tiles
    h->x = cv.take<float>((size_t)R * H * 4);
    h->a = cv.take<void>(R16 * H * es);
    h->qbuf = cv.take<void>((size_t)R * Q * es);
    h->attn = cv.take<void>(R16 * Q * es);
    h->hmlp = cv.take<void>(R16 * c.inter * es);
    const size_t part_rows = (size_t)(R > hn * S ? R : hn * S);
    h->part = cv.take<float>((size_t)MAX_SPLIT * part_rows * H * 4);
    // attention split partials are only used for short query grids (decode): rows per (seq, kv head) = G * kn
    const int G = c.q_heads / c.kv_heads;
    h->att_chunk = att_chunk_keys();
    h->att_splits = (max_ctx + h->att_chunk - 1) / h->att_chunk;
    h->att_rows_pad = ((G * 8 + 31) / 32) * 32;                       // kn <= 8 on the split path
    h->att_o = cv.take<float>((size_t)S * c.kv_heads * h->att_splits * h->att_rows_pad * 64 * 4);
    h->att_ml = cv.take<float>((size_t)S * c.kv_heads * h->att_splits * h->att_rows_pad * 2 * 4);
    h->ylast = cv.take<float>((size_t)S * H * 4);
    h->hx = cv.take<float>((size_t)hn * S * H * 4);
    h->ha = cv.take<void>((size_t)hn * (((size_t)S + 15) / 16 * 16) * H * es);     // (+ tile padding per head: wide grids read it in fragment order)
    h->hv = cv.take<void>((size_t)hn * S * A * es);
    h->hm = cv.take<void>((size_t)hn * S * c.mtp_inter * es);
    h->logits = cv.take<float>((size_t)S * hn * c.vocab_pad * 4);
    return cv.off;
}

}  // namespace

extern "C" {

int hvx_llm_create(const hvx_llm_config* cfg, const void* const* weights, int32_t n_weights, hvx_llm** out) {
    if (!cfg || !weights || !out) return set_error("hvx_llm_create: null argument"), -1;
    const int expect = 6 + 9 * cfg->layers + 7;
    if (n_weights != expect) return set_error("hvx_llm_create: expected %d weight pointers, got %d", expect, n_weights), -1;
    if (cfg->hidden % 32 || cfg->inter % 128 || (cfg->q_heads * 64) % 128 || cfg->mtp_inter % 16 || cfg->mtp_attn_dim % 32 || cfg->vocab_pad % 16 ||
        cfg->q_heads % cfg->kv_heads || cfg->vocab_pad < cfg->vocab)
        return set_error("hvx_llm_create: unsupported dimensions"), -1;
    for (int i = 0; i < n_weights; ++i)
        if (!weights[i]) return set_error("hvx_llm_create: weight %d is null", i), -1;
    hvx_llm* h = new hvx_llm();
    h->c = *cfg;
    h->w.assign(weights, weights + n_weights);
    *out = h;
    return 0;
}

void hvx_llm_destroy(hvx_llm* h) {
    if (h) h->drop_graphs();
    delete h;
}

size_t hvx_llm_workspace_bytes(const hvx_llm* h, int32_t max_seq, int32_t max_rows, int32_t max_ctx) {
    hvx_llm tmp;
    tmp.c = h->c;
    return carve(&tmp, nullptr, max_seq, max_rows, max_ctx);
}

size_t hvx_llm_kv_bytes(const hvx_llm* h, int32_t n_slots, int32_t max_ctx) {
    const hvx_llm_config& c = h->c;
    return 2 * align_up((size_t)n_slots * c.layers * c.kv_heads * max_ctx * 64 * dtype_size(c.dtype));
}

int hvx_llm_bind(hvx_llm* h, void* workspace, size_t ws_bytes, int32_t max_seq, int32_t max_rows, void* kv, size_t kv_bytes,
                 int32_t n_slots, int32_t max_ctx, hvx_stream s) {
    if (!h || !workspace || !kv) return set_error("hvx_llm_bind: null argument"), -1;
    if (max_ctx % 32 || max_ctx > h->c.max_pos) return set_error("hvx_llm_bind: max_ctx=%d must be a multiple of 32 and <= max_pos=%d", max_ctx, h->c.max_pos), -1;
    const size_t need = carve(h, (char*)workspace, max_seq, max_rows, max_ctx);
    if (need > ws_bytes) return set_error("hvx_llm_bind: workspace %zu < %zu bytes", ws_bytes, need), -1;
    if (hvx_llm_kv_bytes(h, n_slots, max_ctx) > kv_bytes) return set_error("hvx_llm_bind: kv buffer too small"), -1;
    h->drop_graphs();                       // captured launches hold the old buffer addresses
    h->ws = (char*)workspace; h->ws_bytes = ws_bytes;
    h->max_seq = max_seq; h->max_rows = max_rows; h->n_slots = n_slots; h->max_ctx = max_ctx;
    h->kcache = kv;
    h->vTcache = (char*)kv + hvx_llm_kv_bytes(h, n_slots, max_ctx) / 2;
    // padded key columns are multiplied by p == 0: they must be finite
    if (hipMemsetAsync(kv, 0, kv_bytes, (hipStream_t)s) != hipSuccess) return set_error("hvx_llm_bind: memset failed"), -1;
    if (hipMemsetAsync(workspace, 0, need, (hipStream_t)s) != hipSuccess) return set_error("hvx_llm_bind: memset failed"), -1;
    return 0;
}

// Residual projection x += A W^T for more than 32 rows (prefill, large decode batches).  The 4-column kernel above re-reads the [16][K]
// activation tile per (workgroup, 16-row chunk): at 128 rows down_proj moves 348 MB from L2 to the CUs and takes 40 us.  Here the
// 16-column / 64-row form with the weights in MFMA fragment order reads them once per 64 rows; when that leaves too few workgroups K is
// split across workgroups (fp32 partials) and the split reduce — fixed order, one writer per element — does the residual add and
// the operand-type copy.
// m_split: the row count the split-K choice is made for.  A grouped prefill (several prefixes of kn rows each in one forward, llm.py: _prefill_group) passes kn, so
// that every row is summed exactly as in a prefill of its sequence alone: a request's KV cache must not depend on what joined the grid with it.
static int resid_wide(hvx_llm* h, SkinnyArgs g, const void* w_frag, void* xcopy, hipStream_t s, int m_split = 0) {
    const int groups = (g.N / 16) * (((m_split > 0 ? m_split : g.M) + 63) / 64);
    int split = (512 + groups - 1) / groups;
    const int kt = g.K / 32;
    if (split > kt / 8) split = kt / 8;
    if (g.M <= 128 && (g.K & 63) == 0 && g.K / 64 / 12 > 1) split = g.K / 64 / 12;       // mid form (gemm_mid_kernel): a workgroup per 64 columns, ~12 K-tiles of 64 each
    if (split > MAX_SPLIT) split = MAX_SPLIT;
    g.W = w_frag; g.w_narrow = 0;
    if (split <= 1) return launch_skinny(g, s);                      // SK_RESID, one writer per element
    g.split_k = split; g.epi = SK_PARTIAL; g.part = h->part; g.part_zs = 0; g.out = nullptr; g.out2 = nullptr;
    if (launch_skinny(g, s)) return -1;
    ReduceNormArgs r;
    memset(&r, 0, sizeof(r));
    r.x = h->x; r.ldx = g.N; r.part = h->part; r.split_k = split; r.part_stride = (long long)g.M * g.N; r.do_norm = 0;
    r.y = xcopy; r.ldy = g.N; r.dtype = g.dtype; r.M = g.M; r.H = g.N; r.rows_per_z = g.M;
    return launch_reduce_rmsnorm(r, s);
}

static int forward_impl(hvx_llm* h, hipStream_t s, int32_t n_seq, int32_t kn, const int32_t* tok, const int32_t* ctrl, int32_t head_k,
                        float* logp);

// The decode step is ~200 short launches; replaying them as one hipGraph removes the per-launch host cost.  Every kernel
// argument is a fixed device address (control arrays, KV cache, workspace) and every data-dependent quantity (context
// length, active rows) is read from device memory, so one instantiated graph per (grid, pointers) serves the whole utterance.
int hvx_llm_forward(hvx_llm* h, hvx_stream stream, int32_t n_seq, int32_t kn, const int32_t* tok, const int32_t* ctrl, int32_t head_k,
                    float* logp) {
    if (!h || !h->ws) return set_error("hvx_llm_forward: handle not bound"), -1;
    hipStream_t s = (hipStream_t)stream;
    // (capture is illegal on the legacy default stream: such callers get the same launches eagerly)
    if (!h->use_graph || head_k <= 0 || prof_enabled() || s == nullptr) return forward_impl(h, s, n_seq, kn, tok, ctrl, head_k, logp);
    GraphKey key{n_seq, kn, head_k, tok, ctrl, logp, s};
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
        hipGraph_t g = nullptr;
        hipGraphExec_t ex = nullptr;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return set_error("hvx_llm_forward: begin capture failed"), -1;
        const int rc = forward_impl(h, s, n_seq, kn, tok, ctrl, head_k, logp);
        const hipError_t e = hipStreamEndCapture(s, &g);
        if (rc != 0 || e != hipSuccess || !g) {
            if (g) hipGraphDestroy(g);
            if (rc == 0) set_error("hvx_llm_forward: graph capture failed: %s", hipGetErrorString(e));
            return -1;
        }
        if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
            hipGraphDestroy(g);
            return set_error("hvx_llm_forward: graph instantiate failed"), -1;
        }
        hipGraphDestroy(g);
        it = h->graphs.emplace(key, ex).first;
    }
    if (hipGraphLaunch(it->second, s) != hipSuccess) return set_error("hvx_llm_forward: graph launch failed"), -1;
    return 0;
}

static int decode_step_impl(hvx_llm* h, hipStream_t s, const hvx_decode_args& a) {
    if (forward_impl(h, s, a.n_seq, a.head_k, a.tok, a.ctrl, a.head_k, a.logp)) return -1;
    SampleArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.n_seq = a.n_seq; sa.head_k = a.head_k; sa.V = h->c.vocab; sa.Vs = h->c.speech_tokens;
    sa.logp = a.logp; sa.logp_ss = (long long)a.head_k * h->c.vocab; sa.logp_hs = h->c.vocab;
    sa.hist = a.hist; sa.hist_ss = a.win_cap; sa.hist_len = a.hist_len; sa.min_len = a.min_adj; sa.active = a.active;
    sa.top_k = a.top_k; sa.top_p = a.top_p; sa.win_size = a.win_size; sa.rep_thresh = a.rep_thresh;
    sa.noise = a.noise; sa.noise_ss = a.noise_seq_stride; sa.noise_len = a.noise_len; sa.noise_limit = (const long long*)a.noise_limit; sa.cursor = (long long*)a.cursor;
    sa.out_ids = a.ids; sa.max_trials = a.max_trials;
    if (launch_ras_sample(sa, s)) return -1;
    AdvanceArgs aa;
    memset(&aa, 0, sizeof(aa));
    aa.n_seq = a.n_seq; aa.head_k = a.head_k; aa.win_cap = a.win_cap; aa.max_out = a.max_out; aa.speech_tokens = h->c.speech_tokens;
    aa.ids = a.ids; aa.tok = a.tok; aa.ctrl = a.ctrl; aa.hist = a.hist; aa.hist_len = a.hist_len; aa.min_adj = a.min_adj; aa.active = a.active;
    aa.seq_state = a.seq_state; aa.out_tokens = a.out_tokens;
    return launch_decode_advance(aa, s);
}

int hvx_llm_decode_steps(hvx_llm* h, hvx_stream stream, const hvx_decode_args* a, int32_t n_steps) {
    if (!h || !h->ws || !a) return set_error("hvx_llm_decode_steps: handle not bound / null argument"), -1;
    if (a->n_seq < 1 || a->head_k < 1 || a->head_k > 8 || a->win_cap < 1 || a->max_out < 1 || !a->tok || !a->ctrl || !a->hist || !a->hist_len ||
        !a->min_adj || !a->active || !a->seq_state || !a->out_tokens || !a->ids || !a->logp || !a->noise || !a->cursor)
        return set_error("hvx_llm_decode_steps: bad argument block"), -1;
    hipStream_t s = (hipStream_t)stream;
    if (!h->use_graph || prof_enabled() || s == nullptr) {
        for (int i = 0; i < n_steps; ++i)
            if (decode_step_impl(h, s, *a)) return -1;
        return 0;
    }
    hipGraphExec_t ex = nullptr;
    for (auto& g : h->step_graphs)
        if (g.s == s && memcmp(&g.a, a, sizeof(*a)) == 0) ex = g.ex;
    if (!ex) {
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return set_error("hvx_llm_decode_steps: begin capture failed"), -1;
        const int rc = decode_step_impl(h, s, *a);
        const hipError_t e = hipStreamEndCapture(s, &g);
        if (rc != 0 || e != hipSuccess || !g) {
            if (g) hipGraphDestroy(g);
            if (rc == 0) set_error("hvx_llm_decode_steps: graph capture failed: %s", hipGetErrorString(e));
            return -1;
        }
        if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
            hipGraphDestroy(g);
            return set_error("hvx_llm_decode_steps: graph instantiate failed"), -1;
        }
        hipGraphDestroy(g);
        if (h->step_graphs.size() >= 8) {                     // callers re-bind control blocks per utterance batch: keep the cache small
            hipGraphExecDestroy(h->step_graphs.front().ex);
            h->step_graphs.erase(h->step_graphs.begin());
        }
        hvx_llm::StepGraph sg;
        memcpy(&sg.a, a, sizeof(*a));                         // (byte copy: padding is compared too, callers pass a zeroed block)
        sg.s = s;
        sg.ex = ex;
        h->step_graphs.push_back(sg);
    }
    for (int i = 0; i < n_steps; ++i)
        if (hipGraphLaunch(ex, s) != hipSuccess) return set_error("hvx_llm_decode_steps: graph launch failed"), -1;
    return 0;
}

int hvx_llm_decode_join(hvx_llm* h, hvx_stream stream, const hvx_decode_args* a, int32_t slot, int32_t first_tok, int32_t pos,
                        int32_t min_len, int32_t max_len) {
    if (!h || !h->ws || !a) return set_error("hvx_llm_decode_join: handle not bound / null argument"), -1;
    if (slot < 0 || slot >= a->n_seq || slot >= h->n_slots || pos < 0 || pos + 1 > h->max_ctx || !a->cursor)
        return set_error("hvx_llm_decode_join: slot %d / position %d outside the bound grid", slot, pos), -1;
    AdvanceArgs aa;
    memset(&aa, 0, sizeof(aa));
    aa.n_seq = a->n_seq; aa.head_k = a->head_k; aa.win_cap = a->win_cap; aa.max_out = a->max_out; aa.speech_tokens = h->c.speech_tokens;
    aa.ids = a->ids; aa.tok = a->tok; aa.ctrl = a->ctrl; aa.hist = a->hist; aa.hist_len = a->hist_len; aa.min_adj = a->min_adj; aa.active = a->active;
    aa.seq_state = a->seq_state; aa.out_tokens = a->out_tokens;
    return launch_decode_join(aa, (long long*)a->cursor, slot, first_tok, pos, min_len, max_len, (hipStream_t)stream);
}

int hvx_llm_use_graph(hvx_llm* h, int32_t enable) {
    if (!h) return set_error("hvx_llm_use_graph: null handle"), -1;
    h->use_graph = enable != 0;
    return 0;
}

int hvx_llm_set_head_mlp_fp8(hvx_llm* h, const void* codes, const float* scales) {
    if (!h) return set_error("hvx_llm_set_head_mlp_fp8: null handle"), -1;
    if (codes && !scales) return set_error("hvx_llm_set_head_mlp_fp8: codes without scales"), -1;
    if (codes && h->c.dtype != DT_BF16) return set_error("hvx_llm_set_head_mlp_fp8: the fp8 stream feeds the bf16 decode forms only"), -1;
    if (codes && (h->c.hidden != 896 || (h->c.mtp_inter & 15))) return set_error("hvx_llm_set_head_mlp_fp8: hidden %d is not the ring form's 28 k-steps", h->c.hidden), -1;
    h->head_mlp_codes = codes;
    h->head_mlp_scales = codes ? scales : nullptr;
    h->drop_graphs();                       // (captured launches hold the weight pointers)
    return 0;
}

static int forward_impl(hvx_llm* h, hipStream_t s, int32_t n_seq, int32_t kn, const int32_t* tok, const int32_t* ctrl, int32_t head_k,
                        float* logp) {
    const hvx_llm_config& c = h->c;
    const int R = n_seq * kn;
    if (n_seq < 1 || kn < 1 || n_seq > h->max_seq || R > h->max_rows) return set_error("hvx_llm_forward: grid %dx%d exceeds the bound workspace", n_seq, kn), -1;
    if (head_k > c.head_num) return set_error("hvx_llm_forward: head_k=%d > head_num=%d", head_k, c.head_num), -1;
    const int H = c.hidden, Q = c.q_heads * 64, G = c.q_heads / c.kv_heads;
    const int dt = c.dtype;
    const size_t es = dtype_size(dt);
    const int32_t* d_slot = ctrl;
    const int32_t* d_pos0 = ctrl + n_seq;
    const int32_t* d_nnew = ctrl + 2 * n_seq;
    const int32_t* d_kvlen = ctrl + 3 * n_seq;
    const int32_t* d_last = ctrl + 4 * n_seq;
    const void* const* w = h->w.data();
    const float* rope_cos = (const float*)w[0];
    const float* rope_sin = (const float*)w[1];

    // ---- embeddings: speech ids (>= 0) and text ids (<= -2) come from different tables -----------------------
    // the fused-norm GEMMs read the residual stream in the operand type: fp32 mode reads x itself, bf16 mode a bf16 copy (h->a) that
    // the embedding and the residual epilogues keep beside the fp32 stream (half the bytes per workgroup)
    void* const xa = dt == DT_F32 ? (void*)h->x : h->a;
    void* const xcopy = dt == DT_F32 ? nullptr : h->a;
    // Wide decode grids (33..128 rows, bf16, the backbone's own dimensions): the four GEMMs of a layer take the A-stationary / weight-ring form
    // (gemm_dec.hip), and every 16-bit activation matrix between them — the copy of x, the attention output, the MLP hidden rows — lives in
    // fragment order (hvx_device.h: frag_index), written that way by its producer.
    const int down_split = c.inter / 32 / 19;
    const bool dec = dt == DT_BF16 && dec_gemm_shape_ok(R, (c.q_heads + 2 * c.kv_heads) * 64, H, SK_QKV_ROPE, 1) && dec_gemm_shape_ok(R, H, Q, SK_RESID, 1) &&
                     dec_gemm_shape_ok(R, 2 * c.inter, H, SK_SWIGLU, 1) && down_split <= MAX_SPLIT && dec_gemm_shape_ok(R, H, c.inter, SK_PARTIAL, down_split);
    auto gemm = [&](const SkinnyArgs& g) {
        if (!dec) return launch_skinny(g, s);
        const int rc = launch_dec_gemm(g, s);
        return rc == 1 ? 0 : (rc == 0 ? (set_error("hvx_llm_forward: decode GEMM form refused a shape it had accepted"), -1) : -1);
    };
    if (launch_embed2(w[4], w[5], dt, tok, h->x, H, xcopy, R, H, s, dec ? 1 : 0)) return -1;
    // Five launches per layer: both RMSNorms ride inside the GEMM that consumes them (gain folded into the weights, 1/rms applied to
    // the accumulator; gemm_skinny.hip) and both residual adds in the epilogue of the GEMM that produces them (x += ..., one writer
    // per element).
    const bool use_split = (G * kn <= h->att_rows_pad) && (kn <= 8) && h->att_splits > 1;
    // more than two 16-row tiles (prefill; decode of >= 17 sequences x 2 heads): the 4-column form would re-read its activation rows per
    // tile, see resid_wide
    const bool wide = R > 32;
    // a grouped prefill (n_seq prefixes of kn > 256 rows): the residual projections choose their K split as for ONE prefix, whatever the group size
    const int m_split = (n_seq > 1 && kn > 256) ? kn : 0;
    for (int l = 0; l < c.layers; ++l) {
        const void* const* lw = w + 6 + 9 * l;
        // 1. QKV + bias + RoPE + KV append
        SkinnyArgs g;
        memset(&g, 0, sizeof(g));
        g.dtype = dt; g.M = R; g.N = (c.q_heads + 2 * c.kv_heads) * 64; g.K = H; g.A = xa; g.lda = H; g.W = lw[1]; g.split_k = 1; g.nz = 1;
        g.a_norm = 1; g.norm_eps = c.rms_eps;               // input_layernorm gain is folded into lw[1] (llm.py)
        g.epi = SK_QKV_ROPE; g.bias = (const float*)lw[2];
        g.kn = kn; g.q_heads = c.q_heads; g.kv_heads = c.kv_heads; g.slot = d_slot; g.pos0 = d_pos0; g.n_new = d_nnew;
        g.rope_cos = rope_cos; g.rope_sin = rope_sin; g.qbuf = h->qbuf;
        const size_t layer_kv = (size_t)c.kv_heads * h->max_ctx * 64 * es;          // per (slot, layer)
        // cache layout [layer][slot][kv_head][...]: the slot stride is what the kernels index with
        g.kcache = (char*)h->kcache + (size_t)l * h->n_slots * layer_kv;
        g.vTcache = (char*)h->vTcache + (size_t)l * h->n_slots * layer_kv;
        g.max_ctx = h->max_ctx; g.kv_frag = 1;
        g.a_frag = dec;
        if (gemm(g)) return -1;
        // 2. attention, GQA-packed: rows = G query heads x kn new positions per KV head
        AttnArgs at;
        memset(&at, 0, sizeof(at));
        at.dtype = dt; at.batch = n_seq; at.heads = c.kv_heads; at.n_rows = G * kn; at.kn = kn;
        at.q = h->qbuf; at.q_bs = (long long)kn * Q; at.q_hs = (long long)G * 64; at.q_hi = 64; at.q_lo = Q;
        at.k = g.kcache; at.k_bs = (long long)c.kv_heads * h->max_ctx * 64; at.k_hs = (long long)h->max_ctx * 64;
        at.vT = g.vTcache; at.v_bs = at.k_bs; at.v_hs = (long long)64 * h->max_ctx; at.v_ld = h->max_ctx;
        at.kv_frag = 1;
        at.kv_slot = d_slot; at.kv_len = d_kvlen; at.causal = 1; at.pos0 = d_pos0; at.n_valid_lo = d_nnew;
        at.scale = 0.125f;
        at.out = h->attn; at.o_bs = at.q_bs; at.o_hs = at.q_hs; at.o_hi = 64; at.o_lo = Q; at.o_frag_kt = dec ? Q / 32 : 0;
        if (use_split) {
            // keys per split: the partial buffers are carved for h->att_chunk (256); a wide grid (>= 32 sequences) takes splits of twice that — with
            // the fragment-order cache two 64-key trips per wave cost less than twice the workgroups (64 sequences: 1.346 -> 1.335 ms per step at
            // context 1536, 1.517 -> 1.473 at 2560, and -2 % beside the acoustic stage); narrow grids lose with it (8 sequences: 0.977 -> 1.009 ms)
            // bf16 only: the fp32 parity mode keeps ONE summation order (256-key splits) at every grid width, like the heads' down projection below
            const int chunk = (n_seq >= 32 && dt == DT_BF16 && !att_chunk_from_env()) ? 2 * h->att_chunk : h->att_chunk;
            // ... and eight waves per split there (bf16, one query tile): one 64-key trip per wave, every load of the split in flight at once
            const int att_waves = (int)opt(OPT_ATT_WAVES);
            const int nsub = (n_seq >= 32 && dt == DT_BF16 && G * kn <= 16 && att_waves == 8 && chunk % 256 == 0) ? 8 : 4;
            at.n_splits = (h->max_ctx + chunk - 1) / chunk; at.split_chunk = chunk; at.n_sub = nsub; at.sub_chunk = chunk / nsub; at.part_o = h->att_o; at.part_ml = h->att_ml; at.n_rows_pad = h->att_rows_pad;
        } else {
            at.n_splits = 1;
        }
        // option dec_fuse_rows (default 0): no combine launch — the o_proj builds its activation fragments from the partials (gemm_skinny.hip:
        // combined_pair).  Bit-identical; measured slower (826 -> 925 us per step for one request, 975 -> 1186 for 8): the launch it removes costs less
        // than the dependent control -> partials -> exp chain it puts in front of the o_proj's MFMAs (the plain o_proj loads its rows beside its weights)
        const bool fuse = use_split && !dec && R <= (int)opt(OPT_DEC_FUSE_ROWS) && Q <= 1024;
        at.skip_combine = fuse;
        if (launch_attention(at, s)) return -1;
        // 3. x += o_proj(attn)
        memset(&g, 0, sizeof(g));
        g.dtype = dt; g.M = R; g.N = H; g.K = Q; g.A = h->attn; g.lda = Q; g.W = lw[3]; g.nz = 1; g.split_k = 1;
        g.epi = SK_RESID; g.out = h->x; g.out_f32 = 1; g.ldo = H; g.out2 = xcopy; g.ldo2 = H; g.w_narrow = 1;
        if (fuse) {
            g.att_o = at.part_o; g.att_ml = at.part_ml; g.att_kvlen = d_kvlen; g.att_splits = at.n_splits; g.att_chunk = at.split_chunk; g.att_rows_pad = at.n_rows_pad;
            g.kn = kn; g.q_heads = c.q_heads; g.kv_heads = c.kv_heads; g.pos0 = d_pos0; g.n_new = d_nnew;
        }
        if (dec) {
            g.W = lw[7]; g.w_narrow = 0; g.a_frag = 1; g.out_frag = 1;                          // (fragment-packed copy of the weights)
            if (gemm(g)) return -1;
        } else if (R > 256 ? resid_wide(h, g, lw[7], xcopy, s, m_split) : launch_skinny(g, s)) return -1;      // K = q*64 is short: the 4-column form holds up to 16 row tiles
        // 4. hmlp = SwiGLU(RMSNorm(x) * ln2)
        memset(&g, 0, sizeof(g));
        g.dtype = dt; g.M = R; g.N = 2 * c.inter; g.K = H; g.A = xa; g.lda = H; g.W = lw[5]; g.split_k = 1; g.nz = 1;
        g.a_norm = 1; g.norm_eps = c.rms_eps;               // post_attention_layernorm gain is folded into lw[5]
        g.epi = SK_SWIGLU; g.out = h->hmlp; g.ldo = c.inter; g.a_frag = dec; g.out_frag = dec;
        if (gemm(g)) return -1;
        // 5. x += down(hmlp)
        memset(&g, 0, sizeof(g));
        g.dtype = dt; g.M = R; g.N = H; g.K = c.inter; g.A = h->hmlp; g.lda = c.inter; g.W = lw[6]; g.nz = 1; g.split_k = 1;
        g.epi = SK_RESID; g.out = h->x; g.out_f32 = 1; g.ldo = H; g.out2 = xcopy; g.ldo2 = H; g.w_narrow = 1;
        if (dec) {
            // split-K partials (8 K slices of 19 k-steps) + the fixed-order reduce, which also writes the fragment-order copy of x
            g.W = lw[8]; g.w_narrow = 0; g.a_frag = 1; g.split_k = down_split; g.epi = SK_PARTIAL; g.part = h->part; g.out = nullptr; g.out2 = nullptr;
            if (gemm(g)) return -1;
            ReduceNormArgs r;
            memset(&r, 0, sizeof(r));
            r.x = h->x; r.ldx = H; r.part = h->part; r.split_k = down_split; r.part_stride = (long long)R * H; r.do_norm = 0;
            r.y = xcopy; r.ldy = H; r.dtype = dt; r.M = R; r.H = H; r.rows_per_z = R; r.y_frag = 1;
            if (launch_reduce_rmsnorm(r, s)) return -1;
        } else if (wide ? resid_wide(h, g, lw[8], xcopy, s, m_split) : launch_skinny(g, s)) return -1;
    }
    if (head_k <= 0) return 0;

    // ---- last rows -> final RMSNorm (hidden_states[-1], llm_multi_head_v3.py:248-260, 886) ---------------------
    const int S = n_seq;
    const void* const* mw = w + 6 + 9 * c.layers;
    const int A = c.mtp_attn_dim, I = c.mtp_inter, K = head_k;
    // ---- K MTP heads, batched over blockIdx.z; their residual copies and input norms come out of the same launch as the final norm ---
    if (launch_heads_prologue(h->x, H, d_last, (const float*)w[2], c.rms_eps, (const float*)mw[0], c.mtp_rms_eps, K, S, H, h->ylast, h->hx, h->ha, dt, s))
        return -1;
    ReduceNormArgs hn;
    memset(&hn, 0, sizeof(hn));
    hn.x = h->hx; hn.ldx = H; hn.gain_zs = H; hn.eps = c.mtp_rms_eps; hn.do_norm = 1; hn.y = h->ha; hn.ldy = H;
    hn.dtype = dt; hn.M = K * S; hn.H = H; hn.rows_per_z = S;
    SkinnyArgs g;
    // v = Wv n1 + bv
    memset(&g, 0, sizeof(g));
    g.dtype = dt; g.M = S; g.N = A; g.K = H; g.A = h->ha; g.lda = H; g.a_zs = (long long)S * H; g.W = mw[1]; g.w_zs = (long long)A * H;
    g.split_k = 1; g.nz = K; g.epi = SK_STORE; g.bias = (const float*)mw[2]; g.bias_zs = A; g.out = h->hv; g.out_f32 = 0; g.ldo = A; g.out_zs = (long long)S * A;
    if (launch_skinny(g, s)) return -1;
    // h1 = y + Wo v
    memset(&g, 0, sizeof(g));
    g.dtype = dt; g.M = S; g.N = H; g.K = A; g.A = h->hv; g.lda = A; g.a_zs = (long long)S * A; g.W = mw[3]; g.w_zs = (long long)H * A;
    g.split_k = pick_split(H, A, K); g.nz = K; g.epi = SK_PARTIAL; g.part = h->part; g.part_zs = (long long)g.split_k * S * H;
    if (launch_skinny(g, s)) return -1;
    hn.part = h->part; hn.split_k = g.split_k; hn.part_stride = (long long)S * H; hn.part_zs = g.part_zs; hn.gain = (const float*)mw[4];
    // wide bf16 grids (33..256 sequences): the heads' gate / up projection takes the weight-ring form (gemm_dec.hip), every head its own fragment-order
    // matrix of S16 rows written by this reduce
    const int S16 = (S + 15) / 16 * 16;
    const int heads_dec = (int)opt(OPT_DEC_HEADS);      // A / B option dec_heads: bit 0 = MLP, bit 1 = output projection
    const bool dec_mlp = (heads_dec & 1) && dt == DT_BF16 && dec_gemm_shape_ok(S, 2 * I, H, SK_SWIGLU, 1);
    hn.y_frag = dec_mlp; hn.y_frag_zrows = dec_mlp ? S16 : 0;
    if (launch_reduce_rmsnorm(hn, s)) return -1;
    hn.y_frag = 0; hn.y_frag_zrows = 0;
    // SwiGLU MLP
    memset(&g, 0, sizeof(g));
    g.dtype = dt; g.M = S; g.N = 2 * I; g.K = H; g.A = h->ha; g.lda = H; g.a_zs = (long long)S * H; g.W = mw[5]; g.w_zs = (long long)2 * I * H;
    g.split_k = 1; g.nz = K; g.epi = SK_SWIGLU; g.out = h->hm; g.ldo = I; g.out_zs = (long long)S * I;
    if (dec_mlp) {
        g.a_frag = 1; g.a_zs = (long long)S16 * H;
        if (h->head_mlp_codes) {                              // the same weights as e4m3 codes x power-of-two column scales: half the bytes, the same products
            g.W = h->head_mlp_codes; g.w_fp8 = 1; g.w_scale = h->head_mlp_scales; g.w_scale_zs = 2 * I;
        }
        const int rc = launch_dec_gemm(g, s);
        if (rc < 0) return -1;
        if (rc == 0) return set_error("hvx_llm_forward: head MLP shape left the decode form"), -1;
    } else if (launch_skinny(g, s)) return -1;
    memset(&g, 0, sizeof(g));
    g.dtype = dt; g.M = S; g.N = H; g.K = I; g.A = h->hm; g.lda = I; g.a_zs = (long long)S * I; g.W = mw[6]; g.w_zs = (long long)H * I;
    g.split_k = pick_split(H, I, K); g.nz = K; g.epi = SK_PARTIAL; g.part = h->part; g.part_zs = (long long)g.split_k * S * H;
    {   // wide grids (the mid-M form, 33..128 rows): a workgroup walks its K slice serially, and pick_split's 3 slices leave 84 workgroups to pull
        // 39 MB per head (57 us for two heads); 8 slices of ~43 K-tiles: 30 us (decode step 1.314 -> 1.287 ms at 64 sequences; 6 / 12 / 16 slices
        // 1.293 / 1.292 / 1.292).  bf16 only: the fp32 parity mode keeps its summation order.
        const int force = (int)opt(OPT_HEAD_DOWN_SPLIT);        // (lab option)
        int want = dt != DT_BF16 ? 0 : (force > 0 ? force : (I / 64 / 12 >= 8 ? 8 : 0));          // (the lab override too: never in the fp32 mode)
        if (want > MAX_SPLIT) want = MAX_SPLIT;
        if (want > g.split_k && S > 32 && S <= 128) { g.split_k = want; g.part_zs = (long long)g.split_k * S * H; }
    }
    if (launch_skinny(g, s)) return -1;
    hn.part = h->part; hn.split_k = g.split_k; hn.part_stride = (long long)S * H; hn.part_zs = g.part_zs; hn.gain = nullptr; hn.do_norm = 0;   // plain cast
    // wide bf16 grids: the K heads' rows are one stacked fragment-order matrix for the weight-ring form of the output projection (gemm_dec.hip)
    const bool dec_out = (heads_dec & 2) && dt == DT_BF16 && dec_gemm_shape_ok(K * S, c.vocab_pad, H, SK_STORE, 1);
    hn.y_frag = dec_out;
    if (launch_reduce_rmsnorm(hn, s)) return -1;
    // logits = llm_decoder(h) (shared weights) ; log_softmax
    memset(&g, 0, sizeof(g));
    g.dtype = dt; g.M = S; g.N = c.vocab_pad; g.K = H; g.A = h->ha; g.lda = H; g.a_zs = (long long)S * H; g.W = w[3]; g.w_zs = 0;
    // straight into the caller's [S][K][vocab] buffer when there is one (the 16-row padding of the decoder is not stored)
    float* dst = logp ? logp : h->logits;
    const int ld = logp ? c.vocab : c.vocab_pad;
    g.split_k = 1; g.nz = K; g.epi = SK_STORE; g.out = dst; g.out_f32 = 1; g.ldo = K * ld; g.out_zs = ld; g.n_valid = c.vocab;
    if (dec_out) {
        g.a_frag = 1;
        const int rc = launch_dec_gemm(g, s);
        if (rc < 0) return -1;
        if (rc == 0) return set_error("hvx_llm_forward: output projection shape left the decode form"), -1;
    } else if (launch_skinny(g, s)) return -1;
    if (launch_log_softmax(dst, ld, S * K, c.vocab, s)) return -1;
    return 0;
}

int hvx_llm_last_hidden(hvx_llm* h, hvx_stream s, int32_t n_seq, float* out) {
    if (!h || !h->ylast) return set_error("hvx_llm_last_hidden: handle not bound"), -1;
    if (hipMemcpyAsync(out, h->ylast, (size_t)n_seq * h->c.hidden * 4, hipMemcpyDeviceToDevice, (hipStream_t)s) != hipSuccess)
        return set_error("hvx_llm_last_hidden: memcpy failed"), -1;
    return 0;
}

}  // extern "C"
