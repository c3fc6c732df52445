// hvx_api.hip — C-ABI glue: error text, device probe, operator-level entry points (include/hvx.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "hvx.h"
#include "hvx_kernels.h"
#include "hvx_options.h"

namespace hvx {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- sampled kernel timing ------------------------------------------------------------------------------------------------
// The one piece of process-wide mutable state in the library (a measurement aid; no result depends on it): `on` is an atomic that launches
// read without the lock (off = the whole cost of the facility), everything else is touched under `mu` only — launches from several host
// threads (two acoustic chains, the decode engine) and hvx_prof_enable / hvx_prof_read may interleave freely.
struct ProfSlot { int kind; double work; hipEvent_t e0, e1; };
static struct {
    std::atomic<bool> on{false};
    std::mutex mu;
    int period = 1;
    long long launched[PK_COUNT] = {};
    double launched_work[PK_COUNT] = {};
    std::vector<ProfSlot> slots;
} g_prof;
constexpr size_t PROF_MAX_SLOTS = 1 << 15;

int prof_begin(int kind, double work, hipStream_t s) {
    if (!g_prof.on.load(std::memory_order_relaxed)) return -1;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (!g_prof.on.load(std::memory_order_relaxed)) return -1;
    const long long n = g_prof.launched[kind]++;
    g_prof.launched_work[kind] += work;
    if (n % g_prof.period != 0 || g_prof.slots.size() >= PROF_MAX_SLOTS) return -1;
    ProfSlot p;
    p.kind = kind; p.work = work;
    if (hipEventCreate(&p.e0) != hipSuccess || hipEventCreate(&p.e1) != hipSuccess) return -1;
    hipEventRecord(p.e0, s);
    g_prof.slots.push_back(p);
    return (int)g_prof.slots.size() - 1;
}
bool prof_enabled() { return g_prof.on.load(std::memory_order_relaxed); }
void prof_end(int slot, hipStream_t s) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if ((size_t)slot < g_prof.slots.size()) hipEventRecord(g_prof.slots[slot].e1, s);      // (a slot of a session that was reset meanwhile is dropped)
}

// ---- run-time options (hvx_options.h) ---------------------------------------------------------------------------------------
const OptDef g_opt_defs[OPT_COUNT] = {
    {"att_chunk", 0, 0}, {"att_waves", 8, 0}, {"gemm_big_gw", 4, 0}, {"gemm_big_min_tiles", 128, 0}, {"dec_gemm", 1, 0}, {"dec_heads", 3, 0},
    {"conv_resident", 1, 0}, {"conv64_resident", 1, 0}, {"x3p8", 1, 0}, {"attn_dit_form", 0, 0}, {"dec_fuse_rows", 0, 0},
    {"gemm_big_mfma", 16, 1}, {"head_down_split", 0, 1}, {"dec_gpw_qkv", 1, 1}, {"dec_gpw_res", 1, 1}, {"dec_gpw_mlp", 3, 1}, {"dec_gpw_down", 2, 1}, {"dec_gpw_out", 3, 1}, {"dec_gpw_hmlp", 11, 1},
    {"attn_lab", 0, 1}, {"attn_nw", 4, 1},
};
static std::atomic<long long> g_opt_val[OPT_COUNT];
static std::atomic<bool> g_opt_init{false};
static std::mutex g_opt_mu;
static void opt_init() {
    if (g_opt_init.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (g_opt_init.load(std::memory_order_relaxed)) return;
    for (int i = 0; i < OPT_COUNT; ++i) g_opt_val[i].store(g_opt_defs[i].dflt, std::memory_order_relaxed);
    g_opt_init.store(true, std::memory_order_release);
}
long long opt(OptId id) {
    opt_init();
    return g_opt_val[id].load(std::memory_order_relaxed);
}
static int opt_find(const char* key) {
    if (key)
        for (int i = 0; i < OPT_COUNT; ++i)
            if (strcmp(key, g_opt_defs[i].name) == 0) return i;
    return -1;
}
}  // namespace hvx

using namespace hvx;

extern "C" {

int hvx_abi_version(void) { return HVX_ABI_VERSION; }
const char* hvx_last_error(void) { return g_err; }

const char* hvx_build_flags(void) {
#ifdef HVX_BUILD_FLAGS
    return HVX_BUILD_FLAGS;
#else
    return "";
#endif
}
int hvx_is_lab_build(void) {
#ifdef HVX_LAB
    return 1;
#else
    return 0;
#endif
}
int hvx_set_option(const char* key, int64_t value) {
    const int i = opt_find(key);
    if (i < 0) return set_error("hvx_set_option: unknown option '%s'", key ? key : "(null)"), -1;
#ifndef HVX_LAB
    if (g_opt_defs[i].lab) return set_error("hvx_set_option: '%s' is a lab option; this library was built without -DHVX_LAB", key), -1;
#endif
    if (i == OPT_ATT_CHUNK && value != 0 && (value < 128 || value % 128)) return set_error("hvx_set_option: att_chunk must be 0 or a multiple of 128, got %lld", (long long)value), -1;
    if (i == OPT_GEMM_BIG_MFMA && value != 16 && value != 32) return set_error("hvx_set_option: gemm_big_mfma must be 16 or 32, got %lld", (long long)value), -1;
    if (i == OPT_DEC_FUSE_ROWS && (value < 0 || value > 256)) return set_error("hvx_set_option: dec_fuse_rows must be 0..256, got %lld", (long long)value), -1;
    if (i == OPT_ATT_WAVES && value != 4 && value != 8) return set_error("hvx_set_option: att_waves must be 4 or 8, got %lld", (long long)value), -1;
    #ifdef HVX_LAB
    const bool form_ok = value == 0 || value == 16 || value == 17 || value == 32 || value == 48;
#else
    const bool form_ok = value == 0 || value == 16 || value == 17 || value == 32;
#endif
    if (i == OPT_ATTN_DIT_FORM && !form_ok) return set_error("hvx_set_option: attn_dit_form must be 0, 16, 17 or 32 (48: lab builds only), got %lld", (long long)value), -1;
    opt_init();
    g_opt_val[i].store(value, std::memory_order_relaxed);
    return 0;
}
int hvx_get_option(const char* key, int64_t* value) {
    const int i = opt_find(key);
    if (i < 0 || !value) return set_error("hvx_get_option: unknown option '%s'", key ? key : "(null)"), -1;
    *value = opt((OptId)i);
    return 0;
}
const char* hvx_option_name(int32_t index, int32_t* lab_only, int64_t* dflt) {
    if (index < 0 || index >= OPT_COUNT) return nullptr;
    if (lab_only) *lab_only = g_opt_defs[index].lab;
    if (dflt) *dflt = g_opt_defs[index].dflt;
    return g_opt_defs[index].name;
}

int hvx_device_ok(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("hipGetDeviceCount: %s (count %d)", hipGetErrorString(e), n);
        (void)hipGetLastError();
        return 0;
    }
    int dev = 0;
    hipDeviceProp_t p;
    if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&p, dev)) != hipSuccess) {
        set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
        return 0;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        set_error("libhvx is built for gfx950 only, found %s", p.gcnArchName);
        return 0;
    }
    return p.multiProcessorCount;
}

int hvx_stream_create_cu_range(int32_t first_cu, int32_t n_cus, hvx_stream* out) {
    if (!out || first_cu < 0 || n_cus <= 0) return set_error("hvx_stream_create_cu_range: bad arguments"), -1;
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return set_error("hvx_stream_create_cu_range: no device"), -1;
    if (first_cu + n_cus > p.multiProcessorCount)
        return set_error("hvx_stream_create_cu_range: CUs [%d, %d) outside the device's %d", first_cu, first_cu + n_cus, p.multiProcessorCount), -1;
    std::vector<uint32_t> mask((p.multiProcessorCount + 31) / 32, 0u);
    for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) return set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e)), -1;
    *out = (hvx_stream)s;
    return 0;
}
int hvx_stream_destroy(hvx_stream s) {
    return hipStreamDestroy((hipStream_t)s) == hipSuccess ? 0 : (set_error("hipStreamDestroy failed"), -1);
}

int hvx_prof_enable(int32_t period) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (auto& p : g_prof.slots) { hipEventDestroy(p.e0); hipEventDestroy(p.e1); }
    g_prof.slots.clear();
    for (int k = 0; k < PK_COUNT; ++k) { g_prof.launched[k] = 0; g_prof.launched_work[k] = 0; }
    g_prof.on = period > 0;
    g_prof.period = period > 0 ? period : 1;
    return 0;
}

int hvx_prof_read(int32_t kind, double* sampled_ms, double* sampled_work, int64_t* n_sampled, int64_t* n_launched, double* launched_work) {
    if (kind < 0 || kind >= PK_COUNT) return set_error("hvx_prof_read: bad kind"), -1;
    double ms = 0, work = 0;
    long long n = 0;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (auto& p : g_prof.slots) {
        if (p.kind != kind) continue;
        float t = 0;
        if (hipEventSynchronize(p.e1) != hipSuccess || hipEventElapsedTime(&t, p.e0, p.e1) != hipSuccess) continue;
        ms += t; work += p.work; ++n;
    }
    if (sampled_ms) *sampled_ms = ms;
    if (sampled_work) *sampled_work = work;
    if (n_sampled) *n_sampled = n;
    if (n_launched) *n_launched = g_prof.launched[kind];
    if (launched_work) *launched_work = g_prof.launched_work[kind];
    return 0;
}

int hvx_ras_sample(const hvx_sample_args* a, hvx_stream s) {
    if (!a) return set_error("hvx_ras_sample: null args"), -1;
    SampleArgs k;
    k.n_seq = a->n_seq; k.head_k = a->head_k; k.V = a->vocab; k.Vs = a->speech_tokens;
    k.logp = a->logp; k.logp_ss = a->logp_seq_stride; k.logp_hs = a->logp_head_stride;
    k.hist = a->hist; k.hist_ss = a->hist_seq_stride; k.hist_len = a->hist_len;
    k.min_len = a->min_len; k.active = a->active;
    k.top_k = a->top_k; k.top_p = a->top_p; k.win_size = a->win_size; k.rep_thresh = a->rep_thresh;
    k.noise = a->noise; k.noise_ss = a->noise_seq_stride; k.noise_len = a->noise_len; k.noise_limit = (const long long*)a->noise_limit;
    k.cursor = (long long*)a->cursor; k.out_ids = a->out_ids; k.max_trials = a->max_trials;
    return launch_ras_sample(k, (hipStream_t)s);
}

int hvx_op_gemm(const hvx_gemm_args* a, hvx_stream s) {
    if (!a) return set_error("hvx_op_gemm: null args"), -1;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.dtype = a->dtype; g.M = a->M; g.N = a->N; g.K = a->K; g.batch = a->batch; g.groups = a->groups;
    g.A = a->A; g.a_bs = a->a_bs; g.lda = a->lda; g.a_gs = a->a_gs; g.rows_in = a->rows_in;
    g.cin_pad = a->cin_pad; g.conv_stride = a->conv_stride; g.conv_dil = a->conv_dil; g.pad_left = a->pad_left; g.up = a->up;
    g.W = a->W; g.w_gs = a->w_gs; g.epi = EPI_GENERIC;
    g.bias = a->bias; g.act = a->act; g.act_param = a->act_param; g.act_alpha = a->act_alpha;
    g.gate = a->gate; g.gate_bs = a->gate_bs;
    g.res = a->res; g.res_bs = a->res_bs; g.ldres = a->ldres; g.res_row_off = a->res_row_off;
    g.scale = a->scale;
    g.out = a->out; g.out_f32 = a->out_f32; g.out_bs = a->out_bs; g.ldo = a->ldo; g.out_row_off = a->out_row_off; g.out_cols = a->out_cols;
    g.out2 = a->out2; g.act2 = a->act2; g.act2_param = a->act2_param; g.act2_alpha = a->act2_alpha; g.out2_bs = a->out2_bs;
    g.ldo2 = a->ldo2; g.out2_row_off = a->out2_row_off; g.out2_cols = a->out2_cols; g.x3 = a->x3;
    g.res_f16 = a->res_f16; g.out_f16 = a->out_f16;
    if (g.out_f16 && g.out_f32) return set_error("hvx_op_gemm: out_f16 and out_f32 are exclusive"), -1;
    return launch_gemm(g, (hipStream_t)s);
}

int hvx_op_attention(const hvx_attn_args* a, hvx_stream s) {
    if (!a) return set_error("hvx_op_attention: null args"), -1;
    AttnArgs k;
    memset(&k, 0, sizeof(k));
    const long long D = (long long)a->heads * 64;
    k.dtype = a->dtype; k.batch = a->batch; k.heads = a->heads; k.n_rows = a->t; k.kn = a->t;
    k.q = a->q; k.q_bs = (long long)a->heads * a->t_pad * 64; k.q_hs = (long long)a->t_pad * 64; k.q_hi = 0; k.q_lo = 64;
    k.k = a->k; k.k_bs = k.q_bs; k.k_hs = k.q_hs;
    k.vT = a->vT; k.v_bs = k.q_bs; k.v_hs = (long long)64 * a->t_pad; k.v_ld = a->t_pad;
    k.kv_len = a->kv_len; k.kv_len_const = a->t; k.causal = a->causal; k.chunk = a->chunk > 0 ? a->chunk : 0; k.scale = a->scale; k.q_log2 = a->q_log2;
    k.out = a->out; k.o_bs = (long long)a->t * D; k.o_hs = 64; k.o_hi = 0; k.o_lo = D;
    k.n_splits = a->n_splits > 1 ? a->n_splits : 1; k.split_chunk = a->split_chunk; k.part_o = a->part_o; k.part_ml = a->part_ml;
    k.n_rows_pad = (a->t + 31) & ~31;
    return launch_attention(k, (hipStream_t)s);
}

int hvx_op_resample_linear(const float* x, int32_t rows, int32_t t_in, float* y, int32_t t_out, hvx_stream s) {
    if (!x || !y) return set_error("hvx_op_resample_linear: null argument"), -1;
    return launch_resample_linear(x, rows, t_in, y, t_out, (hipStream_t)s);
}

int hvx_op_skinny_gemm(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* Wpacked, const float* bias,
                       int32_t split_k, float* part_ws, float* out_f32, int32_t ldo, hvx_stream s) {
    SkinnyArgs k;
    memset(&k, 0, sizeof(k));
    k.dtype = dtype; k.M = M; k.N = N; k.K = K; k.A = A; k.lda = lda; k.W = Wpacked; k.nz = 1;
    if (split_k <= 1) {
        k.split_k = 1; k.epi = SK_STORE; k.bias = bias; k.out = out_f32; k.out_f32 = 1; k.ldo = ldo;
        return launch_skinny(k, (hipStream_t)s);
    }
    // split-K: partials + deterministic reduce into a zero-initialised output (uses the same kernels as the LLM step)
    k.split_k = split_k; k.epi = SK_PARTIAL; k.part = part_ws;
    if (launch_skinny(k, (hipStream_t)s)) return -1;
    if (hipMemset2DAsync(out_f32, (size_t)ldo * 4, 0, (size_t)N * 4, M, (hipStream_t)s) != hipSuccess) return set_error("memset failed"), -1;
    ReduceNormArgs r;
    memset(&r, 0, sizeof(r));
    r.x = out_f32; r.ldx = ldo; r.part = part_ws; r.split_k = split_k; r.part_stride = (long long)M * N; r.bias = bias;
    r.M = M; r.H = N; r.rows_per_z = M; r.dtype = DT_F32;
    return launch_reduce_rmsnorm(r, (hipStream_t)s);
}

}  // extern "C"
