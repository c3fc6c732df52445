// hvx_hift.hip — causal HiFT vocoder: F0 predictor, NSF source, decode (include/hvx.h: hvx_hift_*).
//
// Restates (file:line under server/model_utils/cosyvoice/):
//   hifigan/f0_predictor.py:95-103      CausalConvRNNF0Predictor.forward
//   hifigan/generator.py:233-317,358-375  SineGen2 / SourceModuleHnNSF (causal, eval)
//   hifigan/generator.py:672-711        CausalHiFTGenerator.decode        :110-117 ResBlock.forward
//   transformer/convolution.py:150-258  CausalConv1d / DownSample / Upsample      transformer/activation.py:73-84 Snake
// fp32 end to end (the reference never casts hift, infer_speech_model.py:104).  Every Conv1d is one implicit-GEMM
// launch on the fp32 MFMA path with its neighbours fused into the epilogue: bias, Snake / LeakyReLU / ELU of the
// *next* layer's input, the ResBlock residual, the three-ResBlock mean and the source-branch add.  Activations are
// time-major [L][C] (C padded to 32) so a conv tap is a row shift.
#include <string.h>

#include <vector>

#include "hvx.h"
#include "hvx_device.h"
#include <stdlib.h>

#include "hvx_kernels.h"

using namespace hvx;

struct hvx_hift {
    hvx_hift_config c;
    std::vector<const void*> w;
    std::vector<const void*> wp;            // per weight: its (hi, lo) bf16 plane pair [2][Cout][taps * Cin_pad], or null (hvx_hift_set_weight_planes)
    int up_total = 0;
};

namespace {

#define HVX_CHECK(x) do { if (x) return -1; } while (0)
#define HIP_OK(x) do { if ((x) != hipSuccess) return set_error("hip call failed: %s", #x), -1; } while (0)

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
int pad32(int c) { return (c + 31) & ~31; }

struct Carve {
    char* base;
    size_t off = 0;
    explicit Carve(char* b) : base(b) {}
    float* take(size_t floats) {
        float* p = reinterpret_cast<float*>(base ? base + off : nullptr);
        off += align_up(floats * 4);
        return p;
    }
};

struct Bufs {
    float *melT, *fa, *fb, *phase, *spec, *post;
    float* pool[9];
    size_t pool_floats;
};

size_t carve(const hvx_hift* h, char* base, int T, Bufs& b) {
    const hvx_hift_config& c = h->c;
    Carve cv(base);
    b.melT = cv.take((size_t)T * pad32(c.mel));
    b.fa = cv.take((size_t)T * pad32(c.f0_channels));
    b.fb = cv.take((size_t)T * pad32(c.f0_channels));
    b.phase = cv.take((size_t)T * (c.nb_harmonics + 1));
    const long long Ls = (long long)T * h->up_total;
    const long long frames = Ls / c.hop + 1;
    b.spec = cv.take((size_t)frames * 32);
    b.post = cv.take((size_t)frames * 32);
    size_t mx = (size_t)T * pad32(c.base_channels);
    long long L = T;
    for (int i = 0; i < c.n_up; ++i) {
        L *= c.up_rates[i];
        const size_t need = (size_t)(L + 1) * pad32(c.base_channels >> (i + 1));
        if (need > mx) mx = need;
    }
    b.pool_floats = mx;
    for (int i = 0; i < 9; ++i) b.pool[i] = cv.take(mx);
    return cv.off;
}

GemmArgs conv(int M, int N, int taps, int cin_pad, const float* A, int lda, int rows_in, const float* W, const float* bias) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.dtype = DT_F32; g.M = M; g.N = N; g.K = taps * cin_pad; g.batch = 1; g.groups = 1;
    g.A = A; g.lda = lda; g.rows_in = rows_in; g.cin_pad = cin_pad; g.conv_stride = 1; g.conv_dil = 1; g.pad_left = 0; g.up = 1;
    g.W = W; g.epi = EPI_GENERIC; g.bias = bias; g.scale = 1.0f;
    return g;
}

// decode-path convolutions: fp32 operands as bf16 pairs on the bf16 matrix cores (gemm_x3.hip, ~1e-6 relative); hvx_hift_config.exact_fp32 keeps the
// exact fp32 MFMA form (x3 = false).  The F0 predictor (hvx_hift_f0) always stays exact: its output is integrated into the harmonic phase.
GemmArgs conv3(bool x3, int M, int N, int taps, int cin_pad, const float* A, int lda, int rows_in, const float* W, const float* bias) {
    GemmArgs g = conv(M, N, taps, cin_pad, A, lda, rows_in, W, bias);
    g.x3 = x3 ? 1 : 0;
    return g;
}

struct WCursor {
    const void* const* w;
    int n, i = 0;
    const void* const* wp = nullptr;        // plane pairs, parallel to w (or null)
    bool x3 = true;                         // split-bf16 convolutions (false: hvx_hift_config.exact_fp32)
    const float* next() { return (const float*)(i < n ? w[i++] : (i++, nullptr)); }
    const void* planes_of_last() const { return (wp && i >= 1 && i <= n) ? wp[i - 1] : nullptr; }
};

// a convolution whose input activation and whose weight are (hi, lo) bf16 plane pairs (gemm_x3.hip: gemm_x3p_kernel); a_plane = elements
// between the planes of the activation, the weight planes are Cout * K apart
GemmArgs convp(int M, int N, int taps, int cin_pad, const void* A, long long a_plane, int lda, int rows_in, const void* Wp, const float* bias) {
    GemmArgs g = conv(M, N, taps, cin_pad, (const float*)A, lda, rows_in, (const float*)Wp, bias);
    g.x3 = 1;
    g.a_planes = 1; g.a_plane = a_plane;
    g.w_planes = 1; g.w_plane = (long long)N * taps * cin_pad;
    return g;
}
void out_planes(GemmArgs& g, long long plane) { g.out_planes = 1; g.out_plane = plane; }
void out2_planes(GemmArgs& g, long long plane) { g.out2_planes = 1; g.out2_plane = plane; }

// One ResBlock (generator.py:110-117).  x_raw: [L][C] input (kept intact), x_act = snake(x_raw, alpha1[0]) prepared by the caller.
// Final value v = convs2[2](...) + cur (+ res2) (/ div) goes to `out` (raw, may be null) and act2(v) to `out2` (may be null).
// pp: the activated tensors (x_act, t1, actA / actB, out2) are (hi, lo) bf16 plane pairs L * Cp elements apart and the convolutions read
// them and the weight planes by LDS-DMA; the raw residual stream (x_raw, curA / curB, out, res2) stays fp32 — it is never a GEMM operand.
int resblock(hipStream_t s, WCursor& wc, bool pp, int L, int C, int k, const int* dils, const float* x_raw, const float* x_act, float* t1,
             float* curA, float* curB, float* actA, float* actB, float* out, const float* res2, float div, float* out2, int act2, float act2_param) {
    const int Cp = pad32(C);
    const bool x3 = wc.x3;
    const long long plane = (long long)L * Cp;
    const float* w1[3]; const float* b1[3]; const float* w2[3]; const float* b2[3]; const float* a1[3]; const float* a2[3];
    const void* p1[3]; const void* p2[3];
    for (int d = 0; d < 3; ++d) {
        w1[d] = wc.next(); p1[d] = wc.planes_of_last(); b1[d] = wc.next(); w2[d] = wc.next(); p2[d] = wc.planes_of_last(); b2[d] = wc.next();
        a1[d] = wc.next(); a2[d] = wc.next();
        if (pp && (!p1[d] || !p2[d])) return set_error("hvx_hift_decode: a ResBlock convolution has no weight planes"), -1;
    }
    const float* cur = x_raw;
    const float* cur_act = x_act;
    float* raw_bufs[2] = {curA, curB};
    float* act_bufs[2] = {actA, actB};
    for (int d = 0; d < 3; ++d) {
        GemmArgs g = pp ? convp(L, C, k, Cp, cur_act, plane, Cp, L, p1[d], b1[d]) : conv3(x3, L, C, k, Cp, cur_act, Cp, L, w1[d], b1[d]);
        g.conv_dil = dils[d]; g.pad_left = (k - 1) * dils[d];
        g.act = ACT_SNAKE; g.act_alpha = a2[d];
        g.out = t1; g.out_f32 = 1; g.ldo = Cp; g.out_cols = Cp;
        if (pp) out_planes(g, plane);
        HVX_CHECK(launch_gemm(g, s));
        g = pp ? convp(L, C, k, Cp, t1, plane, Cp, L, p2[d], b2[d]) : conv3(x3, L, C, k, Cp, t1, Cp, L, w2[d], b2[d]);
        g.pad_left = k - 1;
        g.res = cur; g.ldres = Cp;
        if (d < 2) {
            g.out = raw_bufs[d & 1]; g.out_f32 = 1; g.ldo = Cp; g.out_cols = Cp;
            g.out2 = act_bufs[d & 1]; g.act2 = ACT_SNAKE; g.act2_alpha = a1[d + 1]; g.ldo2 = Cp; g.out2_cols = Cp;
            if (pp) out2_planes(g, plane);
            HVX_CHECK(launch_gemm(g, s));
            cur = raw_bufs[d & 1];
            cur_act = act_bufs[d & 1];
        } else {
            g.res2 = res2; g.ldres2 = Cp; g.div = div;
            g.out = out; g.out_f32 = 1; g.ldo = Cp; g.out_cols = Cp;
            g.out2 = out2; g.act2 = act2; g.act2_param = act2_param; g.ldo2 = Cp; g.out2_cols = Cp;
            if (pp && out2) out2_planes(g, plane);
            HVX_CHECK(launch_gemm(g, s));
        }
    }
    return 0;
}

int expected_weights(const hvx_hift_config& c) { return 5 * 2 + 2 + 2 + 2 + c.n_up * (2 + 2 + 18 + c.n_rb * 18) + 2; }

}  // namespace

extern "C" {

int hvx_hift_create(const hvx_hift_config* cfg, const void* const* weights, int32_t n_weights, hvx_hift** out) {
    if (!cfg || !weights || !out) return set_error("hvx_hift_create: null argument"), -1;
    if (cfg->n_up < 1 || cfg->n_up > 4 || cfg->n_rb < 1 || cfg->n_rb > 4 || cfg->n_fft != 16 || cfg->hop != 4)
        return set_error("hvx_hift_create: unsupported configuration (n_fft/hop must be 16/4)"), -1;
    const int expect = expected_weights(*cfg);
    if (n_weights != expect) return set_error("hvx_hift_create: expected %d weight pointers, got %d", expect, n_weights), -1;
    for (int i = 0; i < n_weights; ++i)
        if (!weights[i]) return set_error("hvx_hift_create: weight %d is null", i), -1;
    hvx_hift* h = new hvx_hift();
    h->c = *cfg;
    h->w.assign(weights, weights + n_weights);
    h->wp.assign(n_weights, nullptr);
    h->up_total = cfg->hop;
    for (int i = 0; i < cfg->n_up; ++i) h->up_total *= cfg->up_rates[i];
    *out = h;
    return 0;
}
void hvx_hift_destroy(hvx_hift* h) { delete h; }

int hvx_hift_set_weight_planes(hvx_hift* h, const void* const* planes, int32_t n) {
    if (!h || !planes || n != (int)h->w.size()) return set_error("hvx_hift_set_weight_planes: expected %d entries", h ? (int)h->w.size() : 0), -1;
    h->wp.assign(planes, planes + n);
    return 0;
}

size_t hvx_hift_workspace_bytes(const hvx_hift* h, int32_t t) {
    Bufs b;
    return carve(h, nullptr, t, b) + 256;
}

int hvx_hift_f0(hvx_hift* h, hvx_stream stream, void* ws, size_t ws_bytes, const float* mel, int32_t T, float* f0) {
    hipStream_t s = (hipStream_t)stream;
    const hvx_hift_config& c = h->c;
    Bufs b;
    if (carve(h, (char*)ws, T, b) > ws_bytes) return set_error("hvx_hift_f0: workspace too small"), -1;
    WCursor wc{h->w.data(), (int)h->w.size()};
    const int melp = pad32(c.mel), Fp = pad32(c.f0_channels), F = c.f0_channels;
    HIP_OK(hipMemsetAsync(b.melT, 0, (size_t)T * melp * 4, s));
    HVX_CHECK(launch_transpose_f32(mel, b.melT, c.mel, T, T, melp, s));
    float* cur = b.fa;
    float* nxt = b.fb;
    for (int i = 0; i < 5; ++i) {
        const float* W = wc.next();
        const float* bias = wc.next();
        GemmArgs g = (i == 0) ? conv(T, F, 4, melp, b.melT, melp, T, W, bias)      // k=4, right-looking (causal_type='right')
                              : conv(T, F, 3, Fp, cur, Fp, T, W, bias);            // k=3, left pad 2
        if (i > 0) g.pad_left = 2;
        g.act = ACT_ELU;
        g.out = (i == 0) ? cur : nxt; g.out_f32 = 1; g.ldo = Fp; g.out_cols = Fp;
        HVX_CHECK(launch_gemm(g, s));
        if (i > 0) { float* t = cur; cur = nxt; nxt = t; }
    }
    const float* W = wc.next();
    const float* bias = wc.next();
    GemmArgs g = conv(T, 1, 1, Fp, cur, Fp, T, W, bias);
    g.act = ACT_ABS;
    g.out = f0; g.out_f32 = 1; g.ldo = 1; g.out_cols = 1;
    HVX_CHECK(launch_gemm(g, s));
    return 0;
}

int hvx_hift_source(hvx_hift* h, hvx_stream stream, void* ws, size_t ws_bytes, const float* f0, int32_t T, const float* sine_table,
                    float* source) {
    hipStream_t s = (hipStream_t)stream;
    const hvx_hift_config& c = h->c;
    Bufs b;
    if (carve(h, (char*)ws, T, b) > ws_bytes) return set_error("hvx_hift_source: workspace too small"), -1;
    const int H = c.nb_harmonics + 1;
    const float* lw = (const float*)h->w[12];
    const float* lb = (const float*)h->w[13];
    HVX_CHECK(launch_hift_phase(f0, b.phase, T, H, c.sampling_rate, h->up_total, s));
    HVX_CHECK(launch_hift_source(f0, b.phase, sine_table, lw, lb, source, T, H, h->up_total, c.nsf_alpha, c.nsf_sigma, c.voiced_threshold, s));
    return 0;
}

// decode over T_in mel frames of which the last `look` are real right context of conv_pre instead of zero padding (generator.py:674-680,
// finalize=False): T = T_in - look output frames; the source STFT is taken over all T_in * up samples and cut to 120 T + 1 frames.
static int decode_impl(hvx_hift* h, hvx_stream stream, void* ws, size_t ws_bytes, const float* mel, const float* source, int32_t T_in, int look,
                       float* wav) {
    hipStream_t s = (hipStream_t)stream;
    const hvx_hift_config& c = h->c;
    const int T = T_in - look;
    if (look < 0 || T <= 0) return set_error("hvx_hift_decode: %d frames with %d of look-ahead", T_in, look), -1;
    Bufs b;
    if (carve(h, (char*)ws, T_in, b) > ws_bytes) return set_error("hvx_hift_decode: workspace too small"), -1;
    WCursor wc{h->w.data(), (int)h->w.size()};
    wc.i = 14;                                           // skip f0 predictor (12) + source linear (2)
    wc.wp = h->wp.data();
    // plane-pair dataflow (see resblock): on when the caller handed over weight planes and the split-bf16 form is allowed
    const bool x3 = c.exact_fp32 == 0;
    wc.x3 = x3;
    bool pp = x3;
    {
        bool any = false;
        for (const void* p : h->wp) any = any || p != nullptr;
        pp = pp && any;
    }
    const int melp = pad32(c.mel);
    const long long Ls = (long long)T * h->up_total;
    const int frames = (int)(Ls / c.hop + 1);
    // source STFT -> [frames][32] (18 used)
    HVX_CHECK(launch_hift_stft(source, b.spec, (int)((long long)T_in * h->up_total), 32, s));
    // conv_pre (right-looking k) + LeakyReLU(slope) of stage 0
    HIP_OK(hipMemsetAsync(b.melT, 0, (size_t)T_in * melp * 4, s));
    HVX_CHECK(launch_transpose_f32(mel, b.melT, c.mel, T_in, T_in, melp, s));
    float** P = b.pool;
    float* xin = P[0];
    {
        const float* W = wc.next();
        const float* bias = wc.next();
        const int C0 = c.base_channels;
        GemmArgs g = conv3(x3, T, C0, c.conv_pre_kernel, melp, b.melT, melp, T_in, W, bias);
        g.act = ACT_LRELU; g.act_param = c.lrelu_slope;
        g.out = xin; g.out_f32 = 1; g.ldo = pad32(C0); g.out_cols = pad32(C0);
        if (pp) out_planes(g, (long long)T * pad32(C0));
        HVX_CHECK(launch_gemm(g, s));
    }
    // source down-sampling rates: cumprod([1] + rates[::-1][:-1])[::-1]   (generator.py:637-640)
    int down[4];
    {
        int acc = 1;
        for (int i = c.n_up - 1; i >= 0; --i) {
            down[i] = acc;
            acc *= c.up_rates[i];      // rates reversed, dropping the last of the reversed list (= rates[0])
        }
    }
    long long Lprev = T;
    int Cprev = c.base_channels;
    for (int i = 0; i < c.n_up; ++i) {
        const bool last = (i == c.n_up - 1);
        const int C = c.base_channels >> (i + 1), Cp = pad32(C), u = c.up_rates[i], ku = c.up_kernels[i];
        const long long Lup = Lprev * u;
        const long long L = last ? Lup + 1 : Lup;               // reflection pad (1, 0) on the last stage
        if ((size_t)L * Cp > b.pool_floats) return set_error("hvx_hift_decode: pool too small"), -1;
        const float* upW = wc.next();
        const void* upP = wc.planes_of_last();
        const float* upB = wc.next();
        if (pp && !upP) return set_error("hvx_hift_decode: an up-sampling convolution has no weight planes"), -1;
        const long long plane = (long long)L * Cp;              // between the (hi, lo) planes of every [L][Cp] activation of this stage
        const float* sdW = wc.next();
        const float* sdB = wc.next();
        // ---- source branch: down-sample conv + ResBlock -> si [L][C]
        float *si = P[1], *sd_raw = P[2], *sd_act = P[3], *t1 = P[4], *cA = P[5], *cB = P[6], *aA = P[7], *aB = P[8];
        {
            const int d = down[i];
            GemmArgs g;
            if (d == 1) {
                g = conv3(x3, (int)L, C, 1, 32, b.spec, 32, frames, sdW, sdB);
            } else {
                g = conv3(x3, (int)L, C, 2 * d, 32, b.spec, 32, frames, sdW, sdB);
                g.conv_stride = d; g.pad_left = d - 1;
            }
            g.out = sd_raw; g.out_f32 = 1; g.ldo = Cp; g.out_cols = Cp;
            // first Snake of the source ResBlock: its alpha is 4 entries ahead of the cursor (w1,b1,w2,b2,a1,...)
            g.out2 = sd_act; g.act2 = ACT_SNAKE; g.act2_alpha = (const float*)h->w[wc.i + 4]; g.ldo2 = Cp; g.out2_cols = Cp;
            if (pp) out2_planes(g, plane);
            HVX_CHECK(launch_gemm(g, s));
            HVX_CHECK(resblock(s, wc, pp, (int)L, C, c.src_rb_kernels[i], c.src_rb_dils[i], sd_raw, sd_act, t1, cA, cB, aA, aB, si, nullptr, 0.0f,
                               nullptr, ACT_NONE, 0.0f));
        }
        // ---- nearest-upsample + causal conv, + source branch -> x_raw (P[2]); last stage shifted by the reflection pad
        float* x_raw = P[2];
        {
            const int Cpp = pad32(Cprev);
            GemmArgs g = pp ? convp((int)Lup, C, ku, Cpp, xin, Lprev * Cpp, Cpp, (int)Lprev, upP, upB) : conv3(x3, (int)Lup, C, ku, Cpp, xin, Cpp, (int)Lprev, upW, upB);
            g.up = u; g.pad_left = ku - 1;
            g.res = si; g.ldres = Cp;
            g.out = x_raw; g.out_f32 = 1; g.ldo = Cp; g.out_cols = Cp;
            if (last) {
                g.out_row_off = 1; g.res_row_off = 1;
                HVX_CHECK(launch_gemm(g, s));
                g.M = 2; g.out_row_off = -1; g.res_row_off = -1;       // row 0 = reflect(x_up)[0] + si[0] = x_up[1] + si[0]
                HVX_CHECK(launch_gemm(g, s));
            } else {
                HVX_CHECK(launch_gemm(g, s));
            }
        }
        // ---- mean of the n_rb ResBlocks, then LeakyReLU for the next consumer
        float* xs = P[1];                     // si is dead after the up conv: reuse its buffer as the accumulator
        float* xnext = P[0];                  // xin is dead as well
        float* x_act = P[3];
        for (int j = 0; j < c.n_rb; ++j) {
            const bool lastj = (j == c.n_rb - 1);
            const float* a1_0 = (const float*)h->w[wc.i + 4];
            if (pp) HVX_CHECK(launch_act_rows_planes(x_raw, Cp, x_act, Cp, plane, ACT_SNAKE, 0.0f, a1_0, L, C, s));
            else HVX_CHECK(launch_act_rows(x_raw, Cp, x_act, Cp, DT_F32, ACT_SNAKE, 0.0f, a1_0, L, C, s));
            HVX_CHECK(resblock(s, wc, pp, (int)L, C, c.rb_kernels[j], c.rb_dils[j], x_raw, x_act, t1, cA, cB, aA, aB,
                               lastj ? nullptr : xs, j == 0 ? nullptr : xs, lastj ? (float)c.n_rb : 0.0f,
                               lastj ? xnext : nullptr, ACT_LRELU, last ? 0.01f : c.lrelu_slope));
        }
        xin = xnext;
        Lprev = L;
        Cprev = C;
    }
    // conv_post (k, left) -> exp / sin -> iSTFT -> clamp
    {
        const float* W = wc.next();
        const void* Wp = wc.planes_of_last();
        const float* bias = wc.next();
        const int Cpp = pad32(Cprev);
        if (pp && !Wp) return set_error("hvx_hift_decode: conv_post has no weight planes"), -1;
        GemmArgs g = pp ? convp((int)Lprev, c.n_fft + 2, c.conv_post_kernel, Cpp, xin, Lprev * Cpp, Cpp, (int)Lprev, Wp, bias)
                        : conv3(x3, (int)Lprev, c.n_fft + 2, c.conv_post_kernel, Cpp, xin, Cpp, (int)Lprev, W, bias);
        g.pad_left = c.conv_post_kernel - 1;
        g.out = b.post; g.out_f32 = 1; g.ldo = 32; g.out_cols = 32;
        HVX_CHECK(launch_gemm(g, s));
    }
    if (wc.i != (int)h->w.size()) return set_error("hvx_hift_decode: consumed %d of %zu weights", wc.i, h->w.size()), -1;
    if (Lprev != frames) return set_error("hvx_hift_decode: %lld conv frames vs %d stft frames", Lprev, frames), -1;
    HVX_CHECK(launch_hift_istft(b.post, 32, wav, frames, c.audio_limit, s));
    return 0;
}

int hvx_hift_decode(hvx_hift* h, hvx_stream stream, void* ws, size_t ws_bytes, const float* mel, const float* source, int32_t T, float* wav) {
    return decode_impl(h, stream, ws, ws_bytes, mel, source, T, 0, wav);
}
int hvx_hift_decode_chunk(hvx_hift* h, hvx_stream stream, void* ws, size_t ws_bytes, const float* mel, const float* source, int32_t t_in,
                          int32_t look_right, float* wav) {
    return decode_impl(h, stream, ws, ws_bytes, mel, source, t_in, look_right, wav);
}

}  // extern "C"
