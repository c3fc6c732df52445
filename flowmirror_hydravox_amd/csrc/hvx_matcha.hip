// hvx_matcha.hip — the Matcha-TTS / HiFi-GAN v1 family of the path (include/hvx.h: hvx_matcha_*,
// hvx_hifigan_*, hvx_denoise).
//
// Restates (file:line in the reference tree):
//   matcha/models/components/flow_matching.py:32-85     BASECFM.forward / solve_euler (plain Euler, no CFG)
//   matcha/models/components/decoder.py:363-443         Decoder.forward (1-D U-Net), :40-75 Block1D / ResnetBlock1D, :78-95 Downsample1D,
//                                                       :135-170 Upsample1D (ConvTranspose1d k4 s2 p1), :12-29 SinusoidalPosEmb, :98-132 TimestepEmbedding
//   server/model_utils/cosyvoice/flow/decoder.py:210-291   ConditionalDecoder.forward (adds `cond`, key-padding bias masks, skip trimming)
//   matcha/models/components/transformer.py:243-316     BasicTransformerBlock (LayerNorm -> self-attention -> LayerNorm -> FeedForward), :17-80 SnakeBeta
//   matcha/hifigan/models.py:181-197                    Generator.forward, :90-97 ResBlock1;  matcha/hifigan/config.py v1
//   matcha/hifigan/denoiser.py:57-64                    Denoiser.forward
// All arithmetic is fp32 (the reference runs this family in fp32).  Activations are time-major rows [B][T][C]; every Conv1d — the
// stride-2 down-sampler included — is one implicit-GEMM launch of gemm_tiled, every ConvTranspose1d is `stride` launches (one
// per output phase: a k/stride-tap convolution whose rows land `stride` apart), the STFT / inverse STFT of the denoiser are
// GEMMs against windowed DFT bases built at load time.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "hvx.h"
#include "hvx_device.h"
#include "hvx_kernels.h"

using namespace hvx;

struct hvx_matcha {
    hvx_matcha_config c;
    std::vector<const void*> w;
};
struct hvx_hifigan {
    hvx_hifigan_config c;
    std::vector<const void*> w;
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

GemmArgs conv(int M, int N, int taps, int cin_pad, const void* A, int lda, int rows_in, const void* W, const float* bias) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.dtype = DT_F32; g.M = M; g.N = N; g.K = taps * cin_pad; g.batch = 1; g.groups = 1;
    g.A = A; g.lda = lda; g.rows_in = rows_in; g.cin_pad = cin_pad; g.conv_stride = 1; g.conv_dil = 1; g.pad_left = 0; g.up = 1;
    g.W = W; g.epi = EPI_GENERIC; g.bias = bias; g.scale = 1.0f; g.out_f32 = 1;
    return g;
}
// the HiFi-GAN v1 generator's convolutions: fp32 operands split into bf16 pairs on the bf16 matrix cores (gemm_x3.hip, ~4e-6 of the output scale,
// the reference's contract here is 1e-3); hvx_hifigan_config.exact_fp32 keeps the exact fp32 MFMA form (x3 = false), as for the HiFT vocoder
GemmArgs convx(bool x3, int M, int N, int taps, int cin_pad, const void* A, int lda, int rows_in, const void* W, const float* bias) {
    GemmArgs g = conv(M, N, taps, cin_pad, A, lda, rows_in, W, bias);
    g.x3 = x3 ? 1 : 0;
    return g;
}
void batched(GemmArgs& g, int B, long long a_bs, long long out_bs) {
    g.batch = B; g.a_bs = a_bs; g.out_bs = out_bs;
}

// ---------------------------------------------------------------------------------------------------------------------
// Decoder
// ---------------------------------------------------------------------------------------------------------------------
struct MBufs {
    float *tsin, *th, *temb, *tmish, *tbias;     // time path
    float *xin, *a1, *h1, *h2, *x, *cat, *n, *q, *k, *vT, *att, *ffh, *outrow, *gn;
    float* skip[4];
    float *x_state, *t_dev;                        // solver
    int t_pad;
};

int n_units(const hvx_matcha_config& c) { return 2 * c.n_stages + c.n_mid; }
int weights_per_unit(const hvx_matcha_config& c, bool resample) { return 12 + 12 * c.n_blocks + (resample ? 2 : 0); }
int expected_weights(const hvx_matcha_config& c) {
    return 4 + 2 * c.n_stages * weights_per_unit(c, true) + c.n_mid * weights_per_unit(c, false) + 6;
}
int cmax(const hvx_matcha_config& c) {
    int m = pad32(c.in_channels);
    for (int i = 0; i < c.n_stages; ++i) m = m > 2 * c.channels[i] ? m : 2 * c.channels[i];
    return m;
}

size_t carve_m(const hvx_matcha_config& c, char* base, int B, int T, MBufs& b) {
    Carve cv(base);
    const int TE = c.channels[0] * 4, CM = cmax(c), inner = c.heads * 64, FF = c.ff_mult * CM;
    const int Tp = (T + 63) / 64 * 64;
    b.t_pad = Tp;
    const size_t rows = (size_t)B * T;
    b.tsin = cv.take((size_t)B * pad32(c.in_channels));
    b.th = cv.take((size_t)B * TE);
    b.temb = cv.take((size_t)B * TE);
    b.tmish = cv.take((size_t)B * TE);
    b.tbias = cv.take((size_t)n_units(c) * B * CM);
    b.xin = cv.take(rows * CM);
    b.a1 = cv.take(rows * CM);
    b.h1 = cv.take(rows * CM);
    b.h2 = cv.take(rows * CM);
    b.x = cv.take(rows * CM);
    b.cat = cv.take(rows * CM);
    b.n = cv.take(rows * CM);
    b.q = cv.take((size_t)B * c.heads * Tp * 64);
    b.k = cv.take((size_t)B * c.heads * Tp * 64);
    b.vT = cv.take((size_t)B * c.heads * Tp * 64);
    b.att = cv.take(rows * inner);
    b.ffh = cv.take(rows * FF);
    b.outrow = cv.take(rows * pad32(c.out_channels));
    b.gn = cv.take(groupnorm_ws_bytes(B, T, 8) / 4 + 64);
    for (int i = 0; i < 4; ++i) b.skip[i] = i < c.n_stages ? cv.take(rows * CM) : nullptr;
    b.x_state = cv.take((size_t)B * c.out_channels * T);
    b.t_dev = cv.take(64);
    return cv.off;
}

__global__ void fill_f32_kernel(float* p, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct WCursor {
    const void* const* w;
    int i = 0;
    const void* next() { return w[i++]; }
    const float* nextf() { return (const float*)w[i++]; }
};

// ResnetBlock1D (decoder.py:57-75).  x: [B][T][cin_pad] (rows >= len already zero), out -> o [B][T][ldo], channels Cout
int resnet_block(hipStream_t s, WCursor& wc, const MBufs& b, int B, int T, const float* x, int cin_pad, int Cout, const int* len,
                 const float* tmish, int TE, float* tbias, float* o, int ldo) {
    const float* mlp_w = wc.nextf(); const float* mlp_b = wc.nextf();
    const void* c1_w = wc.next(); const float* c1_b = wc.nextf(); const float* g1 = wc.nextf(); const float* be1 = wc.nextf();
    const void* c2_w = wc.next(); const float* c2_b = wc.nextf(); const float* g2 = wc.nextf(); const float* be2 = wc.nextf();
    const void* r_w = wc.next(); const float* r_b = wc.nextf();
    // time bias: Linear(Mish(t_emb))   (:61, :70)
    GemmArgs g = conv(B, Cout, 1, TE, tmish, TE, B, mlp_w, mlp_b);
    g.out = tbias; g.ldo = Cout; g.out_cols = Cout;
    HVX_CHECK(launch_gemm(g, s));
    // block1: conv k3 p1 -> GroupNorm(8) -> Mish -> mask, + time bias
    g = conv(T, Cout, 3, cin_pad, x, cin_pad, T, c1_w, c1_b);
    g.pad_left = 1; batched(g, B, (long long)T * cin_pad, (long long)T * Cout);
    g.out = b.a1; g.ldo = Cout; g.out_cols = Cout;
    HVX_CHECK(launch_gemm(g, s));
    HVX_CHECK(launch_groupnorm_act(b.a1, Cout, B, T, Cout, 8, len, g1, be1, 1e-5f, tbias, ACT_MISH, b.h1, Cout, b.gn, s));
    // block2
    g = conv(T, Cout, 3, Cout, b.h1, Cout, T, c2_w, c2_b);
    g.pad_left = 1; batched(g, B, (long long)T * Cout, (long long)T * Cout);
    g.out = b.a1; g.ldo = Cout; g.out_cols = Cout;
    HVX_CHECK(launch_gemm(g, s));
    HVX_CHECK(launch_groupnorm_act(b.a1, Cout, B, T, Cout, 8, len, g2, be2, 1e-5f, nullptr, ACT_MISH, b.h2, Cout, b.gn, s));
    // output = h + res_conv(x * mask)
    g = conv(T, Cout, 1, cin_pad, x, cin_pad, T, r_w, r_b);
    batched(g, B, (long long)T * cin_pad, (long long)T * ldo);
    g.res = b.h2; g.res_bs = (long long)T * Cout; g.ldres = Cout;
    g.out = o; g.ldo = ldo; g.out_cols = Cout;
    return launch_gemm(g, s);
}

// BasicTransformerBlock (transformer.py:243-316) in place on x [B][T][C]
int transformer_block(hipStream_t s, WCursor& wc, const MBufs& b, const hvx_matcha_config& c, int B, int T, float* x, int C, const int* len) {
    const float* n1_scale = wc.nextf(); const float* n1_shift = wc.nextf();       // gamma - 1, beta
    const void* wqkv = wc.next();
    const void* wo = wc.next(); const float* bo = wc.nextf();
    const float* n3_scale = wc.nextf(); const float* n3_shift = wc.nextf();
    const void* w1 = wc.next(); const float* b1 = wc.nextf(); const float* snake = wc.nextf();
    const void* w2 = wc.next(); const float* b2 = wc.nextf();
    const int H = c.heads, inner = H * 64, FF = c.ff_mult * C, Tp = b.t_pad;
    HVX_CHECK(launch_layernorm_mod(x, 0, n1_shift, n1_scale, 0, 1e-5f, b.n, DT_F32, B, T, C, s));
    GemmArgs g = conv(T, 3 * inner, 1, C, b.n, C, T, wqkv, nullptr);
    batched(g, B, (long long)T * C, 0);
    g.epi = EPI_QKV_DIT; g.q = b.q; g.k = b.k; g.vT = b.vT; g.heads = H; g.t_pad = Tp;
    HVX_CHECK(launch_gemm(g, s));
    AttnArgs at;
    memset(&at, 0, sizeof(at));
    at.dtype = DT_F32; at.batch = B; at.heads = H; at.n_rows = T; at.kn = T;
    at.q = b.q; at.q_bs = (long long)H * Tp * 64; at.q_hs = (long long)Tp * 64; at.q_lo = 64;
    at.k = b.k; at.k_bs = at.q_bs; at.k_hs = at.q_hs;
    at.vT = b.vT; at.v_bs = at.q_bs; at.v_hs = (long long)64 * Tp; at.v_ld = Tp;
    at.kv_len = len; at.kv_len_const = T; at.causal = 0; at.scale = 0.125f;
    at.out = b.att; at.o_bs = (long long)T * inner; at.o_hs = 64; at.o_lo = inner; at.n_splits = 1;
    HVX_CHECK(launch_attention(at, s));
    g = conv(T, C, 1, inner, b.att, inner, T, wo, bo);
    batched(g, B, (long long)T * inner, (long long)T * C);
    g.res = x; g.res_bs = (long long)T * C; g.ldres = C; g.out = x; g.ldo = C; g.out_cols = C;
    HVX_CHECK(launch_gemm(g, s));
    HVX_CHECK(launch_layernorm_mod(x, 0, n3_shift, n3_scale, 0, 1e-5f, b.n, DT_F32, B, T, C, s));
    g = conv(T, FF, 1, C, b.n, C, T, w1, b1);
    batched(g, B, (long long)T * C, (long long)T * FF);
    g.act = ACT_SNAKEBETA; g.act_alpha = snake; g.out = b.ffh; g.ldo = FF; g.out_cols = FF;
    HVX_CHECK(launch_gemm(g, s));
    g = conv(T, C, 1, FF, b.ffh, FF, T, w2, b2);
    batched(g, B, (long long)T * FF, (long long)T * C);
    g.res = x; g.res_bs = (long long)T * C; g.ldres = C; g.out = x; g.ldo = C; g.out_cols = C;
    return launch_gemm(g, s);
}

// lens: optional device int32 [n_stages][B] (row i = valid length at U-Net level i); null = every row valid
int estimator_core(const hvx_matcha* h, hipStream_t s, MBufs& b, int B, int T, const float* x, const float* mu, const float* spks, int spk_dim,
                   const float* cond, const int* lens, const float* t, float* out_rows) {
    const hvx_matcha_config& c = h->c;
    if (T > c.max_t) return set_error("matcha estimator: T=%d exceeds max_t=%d", T, c.max_t), -1;
    const int mel = c.out_channels, Cin = c.in_channels, Cin_p = pad32(Cin), TE = c.channels[0] * 4;
    if (2 * mel + (spks ? spk_dim : 0) + (cond ? mel : 0) != Cin)
        return set_error("matcha estimator: in_channels=%d does not match x + mu%s%s", Cin, spks ? " + spks" : "", cond ? " + cond" : ""), -1;
    WCursor wc{h->w.data()};
    HIP_OK(hipMemsetAsync(b.vT, 0, (size_t)B * c.heads * b.t_pad * 64 * 4, s));       // padded key columns are multiplied by p == 0: keep them finite
    // ---- time embedding: SinusoidalPosEmb(in_channels) -> Linear -> SiLU -> Linear; Mish(.) feeds every resnet block -------------------
    HVX_CHECK(launch_time_sinus(t, b.tsin, DT_F32, B, Cin, s));
    const void* tw1 = wc.next(); const float* tb1 = wc.nextf(); const void* tw2 = wc.next(); const float* tb2 = wc.nextf();
    GemmArgs g = conv(B, TE, 1, Cin_p, b.tsin, Cin_p, B, tw1, tb1);
    g.act = ACT_SILU; g.out = b.th; g.ldo = TE; g.out_cols = TE;
    HVX_CHECK(launch_gemm(g, s));
    g = conv(B, TE, 1, TE, b.th, TE, B, tw2, tb2);
    g.out = b.temb; g.ldo = TE; g.out_cols = TE;
    HVX_CHECK(launch_gemm(g, s));
    HVX_CHECK(launch_act_rows(b.temb, TE, b.tmish, TE, DT_F32, ACT_MISH, 0.0f, nullptr, B, TE, s));
    // ---- input rows [x | mu | spks | cond] -----------------------------------------------------------------------------------------
    PackRowsArgs pk;
    memset(&pk, 0, sizeof(pk));
    pk.src[0] = x; pk.channels[0] = mel;
    pk.src[1] = mu; pk.channels[1] = mel;
    pk.src[2] = spks; pk.channels[2] = spks ? spk_dim : 0; pk.broadcast[2] = 1;
    pk.src[3] = cond; pk.channels[3] = cond ? mel : 0;
    pk.dst = b.xin; pk.ld = Cin_p; pk.T = T;
    HVX_CHECK(launch_pack_rows(pk, B, s));

    int Ts[5];
    Ts[0] = T;
    for (int i = 1; i <= c.n_stages; ++i) Ts[i] = (Ts[i - 1] + 1) / 2;
    float* cur = b.xin;
    int cur_c = Cin_p, Tc = T, unit = 0;
    const int* len = lens;
    // ---- down path --------------------------------------------------------------------------------------------------------------------
    for (int i = 0; i < c.n_stages; ++i) {
        const int C = c.channels[i];
        len = lens ? lens + (size_t)i * B : nullptr;
        HVX_CHECK(launch_mask_rows(cur, cur_c, B, Tc, cur_c, len, s));
        HVX_CHECK(resnet_block(s, wc, b, B, Tc, cur, cur_c, C, len, b.tmish, TE, b.tbias + (size_t)unit * B * cmax(c), b.x, C));
        for (int j = 0; j < c.n_blocks; ++j) HVX_CHECK(transformer_block(s, wc, b, c, B, Tc, b.x, C, len));
        HIP_OK(hipMemcpyAsync(b.skip[i], b.x, (size_t)B * Tc * C * 4, hipMemcpyDeviceToDevice, s));
        HVX_CHECK(launch_mask_rows(b.x, C, B, Tc, C, len, s));
        const void* dw = wc.next(); const float* db = wc.nextf();
        const bool last = i == c.n_stages - 1;
        const int Tn = last ? Tc : Ts[i + 1];
        g = conv(Tn, C, 3, C, b.x, C, Tc, dw, db);
        g.pad_left = 1; g.conv_stride = last ? 1 : 2;
        batched(g, B, (long long)Tc * C, (long long)Tn * C);
        g.out = b.xin; g.ldo = C; g.out_cols = C;
        HVX_CHECK(launch_gemm(g, s));
        cur = b.xin; cur_c = C; Tc = Tn; ++unit;
    }
    // masks = masks[:-1]; mask_mid = masks[-1]: the middle runs at the resolution of the last down stage
    const int i_mid = c.n_stages - 1;
    const int* len_mid = lens ? lens + (size_t)i_mid * B : nullptr;
    const int Cm = c.channels[c.n_stages - 1];
    for (int m = 0; m < c.n_mid; ++m) {
        HVX_CHECK(launch_mask_rows(cur, cur_c, B, Tc, cur_c, len_mid, s));
        HVX_CHECK(resnet_block(s, wc, b, B, Tc, cur, cur_c, Cm, len_mid, b.tmish, TE, b.tbias + (size_t)unit * B * cmax(c), b.x, Cm));
        for (int j = 0; j < c.n_blocks; ++j) HVX_CHECK(transformer_block(s, wc, b, c, B, Tc, b.x, Cm, len_mid));
        HIP_OK(hipMemcpyAsync(b.xin, b.x, (size_t)B * Tc * Cm * 4, hipMemcpyDeviceToDevice, s));
        cur = b.xin; cur_c = Cm; ++unit;
    }
    // ---- up path: channels reversed + (channels[0],) ----------------------------------------------------------------------------------
    const int* len_up = nullptr;
    for (int i = 0; i < c.n_stages; ++i) {
        const int lvl = c.n_stages - 1 - i;                      // skip / mask level popped from the stacks
        const int Cs = c.channels[lvl];                           // channels of x and of the skip at this level
        const int Co = i + 1 < c.n_stages ? c.channels[lvl - 1] : c.channels[0];
        const int Tl = Ts[lvl];
        len_up = lens ? lens + (size_t)lvl * B : nullptr;
        // pack([x[:, :, :T_skip], skip]): two strided copies into [B][Tl][2*Cs]
        for (int bb = 0; bb < B; ++bb) {
            HIP_OK(hipMemcpy2DAsync(b.cat + (size_t)bb * Tl * 2 * Cs, (size_t)2 * Cs * 4, cur + (size_t)bb * Tc * cur_c, (size_t)cur_c * 4, (size_t)Cs * 4, Tl,
                                    hipMemcpyDeviceToDevice, s));
            HIP_OK(hipMemcpy2DAsync(b.cat + (size_t)bb * Tl * 2 * Cs + Cs, (size_t)2 * Cs * 4, b.skip[lvl] + (size_t)bb * Tl * Cs, (size_t)Cs * 4, (size_t)Cs * 4, Tl,
                                    hipMemcpyDeviceToDevice, s));
        }
        HVX_CHECK(launch_mask_rows(b.cat, 2 * Cs, B, Tl, 2 * Cs, len_up, s));
        HVX_CHECK(resnet_block(s, wc, b, B, Tl, b.cat, 2 * Cs, Co, len_up, b.tmish, TE, b.tbias + (size_t)unit * B * cmax(c), b.x, Co));
        for (int j = 0; j < c.n_blocks; ++j) HVX_CHECK(transformer_block(s, wc, b, c, B, Tl, b.x, Co, len_up));
        HVX_CHECK(launch_mask_rows(b.x, Co, B, Tl, Co, len_up, s));
        const void* uw = wc.next(); const float* ub = wc.nextf();
        const bool last = i == c.n_stages - 1;
        if (last) {
            g = conv(Tl, Co, 3, Co, b.x, Co, Tl, uw, ub);
            g.pad_left = 1;
            batched(g, B, (long long)Tl * Co, (long long)Tl * Co);
            g.out = b.xin; g.ldo = Co; g.out_cols = Co;
            HVX_CHECK(launch_gemm(g, s));
            Tc = Tl;
        } else {
            // ConvTranspose1d(k4, s2, p1): output phase p is a 2-tap convolution, rows 2t + p   (c_p = (p + 1) / 2, pad_left = 1 - c_p)
            for (int p = 0; p < 2; ++p) {
                g = conv(Tl, Co, 2, Co, b.x, Co, Tl, (const float*)uw + (size_t)p * Co * 2 * Co, ub);
                g.pad_left = 1 - (p + 1) / 2;
                batched(g, B, (long long)Tl * Co, (long long)2 * Tl * Co);
                g.out = b.xin + (size_t)p * Co; g.ldo = 2 * Co; g.out_cols = Co;
                HVX_CHECK(launch_gemm(g, s));
            }
            Tc = 2 * Tl;
        }
        cur = b.xin; cur_c = Co; ++unit;
    }
    // ---- final_block (Block1D with the last mask_up) and final_proj(x * mask_up) * mask ------------------------------------------------
    const int C0 = c.channels[0];
    const int* len0 = lens;                                      // level 0 == mask_up of the last up block == the input mask
    const void* fw = wc.next(); const float* fb = wc.nextf(); const float* fg = wc.nextf(); const float* fbe = wc.nextf();
    const void* pw = wc.next(); const float* pb = wc.nextf();
    HVX_CHECK(launch_mask_rows(cur, C0, B, T, C0, len0, s));
    g = conv(T, C0, 3, C0, cur, C0, T, fw, fb);
    g.pad_left = 1;
    batched(g, B, (long long)T * C0, (long long)T * C0);
    g.out = b.a1; g.ldo = C0; g.out_cols = C0;
    HVX_CHECK(launch_gemm(g, s));
    HVX_CHECK(launch_groupnorm_act(b.a1, C0, B, T, C0, 8, len0, fg, fbe, 1e-5f, nullptr, ACT_MISH, b.h1, C0, b.gn, s));
    const int mel_p = pad32(mel);
    g = conv(T, mel, 1, C0, b.h1, C0, T, pw, pb);
    batched(g, B, (long long)T * C0, (long long)T * mel_p);
    g.out = out_rows; g.ldo = mel_p; g.out_cols = mel_p;
    HVX_CHECK(launch_gemm(g, s));
    HVX_CHECK(launch_mask_rows(out_rows, mel_p, B, T, mel_p, len0, s));
    if (wc.i != (int)h->w.size()) return set_error("matcha estimator: consumed %d of %zu weights", wc.i, h->w.size()), -1;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// HiFi-GAN v1
// ---------------------------------------------------------------------------------------------------------------------
struct GBufs {
    float *melT, *x_act, *u, *ua, *t1, *xr, *xa, *acc;
};
int gen_cmax(const hvx_hifigan_config& c) { return c.initial_channel; }
long long gen_out_len(const hvx_hifigan_config& c, int T) {
    long long L = T;
    for (int i = 0; i < c.n_up; ++i) L *= c.up_rates[i];
    return L;
}
size_t carve_g(const hvx_hifigan_config& c, char* base, int T, GBufs& b) {
    Carve cv(base);
    // rows x channels is largest right after an upsample: L_i * C_i
    size_t big = (size_t)T * c.initial_channel;
    long long L = T;
    int C = c.initial_channel;
    for (int i = 0; i < c.n_up; ++i) {
        L *= c.up_rates[i];
        C /= 2;
        big = big > (size_t)L * pad32(C) ? big : (size_t)L * pad32(C);
    }
    b.melT = cv.take((size_t)T * pad32(c.mel));
    b.x_act = cv.take(big);
    b.u = cv.take(big);
    b.ua = cv.take(big);
    b.t1 = cv.take(big);
    b.xr = cv.take(big);
    b.xa = cv.take(big);
    b.acc = cv.take(big);
    return cv.off;
}
int gen_expected_weights(const hvx_hifigan_config& c) { return 2 + c.n_up * (2 + c.n_rb * 12) + 2; }

int generator_core(const hvx_hifigan* h, hipStream_t s, GBufs& b, const float* mel, int T, float* wav) {
    const hvx_hifigan_config& c = h->c;
    const bool x3 = c.exact_fp32 == 0;
    WCursor wc{h->w.data()};
    const int melp = pad32(c.mel);
    HIP_OK(hipMemsetAsync(b.melT, 0, (size_t)T * melp * 4, s));
    HVX_CHECK(launch_transpose_f32(mel, b.melT, c.mel, T, T, melp, s));           // (mel, T) -> [T][mel_pad]
    // conv_pre k7 p3; only lrelu(x, 0.1) is consumed (models.py:183-185)
    const void* w0 = wc.next(); const float* b0 = wc.nextf();
    int C = c.initial_channel;
    GemmArgs g = convx(x3, T, C, 7, melp, b.melT, melp, T, w0, b0);
    g.pad_left = 3; g.act = ACT_LRELU; g.act_param = 0.1f;
    g.out = b.x_act; g.ldo = C; g.out_cols = C;
    HVX_CHECK(launch_gemm(g, s));
    long long L = T;
    for (int i = 0; i < c.n_up; ++i) {
        const int r = c.up_rates[i], k = c.up_kernels[i], pd = (k - r) / 2, taps = k / r;
        const int Ci = C, Co = C / 2, Cop = pad32(Co);
        if (k % r) return set_error("hifigan: upsample kernel %d is not a multiple of the rate %d", k, r), -1;
        const float* uw = wc.nextf(); const float* ub = wc.nextf();
        // ConvTranspose1d as `r` phase convolutions: output row t*r + p, taps over input rows t + c_p - (taps-1) .. t + c_p
        for (int p = 0; p < r; ++p) {
            g = convx(x3, (int)L, Co, taps, Ci, b.x_act, Ci, (int)L, uw + (size_t)p * Co * taps * Ci, ub);
            g.pad_left = taps - 1 - (p + pd) / r;
            g.out = b.u + (size_t)p * Cop; g.ldo = r * Cop; g.out_cols = Cop;
            g.out2 = nullptr;
            HVX_CHECK(launch_gemm(g, s));
        }
        L *= r;
        C = Co;
        HVX_CHECK(launch_act_rows(b.u, Cop, b.ua, Cop, DT_F32, ACT_LRELU, 0.1f, nullptr, L, Cop, s));
        const bool last_stage = i == c.n_up - 1;
        for (int j = 0; j < c.n_rb; ++j) {
            const int kk = c.rb_kernels[j];
            const float* xr = b.u;                       // running residual (raw)
            const float* xa = b.ua;                      // its lrelu
            for (int d = 0; d < 3; ++d) {
                const int dil = c.rb_dils[j][d];
                const void* w1 = wc.next(); const float* b1 = wc.nextf(); const void* w2 = wc.next(); const float* b2 = wc.nextf();
                g = convx(x3, (int)L, Co, kk, Cop, xa, Cop, (int)L, w1, b1);
                g.conv_dil = dil; g.pad_left = dil * (kk - 1) / 2; g.act = ACT_LRELU; g.act_param = 0.1f;
                g.out = b.t1; g.ldo = Cop; g.out_cols = Cop;
                HVX_CHECK(launch_gemm(g, s));
                g = convx(x3, (int)L, Co, kk, Cop, b.t1, Cop, (int)L, w2, b2);
                g.pad_left = (kk - 1) / 2;
                g.res = xr; g.ldres = Cop;
                if (d < 2) {
                    g.out = b.xr; g.ldo = Cop; g.out_cols = Cop;
                    g.out2 = b.xa; g.act2 = ACT_LRELU; g.act2_param = 0.1f; g.ldo2 = Cop; g.out2_cols = Cop;     // f32 dtype: out2 is fp32 too
                } else if (j < c.n_rb - 1) {
                    // xs (+)= resblock_j(x)
                    if (j > 0) { g.res2 = b.acc; g.ldres2 = Cop; }
                    g.out = b.acc; g.ldo = Cop; g.out_cols = Cop;
                } else {
                    // x = xs / num_kernels, then the next consumer's leaky_relu (0.1 before an upsample, F.leaky_relu default 0.01 before conv_post)
                    if (j > 0) { g.res2 = b.acc; g.ldres2 = Cop; }
                    g.scale = 1.0f / (float)c.n_rb;
                    g.out = b.xr; g.ldo = Cop; g.out_cols = Cop;
                    g.out2 = b.x_act; g.act2 = ACT_LRELU; g.act2_param = last_stage ? 0.01f : 0.1f; g.ldo2 = Cop; g.out2_cols = Cop;
                }
                HVX_CHECK(launch_gemm(g, s));
                xr = b.xr;
                xa = b.xa;
            }
        }
        C = Cop;                                           // the next stage reads rows of the padded width
    }
    const void* wp = wc.next(); const float* bp = wc.nextf();
    g = convx(x3, (int)L, 1, 7, C, b.x_act, C, (int)L, wp, bp);
    g.pad_left = 3; g.act = ACT_TANH;
    g.out = wav; g.ldo = 1; g.out_cols = 1;
    HVX_CHECK(launch_gemm(g, s));
    if (wc.i != (int)h->w.size()) return set_error("hifigan: consumed %d of %zu weights", wc.i, h->w.size()), -1;
    return 0;
}

}  // namespace

extern "C" {

int hvx_matcha_create(const hvx_matcha_config* cfg, const void* const* weights, int32_t n_weights, hvx_matcha** out) {
    if (!cfg || !weights || !out) return set_error("hvx_matcha_create: null argument"), -1;
    if (cfg->n_stages < 1 || cfg->n_stages > 4 || cfg->heads < 1 || cfg->n_blocks < 0 || cfg->n_mid < 0 || cfg->ff_mult < 1)
        return set_error("hvx_matcha_create: unsupported geometry"), -1;
    if (cfg->in_channels % 32) return set_error("hvx_matcha_create: in_channels must be a multiple of 32"), -1;
    for (int i = 0; i < cfg->n_stages; ++i)
        if (cfg->channels[i] % 64 || cfg->channels[i] < 64) return set_error("hvx_matcha_create: channels must be multiples of 64 (GroupNorm 8 x 8k, 32-wide K steps)"), -1;
    const int expect = expected_weights(*cfg);
    if (n_weights != expect) return set_error("hvx_matcha_create: expected %d weight pointers, got %d", expect, n_weights), -1;
    for (int i = 0; i < n_weights; ++i)
        if (!weights[i]) return set_error("hvx_matcha_create: weight %d is null", i), -1;
    hvx_matcha* h = new hvx_matcha();
    h->c = *cfg;
    h->w.assign(weights, weights + n_weights);
    *out = h;
    return 0;
}
void hvx_matcha_destroy(hvx_matcha* h) { delete h; }
size_t hvx_matcha_workspace_bytes(const hvx_matcha* h, int32_t batch, int32_t t) {
    MBufs b;
    return carve_m(h->c, nullptr, batch, t, b);
}

int hvx_matcha_estimator(hvx_matcha* h, hvx_stream stream, void* ws, size_t ws_bytes, int32_t batch, int32_t t_len, const float* x, const float* mu,
                         const float* spks, int32_t spk_dim, const float* cond, const int32_t* lens, const float* t, float* out) {
    if (!h || !ws || !x || !mu || !t || !out) return set_error("hvx_matcha_estimator: null argument"), -1;
    MBufs b;
    if (carve_m(h->c, (char*)ws, batch, t_len, b) > ws_bytes) return set_error("hvx_matcha_estimator: workspace too small"), -1;
    if (lens && !h->c.cv_variant)
        return set_error("hvx_matcha_estimator: padded batches need the key-padding variant (the Matcha decoder adds its 0/1 mask to the scores, "
                         "decoder.py:399-401: only full-length masks are served)"), -1;
    hipStream_t s = (hipStream_t)stream;
    HVX_CHECK(estimator_core(h, s, b, batch, t_len, x, mu, spks, spk_dim, cond, lens, t, b.outrow));
    const int mel = h->c.out_channels, mel_p = pad32(mel);
    for (int bb = 0; bb < batch; ++bb)       // rows [T][mel_pad] -> (mel, T)
        HVX_CHECK(launch_transpose_f32(b.outrow + (size_t)bb * t_len * mel_p, out + (size_t)bb * mel * t_len, t_len, mel, mel_p, t_len, s));
    return 0;
}

/* BASECFM.solve_euler: x (B, mel, T) holds z on entry and the sample on return; ts / dts are the fp32 values the reference's loop visits */
int hvx_matcha_solve(hvx_matcha* h, hvx_stream stream, void* ws, size_t ws_bytes, int32_t batch, int32_t t_len, float* x, const float* mu,
                     const float* spks, int32_t spk_dim, const float* cond, const int32_t* lens, int32_t n_steps, const float* ts, const float* dts) {
    if (!h || !ws || !x || !mu || !ts || !dts) return set_error("hvx_matcha_solve: null argument"), -1;
    if (batch > 64) return set_error("hvx_matcha_solve: batch %d > 64", batch), -1;
    MBufs b;
    if (carve_m(h->c, (char*)ws, batch, t_len, b) > ws_bytes) return set_error("hvx_matcha_solve: workspace too small"), -1;
    if (lens && !h->c.cv_variant) return set_error("hvx_matcha_solve: padded batches need the key-padding variant"), -1;
    hipStream_t s = (hipStream_t)stream;
    const int mel_p = pad32(h->c.out_channels);
    for (int i = 0; i < n_steps; ++i) {
        hipLaunchKernelGGL(fill_f32_kernel, dim3(1), dim3(64), 0, s, b.t_dev, batch, ts[i]);
        HVX_CHECK(estimator_core(h, s, b, batch, t_len, x, mu, spks, spk_dim, cond, lens, b.t_dev, b.outrow));
        HVX_CHECK(launch_euler_rows(x, b.outrow, mel_p, dts[i], batch, t_len, h->c.out_channels, s));
    }
    return 0;
}

int hvx_hifigan_create(const hvx_hifigan_config* cfg, const void* const* weights, int32_t n_weights, hvx_hifigan** out) {
    if (!cfg || !weights || !out) return set_error("hvx_hifigan_create: null argument"), -1;
    if (cfg->n_up < 1 || cfg->n_up > 4 || cfg->n_rb < 1 || cfg->n_rb > 4 || (cfg->initial_channel >> cfg->n_up) % 32 || cfg->initial_channel % (1 << cfg->n_up))
        return set_error("hvx_hifigan_create: unsupported geometry (every stage width must be a multiple of 32)"), -1;
    const int expect = gen_expected_weights(*cfg);
    if (n_weights != expect) return set_error("hvx_hifigan_create: expected %d weight pointers, got %d", expect, n_weights), -1;
    for (int i = 0; i < n_weights; ++i)
        if (!weights[i]) return set_error("hvx_hifigan_create: weight %d is null", i), -1;
    hvx_hifigan* h = new hvx_hifigan();
    h->c = *cfg;
    h->w.assign(weights, weights + n_weights);
    *out = h;
    return 0;
}
void hvx_hifigan_destroy(hvx_hifigan* h) { delete h; }
size_t hvx_hifigan_workspace_bytes(const hvx_hifigan* h, int32_t t) {
    GBufs b;
    return carve_g(h->c, nullptr, t, b);
}
/* Generator.forward: mel f32 (mel, T) -> wav f32 [T * prod(up_rates)] */
int hvx_hifigan_forward(hvx_hifigan* h, hvx_stream stream, void* ws, size_t ws_bytes, const float* mel, int32_t T, float* wav) {
    if (!h || !ws || !mel || !wav) return set_error("hvx_hifigan_forward: null argument"), -1;
    GBufs b;
    if (carve_g(h->c, (char*)ws, T, b) > ws_bytes) return set_error("hvx_hifigan_forward: workspace too small"), -1;
    return generator_core(h, (hipStream_t)stream, b, mel, T, wav);
}

/* Denoiser.forward on one waveform of L samples: n_fft / hop STFT (center, reflect), |S| - strength * bias clamped at 0, inverse STFT with
 * the original phase.  stft_basis [2*bins][n_fft] (window folded in), istft_basis [n_fft][pad32(2*bins)] (window and 1/N folded in),
 * wsq [n_fft] = window^2, bias [bins].  Workspace: hvx_denoise_workspace_bytes(L).  out: hop * (frames - 1) samples, frames = 1 + L / hop. */
size_t hvx_denoise_workspace_bytes(int32_t L, int32_t n_fft, int32_t hop) {
    const size_t frames = 1 + (size_t)L / hop, bins2 = (size_t)pad32(n_fft + 2);
    return align_up(((size_t)L + n_fft + 64) * 4) + align_up(frames * bins2 * 4) + align_up(frames * n_fft * 4);
}
/* |STFT| of every frame (torch.stft, center / reflect): mag f32 [frames][bins]; the Denoiser's bias spectrum is frame 0 of the vocoder's
 * answer to a zero mel (denoiser.py:49-55) */
int hvx_stft_magnitude(hvx_stream stream, void* ws, size_t ws_bytes, const float* audio, int32_t L, int32_t n_fft, int32_t hop,
                       const float* stft_basis, float* mag) {
    if (!ws || !audio || !stft_basis || !mag) return set_error("hvx_stft_magnitude: null argument"), -1;
    if (n_fft % 32 || hop % 32 || n_fft % hop || L <= n_fft / 2) return set_error("hvx_stft_magnitude: n_fft=%d hop=%d L=%d", n_fft, hop, L), -1;
    if (hvx_denoise_workspace_bytes(L, n_fft, hop) > ws_bytes) return set_error("hvx_stft_magnitude: workspace too small"), -1;
    hipStream_t s = (hipStream_t)stream;
    const int bins = n_fft / 2 + 1, ldspec = pad32(2 * bins), frames = 1 + L / hop, padded = L + n_fft;
    Carve cv((char*)ws);
    float* xp = cv.take((size_t)padded + 64);
    float* spec = cv.take((size_t)frames * ldspec);
    HVX_CHECK(launch_reflect_pad(audio, xp, L, n_fft / 2, padded + 32, s));
    GemmArgs g = conv(frames, 2 * bins, n_fft / 32, 32, xp, 32, (padded + 31) / 32, stft_basis, nullptr);
    g.conv_stride = hop / 32;
    g.out = spec; g.ldo = ldspec; g.out_cols = ldspec;
    HVX_CHECK(launch_gemm(g, s));
    return launch_spectral_magnitude(spec, ldspec, frames, bins, mag, bins, 0.0f, s);
}

/* mel_spectrogram (matcha/utils/audio.py:45-82; the prompt-feature extractor of cosyvoice/cli/frontend.py:119): reflect pad by
 * (n_fft - hop) / 2, STFT(center=False, hann), sqrt(re^2 + im^2 + 1e-9), mel filterbank, log(clamp(., 1e-5)).
 * mel_basis f32 [n_mels][pad32(n_fft/2 + 1)] (zero padded); out f32 (n_mels, frames), frames = (L + (n_fft - hop)/2*2 - n_fft) / hop + 1. */
size_t hvx_frame_features_workspace_bytes(int32_t L, const hvx_feature_config* c) {
    if (!c || c->hop <= 0) return 0;
    const size_t padded = (size_t)L + 2 * (size_t)c->reflect_pad + c->frame_len + 64, frames = (size_t)c->n_frames;
    return align_up(padded * 4) + align_up(frames * pad32(2 * c->bins) * 4) + align_up(frames * pad32(c->bins) * 4) + align_up(frames * pad32(c->n_mels) * 4);
}
int hvx_frame_features(hvx_stream stream, void* ws, size_t ws_bytes, const float* audio, int32_t L, const hvx_feature_config* c, const float* basis,
                       const float* mel_basis, float* out) {
    if (!ws || !audio || !c || !basis || !mel_basis || !out) return set_error("hvx_frame_features: null argument"), -1;
    if (c->frame_len % 32 || c->hop % 32 || c->hop <= 0 || c->frame_len <= 0 || c->reflect_pad < 0 || c->reflect_pad >= L || c->n_frames < 1 ||
        c->bins < 1 || c->n_mels < 1)
        return set_error("hvx_frame_features: frame_len=%d hop=%d reflect_pad=%d frames=%d L=%d", c->frame_len, c->hop, c->reflect_pad, c->n_frames, L), -1;
    const long long padded = (long long)L + 2 * c->reflect_pad;
    // every frame must start inside the (padded) signal; samples past its end are zero (the caller's basis gives them zero weight anyway)
    if ((long long)(c->n_frames - 1) * c->hop >= padded) return set_error("hvx_frame_features: %d frames do not fit %lld samples", c->n_frames, padded), -1;
    if (hvx_frame_features_workspace_bytes(L, c) > ws_bytes) return set_error("hvx_frame_features: workspace too small"), -1;
    hipStream_t s = (hipStream_t)stream;
    const int frames = c->n_frames, bins = c->bins, ldspec = pad32(2 * bins), ldmag = pad32(bins), ldm = pad32(c->n_mels);
    const long long total = (long long)(frames - 1) * c->hop + c->frame_len;         // samples the framing GEMM touches
    Carve cv((char*)ws);
    float* xp = cv.take((size_t)padded + c->frame_len + 64);
    float* spec = cv.take((size_t)frames * ldspec);
    float* mag = cv.take((size_t)frames * ldmag);
    float* mel = cv.take((size_t)frames * ldm);
    const long long fill = ((total > padded ? total : padded) + 31) / 32 * 32;
    HVX_CHECK(launch_reflect_pad(audio, xp, L, c->reflect_pad, (int)padded, s));
    if (fill > padded) HIP_OK(hipMemsetAsync(xp + padded, 0, (size_t)(fill - padded) * 4, s));
    GemmArgs g = conv(frames, 2 * bins, c->frame_len / 32, 32, xp, 32, (int)(fill / 32), basis, nullptr);
    g.conv_stride = c->hop / 32;
    g.out = spec; g.ldo = ldspec; g.out_cols = ldspec;
    HVX_CHECK(launch_gemm(g, s));
    HVX_CHECK(launch_spectral_magnitude(spec, ldspec, frames, bins, mag, ldmag, c->power ? -1.0f : c->mag_eps, s));
    g = conv(frames, c->n_mels, 1, ldmag, mag, ldmag, frames, mel_basis, nullptr);
    g.act = ACT_LOG_CLAMP; g.act_param = c->log_floor; g.scale = c->log_scale;
    g.out = mel; g.ldo = ldm; g.out_cols = ldm;
    HVX_CHECK(launch_gemm(g, s));
    HVX_CHECK(launch_feature_post(mel, ldm, frames, c->n_mels, c->post, s));
    if (c->time_major) {
        HIP_OK(hipMemcpy2DAsync(out, (size_t)c->n_mels * 4, mel, (size_t)ldm * 4, (size_t)c->n_mels * 4, frames, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    return launch_transpose_f32(mel, out, frames, c->n_mels, ldm, frames, s);
}

size_t hvx_mel_workspace_bytes(int32_t L, int32_t n_fft, int32_t hop, int32_t n_mels) {
    if (hop <= 0 || n_fft < hop) return 0;
    const int pad = (n_fft - hop) / 2;
    hvx_feature_config c;
    memset(&c, 0, sizeof(c));
    c.frame_len = n_fft; c.hop = hop; c.reflect_pad = pad; c.n_frames = (L + 2 * pad - n_fft) / hop + 1; c.bins = n_fft / 2 + 1; c.n_mels = n_mels;
    return c.n_frames < 1 ? 0 : hvx_frame_features_workspace_bytes(L, &c);
}
int hvx_mel_spectrogram(hvx_stream stream, void* ws, size_t ws_bytes, const float* audio, int32_t L, int32_t n_fft, int32_t hop,
                        const float* stft_basis, const float* mel_basis, int32_t n_mels, float* out) {
    if (!ws || !audio || !stft_basis || !mel_basis || !out) return set_error("hvx_mel_spectrogram: null argument"), -1;
    const int pad = (n_fft - hop) / 2;
    if (n_fft % 32 || hop % 32 || hop > n_fft || L <= pad) return set_error("hvx_mel_spectrogram: n_fft=%d hop=%d L=%d", n_fft, hop, L), -1;
    const int frames = (L + 2 * pad - n_fft) / hop + 1;
    if (frames < 1) return set_error("hvx_mel_spectrogram: signal shorter than one frame"), -1;
    // matcha/utils/audio.py:45-82: reflect pad (n_fft - hop) / 2, hann STFT, sqrt(|X|^2 + 1e-9), mel, log(clamp(., 1e-5))
    hvx_feature_config c;
    memset(&c, 0, sizeof(c));
    c.frame_len = n_fft; c.hop = hop; c.reflect_pad = pad; c.n_frames = frames; c.bins = n_fft / 2 + 1; c.power = 0; c.mag_eps = 1e-9f;
    c.n_mels = n_mels; c.log_floor = 1e-5f; c.log_scale = 1.0f; c.post = 0; c.time_major = 0;
    return hvx_frame_features(stream, ws, ws_bytes, audio, L, &c, stft_basis, mel_basis, out);
}

int hvx_denoise(hvx_stream stream, void* ws, size_t ws_bytes, const float* audio, int32_t L, int32_t n_fft, int32_t hop, const float* stft_basis,
                const float* istft_basis, const float* wsq, const float* bias, float strength, float* out) {
    if (!ws || !audio || !stft_basis || !istft_basis || !wsq || !bias || !out) return set_error("hvx_denoise: null argument"), -1;
    if (n_fft % 32 || hop % 32 || n_fft % hop || L <= n_fft / 2) return set_error("hvx_denoise: n_fft=%d hop=%d L=%d", n_fft, hop, L), -1;
    if (hvx_denoise_workspace_bytes(L, n_fft, hop) > ws_bytes) return set_error("hvx_denoise: workspace too small"), -1;
    hipStream_t s = (hipStream_t)stream;
    const int bins = n_fft / 2 + 1, ldspec = pad32(2 * bins), frames = 1 + L / hop;
    const int padded = L + n_fft;
    Carve cv((char*)ws);
    float* xp = cv.take((size_t)padded + 64);
    float* spec = cv.take((size_t)frames * ldspec);
    float* fr = cv.take((size_t)frames * n_fft);
    HVX_CHECK(launch_reflect_pad(audio, xp, L, n_fft / 2, padded + 32, s));
    // frame f = samples [f*hop, f*hop + n_fft): a conv over the padded signal seen as rows of 32 samples (n_fft/32 taps, stride hop/32)
    GemmArgs g = conv(frames, 2 * bins, n_fft / 32, 32, xp, 32, (padded + 31) / 32, stft_basis, nullptr);
    g.conv_stride = hop / 32;
    g.out = spec; g.ldo = ldspec; g.out_cols = ldspec;
    HVX_CHECK(launch_gemm(g, s));
    HVX_CHECK(launch_spectral_subtract(spec, ldspec, frames, bins, bias, strength, s));
    g = conv(frames, n_fft, 1, ldspec, spec, ldspec, frames, istft_basis, nullptr);
    g.out = fr; g.ldo = n_fft; g.out_cols = n_fft;
    HVX_CHECK(launch_gemm(g, s));
    return launch_overlap_add(fr, n_fft, frames, n_fft, hop, wsq, out, hop * (frames - 1), s);
}

}  // extern "C"
