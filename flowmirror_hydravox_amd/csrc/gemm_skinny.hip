// gemm_skinny.hip — weight-streaming GEMM for the AR decode step (see hvx_kernels.h: SkinnyArgs).
//
// The decode step is HBM-bound on weight bytes: M = batch*K_heads rows (16..128)
// against 0.7-1.1 GB of weights.  Each wave owns NT consecutive 16-column tiles of the output and a
// K-slice; it streams its weight fragments straight from HBM into VGPRs — the checkpoint is
// re-packed at load time into MFMA fragment order [N/16][K/32][64 lanes][8], so every wave-level
// load instruction is one fully coalesced 1 KiB (bf16) burst — and re-reads the small, L2-resident
// activation rows as A fragments.  The KW waves of a workgroup split K among themselves and meet once, in
// LDS, where wave 0 adds the partial tiles in fixed order (deterministic, no atomics) and runs the fused
// epilogue: bias + RoPE + KV-cache scatter, SwiGLU, or the in-place residual update.  Split-K across
// workgroups (blockIdx.y, fp32 partials reduced by reduce_rmsnorm) remains for the MTP-head GEMMs.
#include <stdlib.h>

#include <type_traits>

#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

// per-row operands of the QKV epilogue (position, cache slot, rotary factors); loaded before the weight stream when MT == 1
struct RowCtl {
    int ok, pos, slot;
    float cs, sn;
};
__device__ __forceinline__ RowCtl load_rowctl(const SkinnyArgs& a, int row, int f, int which) {
    RowCtl c = {0, 0, 0, 1.0f, 0.0f};
    if (row >= a.M) return c;
    const int si = row / a.kn, lt = row - si * a.kn;
    if (lt >= a.n_new[si]) return c;                            // inactive row
    c.ok = 1;
    c.pos = a.pos0[si] + lt;
    c.slot = a.slot[si];
    if (which < 2) {
        c.cs = a.rope_cos[(long long)c.pos * 32 + f];
        c.sn = a.rope_sin[(long long)c.pos * 32 + f];
    }
    return c;
}

// ---- epilogues of the decode GEMMs, shared by the skinny (M <= 32 per chunk) and the mid (33..128 rows) kernels ------------------------
// acc: MT x NT accumulator tiles in the MFMA C layout, first row m0, first 16-column tile ntile0; HOIST: e_res / e_rc were fetched
// before the weight stream (single-tile decode geometry), otherwise they are loaded here.
template <class T, int MT, int NT, int EPI, bool HOIST>
__device__ __forceinline__ void skinny_finish(const SkinnyArgs& a, f32x4 (&acc)[MT][NT], int m0, int ntile0, int z, int ks, int lane,
                                              const float (&e_bias)[NT], const float (&e_res)[MT][NT][4], const RowCtl (&e_rc)[MT][4],
                                              int q_which, int q_hh, int q_d, int q_f) {
    const int fr = lane & 15, fg = lane >> 4;
    // ---- epilogues ---------------------------------------------------------------------------------
    if constexpr (EPI == SK_PARTIAL) {
        float* part = a.part + (long long)z * a.part_zs + (long long)ks * a.M * a.N;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = (ntile0 + j) * 16 + fr;
            if (col >= a.N) continue;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + i * 16 + fg * 4 + r;
                    if (row < a.M) part[(long long)row * a.N + col] = acc[i][j][r];
                }
        }
    } else if constexpr (EPI == SK_RESID) {
        // residual stream update in place: x[row][col] += acc (+ bias).  split_k == 1, so every element has exactly one writer.
        float* xo = reinterpret_cast<float*>(a.out) + (long long)z * a.out_zs;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = (ntile0 + j) * 16 + fr;
            if (col >= a.N) continue;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + i * 16 + fg * 4 + r;
                    if (row >= a.M) continue;
                    float* px = xo + (long long)row * a.ldo + col;
                    const float x0 = HOIST ? e_res[i][j][r] : *px;
                    const float x1 = x0 + (acc[i][j][r] + e_bias[j]);
                    *px = x1;
                    if (a.out2) reinterpret_cast<T*>(a.out2)[(long long)z * a.out_zs + (long long)row * a.ldo2 + col] = from_f32<T>(x1);
                }
        }
    } else if constexpr (EPI == SK_STORE) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = (ntile0 + j) * 16 + fr;
            if (col >= (a.n_valid ? a.n_valid : a.N)) continue;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + i * 16 + fg * 4 + r;
                    if (row >= a.M) continue;
                    const float v = acc[i][j][r] + e_bias[j];
                    const long long o = (long long)z * a.out_zs + (long long)row * a.ldo + col;
                    if (a.out_f32) reinterpret_cast<float*>(a.out)[o] = v;
                    else reinterpret_cast<T*>(a.out)[o] = from_f32<T>(v);
                }
        }
    } else if constexpr (EPI == SK_SWIGLU) {
        // tiles come in (gate, up) pairs: tile 2p holds gate columns 16p..16p+15, tile 2p+1 the matching up columns
        static_assert(EPI != SK_SWIGLU || NT % 2 == 0, "SwiGLU needs tile pairs");
#pragma unroll
        for (int j = 0; j + 1 < NT; j += 2) {
            const int col = ((ntile0 + j) >> 1) * 16 + fr;
            if (col * 2 >= a.N) continue;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + i * 16 + fg * 4 + r;
                    if (row >= a.M) continue;
                    const float gte = acc[i][j][r], up = acc[i][j + 1][r];
                    const float v = gte * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gte)) * up;   // silu(gate) * up
                    reinterpret_cast<T*>(a.out)[(long long)z * a.out_zs + (long long)row * a.ldo + col] = from_f32<T>(v);
                }
        }
    } else {   // SK_QKV_ROPE
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + i * 16 + fg * 4 + r;
                float x = acc[i][0][r] + e_bias[0];
                const float partner = __shfl_xor(x, 8, 64);         // same row (same fg), the other half of the head
                const RowCtl c = HOIST ? e_rc[i][r] : load_rowctl(a, row, q_f, q_which);
                if (!c.ok) continue;
                if (q_which < 2)                                    // rotate-half RoPE (HF Qwen2): d pairs with d +- 32
                    x = (fr < 8) ? (x * c.cs - partner * c.sn) : (x * c.cs + partner * c.sn);
                if (q_which == 0) {
                    reinterpret_cast<T*>(a.qbuf)[((long long)row * a.q_heads + q_hh) * 64 + q_d] = from_f32<T>(x);
                } else if (q_which == 1) {
                    const long long blk = ((long long)c.slot * a.kv_heads + q_hh) * a.max_ctx * 64;
                    reinterpret_cast<T*>(a.kcache)[blk + (a.kv_frag ? frag_index(c.pos, q_d, 2) : (long long)c.pos * 64 + q_d)] = from_f32<T>(x);
                } else {
                    const long long blk = ((long long)c.slot * a.kv_heads + q_hh) * a.max_ctx * 64;
                    reinterpret_cast<T*>(a.vTcache)[blk + (a.kv_frag ? vfrag_index(c.pos, q_d) : (long long)q_d * a.max_ctx + c.pos)] = from_f32<T>(x);
                }
            }
        }
    }
}

// ANORM == 1: A is the residual stream x (in the operand type: the producing epilogue keeps a T copy beside the fp32 stream) and the
// GEMM computes RMSNorm(x) @ W^T without a norm kernel in front of it.  RMSNorm is a per-row scale:
// norm(x)[k] = gain[k] * x[k] * rsqrt(mean(x^2) + eps).  The gain is folded into the weight columns when the checkpoint is packed
// (llm.py), the sum of squares is accumulated from the very fragments the MFMA consumes (every wave sees its K slice of the 16
// rows), and the epilogue multiplies the accumulator by rsqrt(ss / K + eps): no extra loads, no extra pass, nothing in front of
// the weight stream.
template <class T, int MT, int NT, int EPI, int KW, int ANORM, int U>
__global__ __launch_bounds__(64 * KW) void gemm_skinny_kernel(SkinnyArgs a) {
    typedef typename Vec8<T>::type V8;
    constexpr bool HOIST = MT == 1;      // decode geometry: epilogue operands are fetched ahead of the weight stream
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int z = blockIdx.z % a.nz;
    const int mchunk = blockIdx.z / a.nz;
    const int m0 = mchunk * (MT * 16);
    const int ntile0 = blockIdx.x * NT;                        // the KW waves of a workgroup share these NT tiles ...
    const int KT = a.K >> 5;
    const int ks = blockIdx.y;
    const int kt_per = (KT + a.split_k - 1) / a.split_k;
    const int kb0 = ks * kt_per;
    const int kb1 = min(KT, kb0 + kt_per);
    const int kq = (kb1 - kb0 + KW - 1) / KW;                  // ... and split its K range among them (in-block split-K)
    const int kt0 = kb0 + wave * kq;
    const int kt1 = min(kb1, kt0 + kq);

    const T* __restrict__ W = reinterpret_cast<const T*>(a.W) + (long long)z * a.w_zs;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    // A fragment rows (clamped; masked at the store)
    const T* arow[MT];
    float ssq[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int r = m0 + i * 16 + fr;
        r = r < a.M ? r : a.M - 1;
        arow[i] = reinterpret_cast<const T*>(a.A) + (long long)z * a.a_zs + (long long)r * a.lda + fg * 8;
        ssq[i] = 0.0f;
    }
    const T* wtile[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int nt = ntile0 + j;
        nt = (nt * 16 < a.N) ? nt : (a.N / 16 - 1);
        wtile[j] = W + ((long long)nt * KT) * 512 + lane * 8;
    }

    // ---- epilogue operands of the finishing wave, requested before the weight stream so their latency hides under it ------------
    float e_bias[NT];
    float e_res[MT][NT][4];
    RowCtl e_rc[MT][4];
    int q_which = 0, q_hh = 0, q_d = 0, q_f = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) e_bias[j] = 0.0f;
    if constexpr (EPI == SK_QKV_ROPE) {
        // The checkpoint rows of every 64-wide head are permuted at pack time (packing.qkv_row_perm) so that tile t of a
        // head holds d = 8t..8t+7 (lanes fr < 8) and their rotate-half partners d + 32 (lanes fr >= 8): the RoPE pair is one
        // xor-8 shuffle inside a single 16-column tile, and the GEMM can use one tile per workgroup (4x the workgroups).
        static_assert(EPI != SK_QKV_ROPE || NT == 1, "QKV epilogue works on single permuted tiles");
        const int head = ntile0 >> 2, t = ntile0 & 3;       // global head index over [q heads | k heads | v heads]
        q_which = head < a.q_heads ? 0 : (head < a.q_heads + a.kv_heads ? 1 : 2);
        q_hh = q_which == 0 ? head : (q_which == 1 ? head - a.q_heads : head - a.q_heads - a.kv_heads);
        q_d = (fr < 8) ? (8 * t + fr) : (32 + 8 * t + (fr - 8));
        q_f = 8 * t + (fr & 7);                             // rotary frequency index (d mod 32)
    }
    constexpr bool SPREAD = (MT == KW) && MT > 1;     // every wave finishes one of the MT row tiles (see the reduction below)
    if (wave == 0 || SPREAD) {
        if constexpr (EPI == SK_RESID || EPI == SK_STORE || EPI == SK_QKV_ROPE) {
            const float* bias = a.bias ? a.bias + (long long)z * a.bias_zs : nullptr;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = (ntile0 + j) * 16 + fr;
                e_bias[j] = (bias && col < a.N) ? bias[col] : 0.0f;
            }
        }
        if constexpr (EPI == SK_RESID && HOIST) {
            const float* xo = reinterpret_cast<const float*>(a.out) + (long long)z * a.out_zs;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + i * 16 + fg * 4 + r, col = (ntile0 + j) * 16 + fr;
                        e_res[i][j][r] = (row < a.M && col < a.N) ? xo[(long long)row * a.ldo + col] : 0.0f;
                    }
        }
        if constexpr (EPI == SK_QKV_ROPE && HOIST) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) e_rc[i][r] = load_rowctl(a, m0 + i * 16 + fg * 4 + r, q_f, q_which);
        }
    }

    // U k-steps per trip: all of their weight / activation loads are issued before the first MFMA so that each wave keeps
    // U * NT KiB of the weight stream in flight (a dependent load->MFMA chain per k-step is latency-bound: 0.6 TB/s measured).
    // The decode shapes are dispatched so that a wave's whole K slice is one trip.
    for (int kt = kt0; kt < kt1; kt += U) {
        V8 wf[U][NT], af[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(kt + u, kt1 - 1);
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[u][j] = load8_nt(wtile[j] + (long long)k * 512);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[u][i] = load8(arow[i] + k * 32);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kt + u < kt1) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if constexpr (ANORM) {
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = to_f32(af[u][i][e]);
                        ssq[i] += ((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3])) + ((x[4] * x[4] + x[5] * x[5]) + (x[6] * x[6] + x[7] * x[7]));
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j) mma32(acc[i][j], af[u][i], wf[u][j]);
                }
            }
        }
    }

    // in-block reduction in fixed wave order (deterministic): waves 1.. park their tiles (and row sums of squares) in LDS,
    // wave 0 adds them and finishes.  SPREAD (as many row tiles as waves: 64-row chunks of a wide decode grid): every wave parks its
    // partials and wave w adds — in the same fixed order 0, 1, .. — and finishes row tile w, so the epilogue (16 elements per lane of
    // scattered loads / stores, the RoPE controls) runs on all waves at once instead of on wave 0 for all 64 rows.
    __shared__ f32x4 red[SPREAD ? KW : (KW > 1 ? KW - 1 : 1)][MT][NT][64];
    __shared__ float s_ss[ANORM ? KW : 1][MT * 16];
    if constexpr (SPREAD) {
        if constexpr (ANORM) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float v = ssq[i];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (fg == 0) s_ss[wave][i * 16 + fr] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) red[wave][i][j][lane] = acc[i][j];
        __syncthreads();
        f32x4 fin[1][NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            fin[0][j] = red[0][wave][j][lane];
#pragma unroll
            for (int w = 1; w < KW; ++w) fin[0][j] += red[w][wave][j][lane];
        }
        if constexpr (ANORM) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float ss = 0.0f;
#pragma unroll
                for (int w = 0; w < KW; ++w) ss += s_ss[w][wave * 16 + fg * 4 + r];
                const float inv = rsqrtf(ss / (float)a.K + a.norm_eps);
#pragma unroll
                for (int j = 0; j < NT; ++j) fin[0][j][r] *= inv;
            }
        }
        float d_res[1][NT][4];
        RowCtl d_rc[1][4];
        skinny_finish<T, 1, NT, EPI, false>(a, fin, m0 + wave * 16, ntile0, z, ks, lane, e_bias, d_res, d_rc, q_which, q_hh, q_d, q_f);
        return;
    }
    if constexpr (ANORM) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float v = ssq[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (fg == 0) s_ss[wave][i * 16 + fr] = v;
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) red[wave - 1][i][j][lane] = acc[i][j];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < KW - 1; ++w)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] += red[w][i][j][lane];
    if constexpr (ANORM) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float ss = 0.0f;
#pragma unroll
                for (int w = 0; w < KW; ++w) ss += s_ss[w][i * 16 + fg * 4 + r];
                const float inv = rsqrtf(ss / (float)a.K + a.norm_eps);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j][r] *= inv;
            }
    }

    skinny_finish<T, MT, NT, EPI, HOIST>(a, acc, m0, ntile0, z, ks, lane, e_bias, e_res, e_rc, q_which, q_hh, q_d, q_f);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Mid form: 33..128 rows — the decode grid of continuous batching (17..64 sequences x 2 heads).  The skinny kernel above serves such a
// grid in 64-row chunks whose waves fetch their activation fragments straight from L2 (16 rows x 64 B per load instruction) and re-read the
// weights per chunk: at 128 rows SwiGLU takes 24 us and down_proj 19 + 5 us for 3.8 GF and 30 MB of weights per layer.  Here a workgroup
// of 4 waves owns ALL rows (8 row tiles) of 4 x NT column tiles: the activation K-tiles (128 rows x 64) go through LDS once per workgroup
// (full 128-byte rows, double-buffered, padded against bank conflicts) and are shared by the four waves, each wave streams the
// fragment-packed weights of its own NT tiles straight into registers one K-tile ahead, and every weight fragment feeds 8 MFMAs.  No
// in-block K split, so no LDS reduction; split-K across workgroups (fp32 partials) remains for the long-K down projection.
template <class T, int NT, int EPI, int ANORM>
__global__ __launch_bounds__(256) void gemm_mid_kernel(SkinnyArgs a) {
    typedef typename Vec8<T>::type V8;
    constexpr int MT = 8, BK = 64, KS = BK / 32, ROWS = MT * 16;
    constexpr int LDK = BK + (sizeof(T) == 2 ? 8 : 4);
    __shared__ __attribute__((aligned(16))) T xs[2][ROWS * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int z = blockIdx.z, ks = blockIdx.y;
    const int ntile0 = (blockIdx.x * 4 + wave) * NT;
    const int KT = a.K >> 5;                                   // 32-wide k-steps of the packed weights
    const int nkt = a.K / BK;                                  // K-tiles
    const int per = (nkt + a.split_k - 1) / a.split_k;
    const int kc0 = ks * per, kc1 = min(nkt, kc0 + per);
    const int nk = kc1 - kc0;

    const T* __restrict__ W = reinterpret_cast<const T*>(a.W) + (long long)z * a.w_zs;
    const T* wtile[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int nt = ntile0 + j;
        nt = (nt * 16 < a.N) ? nt : (a.N / 16 - 1);
        wtile[j] = W + ((long long)nt * KT) * 512 + lane * 8;
    }
    // activation tile loader: thread t moves 4 x 8 elements: rows t/8 + 32 i, chunk t % 8
    const T* __restrict__ Ab = reinterpret_cast<const T*>(a.A) + (long long)z * a.a_zs;
    const int lrow = tid >> 3, lchunk = (tid & 7) * 8;
    const T* arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = lrow + 32 * i;
        r = r < a.M ? r : a.M - 1;
        arow[i] = Ab + (long long)r * a.lda + lchunk;
    }
    // Both operand streams run PF K-tiles ahead in registers (static ring, the loop is unrolled by PF): a workgroup is alone on its CU
    // (14..76 workgroups per launch), so nothing else hides the ~2 us of an HBM round trip — with one tile of look-ahead every K-tile
    // cost that round trip (28 us for the 14 K-tiles of the QKV projection).
    constexpr int PF = 4;
    V8 xr[PF][4], wr[PF][NT][KS];
    auto load_x = [&](int kc, V8 (&dst)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = load8(arow[i] + (long long)kc * BK);
    };
    auto stash_x = [&](int buf, const V8 (&src)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) store8(&xs[buf][(lrow + 32 * i) * LDK + lchunk], src[i]);
    };
    auto load_w = [&](int kc, V8 (&dst)[NT][KS]) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) dst[j][kk] = load8_nt(wtile[j] + (long long)(kc * KS + kk) * 512);
    };

    f32x4 acc[MT][NT];
    float ssq[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        ssq[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    }
    // epilogue operands (bias, the residual rows this lane will update, per-row position / cache slot / rotary factors): requested before
    // the operand streams so that their latency — 32 rows per lane, dependent loads for the QKV controls — hides under the K loop
    float e_bias[NT];
    float e_res[MT][NT][4];
    RowCtl e_rc[MT][4];
    int q_which = 0, q_hh = 0, q_d = 0, q_f = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) e_bias[j] = 0.0f;
    if constexpr (EPI == SK_QKV_ROPE) {
        static_assert(EPI != SK_QKV_ROPE || NT == 1, "QKV epilogue works on single permuted tiles");
        const int head = ntile0 >> 2, tq = ntile0 & 3;
        q_which = head < a.q_heads ? 0 : (head < a.q_heads + a.kv_heads ? 1 : 2);
        q_hh = q_which == 0 ? head : (q_which == 1 ? head - a.q_heads : head - a.q_heads - a.kv_heads);
        q_d = (fr < 8) ? (8 * tq + fr) : (32 + 8 * tq + (fr - 8));
        q_f = 8 * tq + (fr & 7);
        if (ntile0 * 16 < a.N) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) e_rc[i][r] = load_rowctl(a, i * 16 + fg * 4 + r, q_f, q_which);
        }
    }
    if constexpr (EPI == SK_RESID || EPI == SK_STORE || EPI == SK_QKV_ROPE) {
        const float* bias = a.bias ? a.bias + (long long)z * a.bias_zs : nullptr;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = (ntile0 + j) * 16 + fr;
            e_bias[j] = (bias && col < a.N) ? bias[col] : 0.0f;
        }
    }
    if constexpr (EPI == SK_RESID) {
        const float* xo = reinterpret_cast<const float*>(a.out) + (long long)z * a.out_zs;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i * 16 + fg * 4 + r, col = (ntile0 + j) * 16 + fr;
                    e_res[i][j][r] = (row < a.M && col < a.N) ? xo[(long long)row * a.ldo + col] : 0.0f;
                }
    }
    // Every load below is unconditional (tile indices are clamped to the last tile of the range instead): with a load under a runtime
    // condition hipcc cannot count the outstanding requests and waits for ALL of them before each use — which put the HBM round trip
    // back into every K-tile.  The surplus loads at the end of the range re-read a tile that is already in L2.
    const int klast = kc1 - 1;
    auto step = [&](int t, auto P, bool refill) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value;
        __syncthreads();                                       // tile t is visible; buffer (t + 1) & 1 is free
        stash_x((t + 1) & 1, xr[(p + 1) % PF]);                // tile u travels in ring slot u % PF (past the end: a duplicate nobody reads)
        if (refill) load_x(min(kc0 + t + 1 + PF, klast), xr[(p + 1) % PF]);
        const T* xb = xs[t & 1];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            V8 af[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = load8(&xb[(i * 16 + fr) * LDK + kk * 32 + fg * 8]);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (ANORM) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = to_f32(af[i][e]);
                    ssq[i] += ((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3])) + ((x[4] * x[4] + x[5] * x[5]) + (x[6] * x[6] + x[7] * x[7]));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) mma32(acc[i][j], af[i], wr[p][j][kk]);
            }
        }
        if (refill) load_w(min(kc0 + t + PF, klast), wr[p]);
    };
    if (nk > 0) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            load_w(min(kc0 + p, klast), wr[p]);
            load_x(min(kc0 + p, klast), xr[p]);
        }
        stash_x(0, xr[0]);
        load_x(min(kc0 + PF, klast), xr[0]);
        const int nfull = nk / PF * PF;
        for (int t0 = 0; t0 < nfull; t0 += PF) {
            step(t0 + 0, std::integral_constant<int, 0>{}, true);
            step(t0 + 1, std::integral_constant<int, 1>{}, true);
            step(t0 + 2, std::integral_constant<int, 2>{}, true);
            step(t0 + 3, std::integral_constant<int, 3>{}, true);
        }
        static_assert(PF == 4, "the step sequence is written out for a ring of 4");
        const int rem = nk - nfull;                            // the last 0..3 tiles are already in the ring
        if (rem > 0) step(nfull + 0, std::integral_constant<int, 0>{}, false);
        if (rem > 1) step(nfull + 1, std::integral_constant<int, 1>{}, false);
        if (rem > 2) step(nfull + 2, std::integral_constant<int, 2>{}, false);
    }
    if (ntile0 * 16 >= a.N) return;                            // (a wave past the last column tile only helped with the staging)
    if constexpr (ANORM) {
        // every wave has seen the whole K of its rows: lane (fr, fg) holds a quarter of row 16 i + fr's sum of squares
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float v = ssq[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float inv = rsqrtf(__shfl(v, fg * 4 + r, 64) / (float)a.K + a.norm_eps);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j][r] *= inv;
            }
        }
    }
    skinny_finish<T, MT, NT, EPI, true>(a, acc, 0, ntile0, z, ks, lane, e_bias, e_res, e_rc, q_which, q_hh, q_d, q_f);
}

template <class T, int NT, int EPI, int ANORM>
static int launch_mid_one(const SkinnyArgs& a, hipStream_t s) {
    const int ntiles = a.N / 16;
    dim3 grid((ntiles + 4 * NT - 1) / (4 * NT), a.split_k, a.nz);
    const double bytes = (double)a.nz * ((double)a.N * a.K * sizeof(T) + (double)a.M * a.K * sizeof(T) +
                                         (double)a.M * a.N * (EPI == SK_PARTIAL ? 4.0 * a.split_k : (double)sizeof(T)));
    const int slot = prof_begin(PK_SKINNY, bytes, s);
    hipLaunchKernelGGL((gemm_mid_kernel<T, NT, EPI, ANORM>), grid, dim3(256), 0, s, a);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("mid gemm launch failed"), -1);
}

// 33..128 rows of one matrix (nz == 1), K a multiple of 64.  Measured at 128 rows (64 sequences x 2 heads, tools/bench_decode.py, us per
// launch, mid form vs the 64-row-chunk skinny form): split-K partials of down_proj 11.2 vs 18.6; QKV + RoPE 22 vs 15, SwiGLU 32-36 vs 24,
// o_proj 15 vs 9.8 (4-column form) — with one workgroup per CU and 14..76 workgroups per launch the short-K GEMMs are bound by their
// per-K-tile barrier / LDS round trip and by the 32-rows-per-lane epilogues, not by the operand streams.  Only the long-K partial form
// is dispatched here; the other epilogues stay on the skinny kernel.  (Also measured: the skinny kernel itself with all 128 rows per
// workgroup, MT = 8: SwiGLU 33 us with one (gate, up) pair per workgroup, 22 us with two; QKV 26 us — the finishing wave's epilogue over
// 32 rows per lane and the longer dependent chains cost more than the halved weight re-reads save.)
template <class T>
static int launch_mid(const SkinnyArgs& a, hipStream_t s) {
    return launch_mid_one<T, 1, SK_PARTIAL, 0>(a, s);
}

template <class T, int MT, int NT, int EPI, int KW = 4, int ANORM = 0, int U_ = 0>
static int launch_one(const SkinnyArgs& a, hipStream_t s) {
    // k-steps in flight per wave: sized so that the fragment registers stay within ~96 VGPRs unless the caller knows better
    constexpr int REGS = (NT + MT) * (sizeof(T) == 2 ? 4 : 8);
    constexpr int U = U_ ? U_ : (REGS <= 16 ? 8 : (REGS <= 32 ? 4 : 2));
    const int ntiles = a.N / 16;
    const int groups = (ntiles + NT - 1) / NT;
    const int mchunks = (a.M + MT * 16 - 1) / (MT * 16);
    dim3 grid(groups, a.split_k, mchunks * a.nz);
    // algorithmic HBM bytes of this launch: every weight once, the activation rows once, the result once
    const double bytes = (double)a.nz * ((double)a.N * a.K * sizeof(T) + (double)a.M * a.K * sizeof(T) +
                                         (double)a.M * a.N * (EPI == SK_PARTIAL ? 4.0 * a.split_k : (double)sizeof(T)));
    const int slot = prof_begin(PK_SKINNY, bytes, s);
    hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, NT, EPI, KW, ANORM, U>), grid, dim3(64 * KW), 0, s, a);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("skinny gemm launch failed"), -1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Narrow form for the two residual projections of the decode step (o_proj, down_proj: N = hidden, one 16-row activation tile).
// A CU pulls ~45 GB/s out of HBM however many waves ask (tools/stream_probe.hip), so a weight matrix has to be spread over ALL CUs:
// with 16-column tiles N = 896 gives only 56 workgroups, and each of them also re-reads the whole [16][K] activation tile.  Here a
// workgroup owns FOUR output columns over the full K (N/4 = 224 workgroups) and still feeds the 16x16x32 MFMA: the 16 tile columns
// are (4 output columns) x (4 consecutive 32-wide K sub-blocks), i.e. lane (c16, g) holds W[4*nt + (c16 & 3)][128*kb + 32*(c16 >> 2)
// + 8g ..+8] — one fully coalesced 1 KiB load per 128-wide K block (checkpoint packed as [N/4][K/128][64][8], packing.pack_narrow4).
// MFMA number s of a block multiplies the activation sub-block s with the weight fragment masked to the lanes whose sub-block is s,
// so tile column (c, s) accumulates exactly its own K quarter; two xor-shuffles add the four quarters at the end.  The matrix cores
// do 4x redundant work (free: the kernel is bound by bytes per CU), the K split stays inside the workgroup (LDS, fixed order), and
// the residual update x += ... has one writer per element.
// ---------------------------------------------------------------------------------------------------------------------------------
// COMB: the two A fragments (d = 8 fg .. + 8 and 32 + 8 fg .. + 8) of query head hq for grid row `row`, combined from the key-split partials
// of the attention launch in front — operation for operation what attn_combine_kernel computes and rounds (attention.hip), so the fused
// and the two-launch forms feed the MFMA the same bits.
template <class T>
__device__ __forceinline__ void combined_pair(const SkinnyArgs& a, int row, int hq, int fg, typename Vec8<T>::type& f_lo, typename Vec8<T>::type& f_hi) {
    const int si = row / a.kn, lt = row - si * a.kn;
    const int G = a.q_heads / a.kv_heads;
    const int h = hq / G, rh = hq - h * G;
    const int vis = min(a.att_kvlen[si], a.pos0[si] + lt + 1);
    const int n_live = lt < a.n_new[si] ? min(a.att_splits, (vis + a.att_chunk - 1) / a.att_chunk) : 0;
    const long long base0 = (((long long)si * a.kv_heads + h) * a.att_splits) * a.att_rows_pad + rh * a.kn + lt;
    const long long sstride = a.att_rows_pad;
    constexpr int SB = 8;
    float m = -INFINITY, l = 0.0f, acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    for (int s0 = 0; s0 < n_live; s0 += SB) {
        float ms[SB], ls[SB];
        f32x4 os[SB][4];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            // (the combine kernel re-reads the last live split for the dead slots of a batch and weighs it with 0; here a dead slot is 0 itself and costs
            // no load: 0 * finite == 0 * 0 in every accumulator)
            ms[u] = -INFINITY;
            ls[u] = 0.0f;
            os[u][0] = os[u][1] = os[u][2] = os[u][3] = f32x4{0, 0, 0, 0};
            if (s0 + u < n_live) {
                const long long base = base0 + (long long)(s0 + u) * sstride;
                const float2 ml = *reinterpret_cast<const float2*>(a.att_ml + base * 2);
                ms[u] = ml.x;
                ls[u] = ml.y;
                const f32x4* po = reinterpret_cast<const f32x4*>(a.att_o + base * 64 + fg * 8);
                os[u][0] = po[0]; os[u][1] = po[1]; os[u][2] = po[8]; os[u][3] = po[9];
            }
        }
        float mb = m;
#pragma unroll
        for (int u = 0; u < SB; ++u) mb = fmaxf(mb, ms[u]);
        const float resc = (m == -INFINITY) ? 0.0f : expf(m - mb);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] *= resc;
        l *= resc;
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const float wgt = (ms[u] == -INFINITY) ? 0.0f : expf(ms[u] - mb);
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += wgt * os[u][e >> 2][e & 3];
            l += wgt * ls[u];
        }
        m = mb;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        f_lo[e] = from_f32<T>(l > 0.0f ? acc[e] / l : 0.0f);
        f_hi[e] = from_f32<T>(l > 0.0f ? acc[8 + e] / l : 0.0f);
    }
}

template <class T, int KW, int UB, bool COMB = false>
__global__ __launch_bounds__(64 * KW) void gemm_narrow_resid_kernel(SkinnyArgs a) {
    typedef typename Vec8<T>::type V8;
    static_assert(!COMB || UB == 1, "the combining form requests one K block per wave");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, fg = lane >> 4;
    const int c4 = c16 & 3, ksub = c16 >> 2;
    const int nt = blockIdx.x, m0 = blockIdx.z * 16;
    const int KB = a.K >> 7;                                   // 128-wide K blocks
    const int bq = (KB + KW - 1) / KW;
    const int kb0 = wave * bq, kb1 = min(KB, kb0 + bq);
    int r = m0 + c16;
    r = r < a.M ? r : a.M - 1;
    const T* __restrict__ arow = reinterpret_cast<const T*>(a.A) + (long long)r * a.lda + fg * 8;
    const T* __restrict__ wp = reinterpret_cast<const T*>(a.W) + ((long long)nt * KB) * 512 + lane * 8;

    // epilogue operands of the finishing wave, requested ahead of the weight stream
    float e_res[4] = {0, 0, 0, 0}, e_bias = 0.0f;
    float* xo = reinterpret_cast<float*>(a.out);
    const int col = nt * 4 + c4;
    if (wave == 0 && ksub == 0) {
        if (a.bias) e_bias = a.bias[col];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = m0 + fg * 4 + q;
            if (row < a.M) e_res[q] = xo[(long long)row * a.ldo + col];
        }
    }

    f32x4 acc = {0, 0, 0, 0};
    for (int kb = kb0; kb < kb1; kb += UB) {
        V8 wf[UB], af[UB][4];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = min(kb + u, kb1 - 1);
            wf[u] = load8_nt(wp + (long long)k * 512);
            if constexpr (COMB) {                       // K block k = query heads 2k, 2k + 1; sub-blocks (0, 1) / (2, 3) are their d halves
#pragma unroll
                for (int sblk = 0; sblk < 4; ++sblk)
#pragma unroll
                    for (int e = 0; e < 8; ++e) af[u][sblk][e] = from_f32<T>(0.0f);
                if (m0 + c16 < a.M) {                   // rows past the grid are masked at the store: their lanes ask for nothing (a single request has 1..2 live rows of 16)
                    combined_pair<T>(a, r, 2 * k, fg, af[u][0], af[u][1]);
                    combined_pair<T>(a, r, 2 * k + 1, fg, af[u][2], af[u][3]);
                }
            } else {
#pragma unroll
                for (int sblk = 0; sblk < 4; ++sblk) af[u][sblk] = load8(arow + k * 128 + sblk * 32);
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (kb + u < kb1) {
#pragma unroll
                for (int sblk = 0; sblk < 4; ++sblk) {
                    V8 wm;
#pragma unroll
                    for (int e = 0; e < 8; ++e) wm[e] = (ksub == sblk) ? wf[u][e] : from_f32<T>(0.0f);
                    mma32(acc, af[u][sblk], wm);
                }
            }
        }
    }
    // the four K quarters of an output column sit in tile columns c4, c4 + 4, c4 + 8, c4 + 12
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc[q] += __shfl_xor(acc[q], 4, 64);
        acc[q] += __shfl_xor(acc[q], 8, 64);
    }
    __shared__ f32x4 red[KW > 1 ? KW - 1 : 1][64];
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave > 0 || ksub != 0) return;
#pragma unroll
    for (int w = 0; w < KW - 1; ++w) acc += red[w][lane];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = m0 + fg * 4 + q;
        if (row >= a.M) continue;
        const float x1 = e_res[q] + (acc[q] + e_bias);
        xo[(long long)row * a.ldo + col] = x1;
        if (a.out2) reinterpret_cast<T*>(a.out2)[(long long)row * a.ldo2 + col] = from_f32<T>(x1);
    }
}

template <class T, int KW, int UB, bool COMB = false>
static int launch_narrow(const SkinnyArgs& a, hipStream_t s) {
    dim3 grid(a.N / 4, 1, (a.M + 15) / 16);
    const double bytes = (double)a.N * a.K * sizeof(T) + (double)a.M * a.K * sizeof(T) + (double)a.M * a.N * 8.0;
    const int slot = prof_begin(PK_SKINNY, bytes, s);
    hipLaunchKernelGGL((gemm_narrow_resid_kernel<T, KW, UB, COMB>), grid, dim3(64 * KW), 0, s, a);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("narrow gemm launch failed"), -1);
}

template <class T>
static int launch_narrow_t(const SkinnyArgs& a, hipStream_t s) {
    const int KB = a.K >> 7;
    if (a.att_o) {
        if (KB > 8 || a.K != a.q_heads * 64 || !a.att_ml || !a.att_kvlen || !a.pos0 || !a.n_new || a.kn < 1 || a.kv_heads < 1 || a.q_heads % a.kv_heads)
            return set_error("launch_skinny: the combining o_proj needs K = q_heads * 64 <= 1024 and the attention launch's controls"), -1;
        return launch_narrow<T, 8, 1, true>(a, s);
    }
    if (KB <= 8) return launch_narrow<T, 8, 1>(a, s);                                  // o_proj: one block per wave
    if (KB <= 40) return launch_narrow<T, 8, (sizeof(T) == 2 ? 5 : 3)>(a, s);          // down_proj (K = 4864: 38 blocks, 5 per wave)
    return launch_narrow<T, 8, 4>(a, s);
}

template <class T, int MT>
static int launch_mt(const SkinnyArgs& a, hipStream_t s) {
    switch (a.epi) {
        case SK_PARTIAL: return launch_one<T, MT, 1, SK_PARTIAL>(a, s);
        case SK_STORE: return launch_one<T, MT, 1, SK_STORE>(a, s);
        case SK_SWIGLU:
            // decode: two (gate, up) pairs per workgroup halve the number of workgroups that re-read the activation rows, and a wave's
            // whole K slice (K <= 1024: 8 k-steps) is requested in one go
            if constexpr (MT == 1 && sizeof(T) == 2) {
                if (a.a_norm && (a.N & 63) == 0 && a.K <= 4 * 8 * 32) return launch_one<T, 1, 4, SK_SWIGLU, 4, 1, 8>(a, s);
            }
            // 64-row chunks (wide decode grids): two (gate, up) pairs per workgroup halve the workgroups that re-read the activation rows
            // (decode step at 128 rows 2.08 -> 1.97 ms; four pairs: 2.01 ms; k-steps in flight 1, 2, 4: 1.96 / 1.97 / 2.03 ms)
            if constexpr (MT == 4 && sizeof(T) == 2) {
                if (a.a_norm && (a.N & 63) == 0) return launch_one<T, 4, 4, SK_SWIGLU, 4, 1, 2>(a, s);
                if (!a.a_norm && (a.N & 63) == 0) return launch_one<T, 4, 4, SK_SWIGLU, 4, 0, 2>(a, s);      // (the MTP heads of a wide grid)
            }
            return a.a_norm ? launch_one<T, MT, 2, SK_SWIGLU, 4, 1>(a, s) : launch_one<T, MT, 2, SK_SWIGLU>(a, s);
        case SK_QKV_ROPE: return a.a_norm ? launch_one<T, MT, 1, SK_QKV_ROPE, 4, 1>(a, s) : launch_one<T, MT, 1, SK_QKV_ROPE>(a, s);
        case SK_RESID:
            // long-K residual projections (down_proj): only N/16 workgroups exist, so K is split over 16 waves and each wave's
            // whole slice (<= 10 k-steps for K = 4864) is requested in one go
            if constexpr (MT == 1) {
                if (a.K >= 2048 && a.K <= 16 * 10 * 32) return launch_one<T, 1, 1, SK_RESID, 16, 0, (sizeof(T) == 2 ? 10 : 5)>(a, s);
            }
            return launch_one<T, MT, 1, SK_RESID>(a, s);
    }
    set_error("launch_skinny: bad epilogue %d", a.epi);
    return -1;
}

template <class T>
static int launch_t(const SkinnyArgs& a, hipStream_t s) {
    // rows per launch chunk: one A fragment set per 16 rows; larger M is covered by blockIdx.z chunks
    if (a.M <= 16) return launch_mt<T, 1>(a, s);
    if (a.M <= 32) return launch_mt<T, 2>(a, s);
    if (a.M <= 128 && (a.K & 63) == 0 && a.epi == SK_PARTIAL && a.split_k > 1) return launch_mid<T>(a, s);      // (backbone down_proj and the stacked MTP heads)
    return launch_mt<T, 4>(a, s);
}

int launch_skinny(const SkinnyArgs& a_in, hipStream_t s) {
    SkinnyArgs a = a_in;
    if (a.M <= 0) return 0;
    if (a.nz < 1) a.nz = 1;
    if (a.split_k < 1) a.split_k = 1;
    if ((a.N & 15) || (a.K & 31) || (a.lda & 7)) {
        set_error("launch_skinny: bad geometry M=%d N=%d K=%d lda=%d", a.M, a.N, a.K, a.lda);
        return -1;
    }
    if (a.epi != SK_PARTIAL && a.split_k != 1) {
        set_error("launch_skinny: fused epilogues need split_k == 1");
        return -1;
    }
    if (a.w_narrow) {
        if (a.epi != SK_RESID || (a.K & 127) || (a.N & 3) || a.nz != 1 || a.split_k != 1 || (a.lda & 7)) {
            set_error("launch_skinny: the narrow weight layout serves the residual projections only (K %% 128 == 0, N %% 4 == 0)");
            return -1;
        }
        return a.dtype == DT_BF16 ? launch_narrow_t<bf16_t>(a, s) : launch_narrow_t<float>(a, s);
    }
    if (a.a_norm && ((a.epi != SK_SWIGLU && a.epi != SK_QKV_ROPE) || (a.K & 3) || a.nz != 1)) {
        set_error("launch_skinny: fused RMSNorm prologue is available for the QKV and gate/up GEMMs only");
        return -1;
    }
    if (a.epi == SK_QKV_ROPE && (a.N != (a.q_heads + 2 * a.kv_heads) * 64)) {
        set_error("launch_skinny: QKV width %d != (q+2kv)*64", a.N);
        return -1;
    }
    if (a.epi == SK_SWIGLU && (a.N & 31)) {
        set_error("launch_skinny: SwiGLU width %d must be a multiple of 32", a.N);
        return -1;
    }
    return a.dtype == DT_BF16 ? launch_t<bf16_t>(a, s) : launch_t<float>(a, s);
}

}  // namespace hvx
