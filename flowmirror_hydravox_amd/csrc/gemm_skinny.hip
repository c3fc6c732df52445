// gemm_skinny.hip — weight-streaming GEMM for the AR decode step (see hvx_kernels.h: SkinnyArgs).
//
// The decode step is HBM-bound on weight bytes: M = batch*K_heads rows (16..128)
// against 0.7-1.1 GB of weights.  Each wave owns NT consecutive 16-column tiles of the output and a
// K-slice; it streams its weight fragments straight from HBM into VGPRs — the checkpoint is
// re-packed at load time into MFMA fragment order [N/16][K/32][64 lanes][8], so every wave-level
// load instruction is one fully coalesced 1 KiB (bf16) burst — and re-reads the small, L2-resident
// activation rows as A fragments.  No LDS, no barriers: a wave never waits on another wave.
// Split-K (blockIdx.y) gives the narrow projections (N = 896) enough waves to cover 256 CUs; the fp32
// partials are reduced in fixed order by reduce_rmsnorm (deterministic, no atomics).
#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

template <class T, int MT, int NT, int EPI>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(SkinnyArgs a) {
    typedef typename Vec8<T>::type V8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int z = blockIdx.z % a.nz;
    const int mchunk = blockIdx.z / a.nz;
    const int m0 = mchunk * (MT * 16);
    const int ntile0 = blockIdx.x * NT;                        // the 4 waves of a workgroup share these NT tiles ...
    const int KT = a.K >> 5;
    const int ks = blockIdx.y;
    const int kt_per = (KT + a.split_k - 1) / a.split_k;
    const int kb0 = ks * kt_per;
    const int kb1 = min(KT, kb0 + kt_per);
    const int kq = (kb1 - kb0 + 3) >> 2;                       // ... and split its K range in four (in-block split-K)
    const int kt0 = kb0 + wave * kq;
    const int kt1 = min(kb1, kt0 + kq);

    const T* __restrict__ A = reinterpret_cast<const T*>(a.A) + (long long)z * a.a_zs;
    const T* __restrict__ W = reinterpret_cast<const T*>(a.W) + (long long)z * a.w_zs;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    // A fragment rows (clamped; masked at the store)
    const T* arow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int r = m0 + i * 16 + fr;
        r = r < a.M ? r : a.M - 1;
        arow[i] = A + (long long)r * a.lda + fg * 8;
    }
    const T* wtile[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int nt = ntile0 + j;
        nt = (nt * 16 < a.N) ? nt : (a.N / 16 - 1);
        wtile[j] = W + ((long long)nt * KT) * 512 + lane * 8;
    }

    // U k-steps per trip: all of their weight / activation loads are issued before the first MFMA so that each wave keeps
    // U * NT KiB of the weight stream in flight (a dependent load->MFMA chain per k-step is latency-bound: 0.6 TB/s measured).
    constexpr int REGS = (NT + MT) * (sizeof(T) == 2 ? 4 : 8);
    constexpr int U = REGS <= 16 ? 8 : (REGS <= 32 ? 4 : 2);
    for (int kt = kt0; kt < kt1; kt += U) {
        V8 wf[U][NT], af[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(kt + u, kt1 - 1);
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[u][j] = load8(wtile[j] + (long long)k * 512);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[u][i] = load8(arow[i] + k * 32);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kt + u < kt1) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) mma32(acc[i][j], af[u][i], wf[u][j]);
            }
        }
    }

    // in-block reduction in fixed wave order (deterministic): waves 1..3 park their tiles in LDS, wave 0 adds and finishes
    __shared__ f32x4 red[3][MT][NT][64];
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) red[wave - 1][i][j][lane] = acc[i][j];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] += red[w][i][j][lane];

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
    } else if constexpr (EPI == SK_STORE) {
        const float* bias = a.bias ? a.bias + (long long)z * a.bias_zs : nullptr;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = (ntile0 + j) * 16 + fr;
            if (col >= (a.n_valid ? a.n_valid : a.N)) continue;
            const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + i * 16 + fg * 4 + r;
                    if (row >= a.M) continue;
                    const float v = acc[i][j][r] + bv;
                    const long long o = (long long)z * a.out_zs + (long long)row * a.ldo + col;
                    if (a.out_f32) reinterpret_cast<float*>(a.out)[o] = v;
                    else reinterpret_cast<T*>(a.out)[o] = from_f32<T>(v);
                }
        }
    } else if constexpr (EPI == SK_SWIGLU) {
        // tiles come in (gate, up) pairs: tile 2p holds gate columns 16p..16p+15, tile 2p+1 the matching up columns
        static_assert(NT % 2 == 0, "SwiGLU needs tile pairs");
#pragma unroll
        for (int j = 0; j < NT; j += 2) {
            const int col = ((ntile0 + j) >> 1) * 16 + fr;
            if (col * 2 >= a.N) continue;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + i * 16 + fg * 4 + r;
                    if (row >= a.M) continue;
                    const float gte = acc[i][j][r], up = acc[i][j + 1][r];
                    const float v = (gte / (1.0f + expf(-gte))) * up;
                    reinterpret_cast<T*>(a.out)[(long long)z * a.out_zs + (long long)row * a.ldo + col] = from_f32<T>(v);
                }
        }
    } else {   // SK_QKV_ROPE
        // The checkpoint rows of every 64-wide head are permuted at pack time (packing.pack_qkv_rows) so that tile t of a
        // head holds d = 8t..8t+7 (lanes fr < 8) and their rotate-half partners d + 32 (lanes fr >= 8): the RoPE pair is one
        // xor-8 shuffle inside a single 16-column tile, and the GEMM can use one tile per workgroup (4x the workgroups).
        static_assert(EPI != SK_QKV_ROPE || NT == 1, "QKV epilogue works on single permuted tiles");
        const int head = ntile0 >> 2, t = ntile0 & 3;       // global head index over [q heads | k heads | v heads]
        const int which = head < a.q_heads ? 0 : (head < a.q_heads + a.kv_heads ? 1 : 2);
        const int hh = which == 0 ? head : (which == 1 ? head - a.q_heads : head - a.q_heads - a.kv_heads);
        const int d = (fr < 8) ? (8 * t + fr) : (32 + 8 * t + (fr - 8));
        const int f = 8 * t + (fr & 7);                     // rotary frequency index (d mod 32)
        const float bias = a.bias ? a.bias[ntile0 * 16 + fr] : 0.0f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + i * 16 + fg * 4 + r;
                float x = acc[i][0][r] + bias;
                const float partner = __shfl_xor(x, 8, 64);         // same row (same fg), the other half of the head
                if (row >= a.M) continue;
                const int si = row / a.kn, lt = row - si * a.kn;
                if (lt >= a.n_new[si]) continue;                    // inactive row
                const int pos = a.pos0[si] + lt;
                if (which < 2) {                                    // rotate-half RoPE (HF Qwen2): d pairs with d +- 32
                    const float cs = a.rope_cos[(long long)pos * 32 + f], sn = a.rope_sin[(long long)pos * 32 + f];
                    x = (fr < 8) ? (x * cs - partner * sn) : (x * cs + partner * sn);
                }
                if (which == 0) {
                    reinterpret_cast<T*>(a.qbuf)[((long long)row * a.q_heads + hh) * 64 + d] = from_f32<T>(x);
                } else if (which == 1) {
                    reinterpret_cast<T*>(a.kcache)[(((long long)a.slot[si] * a.kv_heads + hh) * a.max_ctx + pos) * 64 + d] = from_f32<T>(x);
                } else {
                    reinterpret_cast<T*>(a.vTcache)[(((long long)a.slot[si] * a.kv_heads + hh) * 64 + d) * a.max_ctx + pos] = from_f32<T>(x);
                }
            }
        }
    }
}

template <class T, int MT, int NT, int EPI>
static int launch_one(const SkinnyArgs& a, hipStream_t s) {
    const int ntiles = a.N / 16;
    const int groups = (ntiles + NT - 1) / NT;
    const int mchunks = (a.M + MT * 16 - 1) / (MT * 16);
    dim3 grid(groups, a.split_k, mchunks * a.nz);
    // algorithmic HBM bytes of this launch: every weight once, the activation rows once, the result once
    const double bytes = (double)a.nz * ((double)a.N * a.K * sizeof(T) + (double)a.M * a.K * sizeof(T) +
                                         (double)a.M * a.N * (EPI == SK_PARTIAL ? 4.0 * a.split_k : (double)sizeof(T)));
    const int slot = prof_begin(PK_SKINNY, bytes, s);
    hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, NT, EPI>), grid, dim3(256), 0, s, a);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("skinny gemm launch failed"), -1);
}

template <class T, int MT>
static int launch_mt(const SkinnyArgs& a, hipStream_t s) {
    switch (a.epi) {
        case SK_PARTIAL: return launch_one<T, MT, 1, SK_PARTIAL>(a, s);
        case SK_STORE: return launch_one<T, MT, 1, SK_STORE>(a, s);
        case SK_SWIGLU: return launch_one<T, MT, 2, SK_SWIGLU>(a, s);
        case SK_QKV_ROPE: return launch_one<T, MT, 1, SK_QKV_ROPE>(a, s);
    }
    set_error("launch_skinny: bad epilogue %d", a.epi);
    return -1;
}

template <class T>
static int launch_t(const SkinnyArgs& a, hipStream_t s) {
    // rows per launch chunk: one A fragment set per 16 rows; larger M is covered by blockIdx.z chunks
    if (a.M <= 16) return launch_mt<T, 1>(a, s);
    if (a.M <= 32) return launch_mt<T, 2>(a, s);
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
