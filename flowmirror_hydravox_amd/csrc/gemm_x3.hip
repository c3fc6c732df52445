// gemm_x3.hip — fp32 implicit-GEMM (Conv1d / Linear, same GemmArgs and epilogues as gemm_tiled.hip) on the bf16 matrix cores by operand
// splitting, for gfx950.  The vocoder is fp32 in the reference (hifigan/generator.py) and its convolutions are 672 MF per mel frame:
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 MFMA rate, so the exact-fp32 form spends 256 matrix-pipe cycles per 16 x 16 x 32 step.
// Here every fp32 operand x is split on its way into LDS into two bf16 numbers, hi = bf16(x) and lo = bf16(x - hi) (x = hi + lo up to
// 2^-17 |x|), and a step is three bf16 MFMAs into the same fp32 accumulator,
//      acc += a_lo . b_hi + a_hi . b_lo + a_hi . b_hi          (the dropped a_lo . b_lo term is below 2^-16 of the product),
// 48 cycles instead of 256.  The result differs from an fp32 fma chain by a few 1e-6 relative (tests bound it at 2e-5 against fp64), far
// inside the 1e-3 contract of the vocoder stages; the F0 predictor — whose output is integrated into a phase — stays on the exact form.
//
// Tile forms, loader and epilogues mirror gemm_tiled.hip's fp32 path: 256 threads = 4 waves, K-step 32, global -> registers (fp32) ->
// split -> LDS (two bf16 planes per operand, rows padded to 80 B: conflict-free ds_read_b128), next K-step's global loads in flight during
// the MFMAs, the conv index map (tap, dilation, stride, nearest-upsample, zero padding) in the A loader.  (K-steps of 64 — half the
// barriers, 144-byte rows — measured slower: HiFT 0.201 vs 0.173 s on 12 streams; the larger staging registers cost a wave per SIMD.)
#include "gemm_epilogue.h"

namespace hvx {

namespace {

__device__ __forceinline__ void split8(const f32x8& x, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bf16_t h = f32_to_bf16(x[e]);
        hi[e] = h;
        lo[e] = f32_to_bf16(x[e] - bf16_to_f32(h));
    }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_x3_kernel(GemmArgs a) {
    constexpr int BK = 32, LDK = BK + 8;               // bf16 elements per LDS row (80 B)
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    constexpr int A_VECS = BM / 64, B_VECS = BN / 64;   // 256 threads cover 64 rows x 4 chunks of 8 per pass
    constexpr int SLD = WN + 4;
    constexpr int ROWS_PASS = (64 / WN) * 16;
    constexpr int TILE_BYTES = 2 * (BM + BN) * LDK * 2;
    constexpr int SCR_BYTES = 4 * ROWS_PASS * SLD * 4;
    __shared__ __attribute__((aligned(16))) char smem[TILE_BYTES > SCR_BYTES ? TILE_BYTES : SCR_BYTES];
    bf16_t* const Ah = reinterpret_cast<bf16_t*>(smem);
    bf16_t* const Al = Ah + BM * LDK;
    bf16_t* const Bh = Al + BM * LDK;
    bf16_t* const Bl = Bh + BN * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int bz = blockIdx.z / a.groups, g = blockIdx.z % a.groups;
    const float* __restrict__ Ab = reinterpret_cast<const float*>(a.A) + (long long)bz * a.a_bs + (long long)g * a.a_gs;
    const float* __restrict__ Wb = reinterpret_cast<const float*>(a.W) + (long long)g * a.w_gs;

    const int chunk = tid & 3, lrow = tid >> 2;
    const int nk = a.K / BK;
    const long long in_span = (long long)a.rows_in * a.up;

    auto load_a = [&](int kc, f32x8 (&ra)[A_VECS]) {
        const int k0 = kc * BK;
        const int tap = k0 / a.cin_pad;
        const int ci = k0 - tap * a.cin_pad + chunk * 8;
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
            const int m = m0 + lrow + i * 64;
            const long long idx = (long long)m * a.conv_stride + (long long)tap * a.conv_dil - a.pad_left;
            if (m < a.M && idx >= 0 && idx < in_span) {
                const long long src = (a.up == 1) ? idx : idx / a.up;
                ra[i] = load8(Ab + src * a.lda + ci);
            } else {
                ra[i] = zero8<float>();
            }
        }
    };
    auto load_b = [&](int kc, f32x8 (&rb)[B_VECS]) {
        const int k0 = kc * BK + chunk * 8;
#pragma unroll
        for (int i = 0; i < B_VECS; ++i) {
            const int n = n0 + lrow + i * 64;
            if (n < a.N) rb[i] = load8(Wb + (long long)n * a.K + k0);
            else rb[i] = zero8<float>();
        }
    };
    auto stash = [&](const f32x8 (&ra)[A_VECS], const f32x8 (&rb)[B_VECS]) {
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
            bf16x8 hi, lo;
            split8(ra[i], hi, lo);
            store8(Ah + (lrow + i * 64) * LDK + chunk * 8, hi);
            store8(Al + (lrow + i * 64) * LDK + chunk * 8, lo);
        }
#pragma unroll
        for (int i = 0; i < B_VECS; ++i) {
            bf16x8 hi, lo;
            split8(rb[i], hi, lo);
            store8(Bh + (lrow + i * 64) * LDK + chunk * 8, hi);
            store8(Bl + (lrow + i * 64) * LDK + chunk * 8, lo);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    const int fr = lane & 15, fg = lane >> 4;
    auto compute = [&]() {
        bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            ah[i] = load8(Ah + (wm0 + i * 16 + fr) * LDK + fg * 8);
            al[i] = load8(Al + (wm0 + i * 16 + fr) * LDK + fg * 8);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bh[j] = load8(Bh + (wn0 + j * 16 + fr) * LDK + fg * 8);
            bl[j] = load8(Bl + (wn0 + j * 16 + fr) * LDK + fg * 8);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                mma32(acc[i][j], al[i], bh[j]);
                mma32(acc[i][j], ah[i], bl[j]);
                mma32(acc[i][j], ah[i], bh[j]);
            }
    };

    f32x8 ra[A_VECS], rb[B_VECS];
    load_a(0, ra);
    load_b(0, rb);
    stash(ra, rb);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const bool more = (kc + 1) < nk;
        if (more) {
            load_a(kc + 1, ra);
            load_b(kc + 1, rb);
        }
        compute();
        __syncthreads();
        if (more) {
            stash(ra, rb);
            __syncthreads();
        }
    }
    gemm_epilogue<float, MT, NT, WN, EPI_GENERIC, 2>(a, acc, reinterpret_cast<float*>(smem) + wave * ROWS_PASS * SLD, lane, m0 + wm0, n0 + wn0, bz, g);
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const GemmArgs& a, hipStream_t s) {
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.batch * a.groups);
    // (work is counted in fp32 flops of the convolution, against the fp32 matrix peak: what the exact form would have to do)
    const int slot = prof_begin(PK_GEMM_F32, 2.0 * a.M * a.N * (double)a.K * a.batch * a.groups, s);
    hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, WM, WN>), grid, dim3(256), 0, s, a);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 1 : (set_error("gemm (split-bf16 form) launch failed"), -1);
}

}  // namespace

// 1 = launched, 0 = not eligible, -1 = error.  Taken when the caller allows the split form (GemmArgs.x3) for an fp32 generic-epilogue GEMM.
int launch_gemm_x3(const GemmArgs& a, hipStream_t s) {
    if (!a.x3 || a.dtype != DT_F32 || a.epi != EPI_GENERIC) return 0;
    const long long blocks128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch * a.groups;
    if (a.N <= 64) return a.M <= 4096 ? 0 : launch_cfg<128, 64, 32, 64>(a, s);
    if (blocks128 < 256) return 0;
    return launch_cfg<128, 128, 64, 64>(a, s);
}

}  // namespace hvx
