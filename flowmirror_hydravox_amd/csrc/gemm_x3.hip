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
#include <stdlib.h>

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


// ---- the same product with BOTH operands already stored as (hi, lo) bf16 plane pairs (GemmArgs.a_planes / w_planes) ---------------------------
// What gemm_x3_kernel spends its time on is not the matrix cores: per K-step a wave issues 48 MFMAs beside ~570 other instructions (global
// loads into registers, eight conversions + subtractions per 8 values, LDS stores, two barriers) at a prefetch distance of one K-step.  When the
// producer has written the pair (a GEMM epilogue with out_planes / out2_planes, or the weight loader) the tiles go from global memory straight
// into LDS by LDS-DMA, double-buffered, one barrier per K-step, and the loop body is fragment reads + 3 MFMAs per 16 x 16 x 32 step — the
// structure of gemm_tiled.hip's bf16 form with four planes instead of two.  Same arithmetic as above, value for value (same (hi, lo) pairs,
// same MFMA order), so the two forms give identical results.
// LDS image of a plane tile: [rows][32 bf16] (64-byte rows, lane-linear as the DMA deposits it); the lane that fills slot s of row r fetches
// chunk s ^ f(r), f(r) = (-(r >> 2)) & 3, which puts the 16 reads of every ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) on 16 distinct
// 16-byte slots (checked with SQ_LDS_BANK_CONFLICT on gemm_tiled's bf16 form: 0.1 % of the LDS cycles).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_x3p_kernel(GemmArgs a) {
    constexpr int BK = 32;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    static_assert(BM % 64 == 0 && BN % 64 == 0, "a wave instruction deposits 16 rows; 4 waves");
    constexpr int SLD = WN + 4;
    constexpr int ROWS_PASS = (64 / WN) * 16;
    constexpr int STAGE = 2 * (BM + BN) * BK;              // bf16 elements per stage: Ah | Al | Bh | Bl
    constexpr int TILE_BYTES = 2 * STAGE * 2;
    constexpr int SCR_BYTES = 4 * ROWS_PASS * SLD * 4;
    __shared__ __attribute__((aligned(16))) char smem[TILE_BYTES > SCR_BYTES ? TILE_BYTES : SCR_BYTES];
    bf16_t* const tiles = reinterpret_cast<bf16_t*>(smem);
    auto Ah = [&](int buf) { return tiles + buf * STAGE; };
    auto Al = [&](int buf) { return tiles + buf * STAGE + BM * BK; };
    auto Bh = [&](int buf) { return tiles + buf * STAGE + 2 * BM * BK; };
    auto Bl = [&](int buf) { return tiles + buf * STAGE + 2 * BM * BK + BN * BK; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int bz = blockIdx.z / a.groups, g = blockIdx.z % a.groups;
    const bf16_t* __restrict__ Ab = reinterpret_cast<const bf16_t*>(a.A) + (long long)bz * a.a_bs + (long long)g * a.a_gs;
    const bf16_t* __restrict__ Wb = reinterpret_cast<const bf16_t*>(a.W) + (long long)g * a.w_gs;
    const int nk = a.K / BK;
    const long long in_span = (long long)a.rows_in * a.up;

    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int lrow = lane >> 2, lslot = lane & 3;
    const int gchunk = lslot ^ ((-(lrow >> 2)) & 3);
    constexpr int A_INS = BM / 64, B_INS = BN / 64;
    auto issue = [&](int kc, int buf) {
        const int k0 = kc * BK;
        const int tap = k0 / a.cin_pad;
        const int ci = k0 - tap * a.cin_pad + gchunk * 8;
#pragma unroll
        for (int q = 0; q < A_INS; ++q) {
            const int r16 = (wave * A_INS + q) * 16;
            const int m = m0 + r16 + lrow;
            const long long idx = (long long)m * a.conv_stride + (long long)tap * a.conv_dil - a.pad_left;
            const bool ok = m < a.M && idx >= 0 && idx < in_span;
            const long long src = (a.up == 1) ? idx : idx / a.up;
            const bf16_t* gp = Ab + src * a.lda + ci;
            const void* ph = ok ? static_cast<const void*>(gp) : static_cast<const void*>(g_zero_row);
            const void* pl = ok ? static_cast<const void*>(gp + a.a_plane) : static_cast<const void*>(g_zero_row);
            __builtin_amdgcn_global_load_lds((glb_ptr)ph, (lds_ptr)(Ah(buf) + r16 * BK), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)pl, (lds_ptr)(Al(buf) + r16 * BK), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < B_INS; ++q) {
            const int r16 = (wave * B_INS + q) * 16;
            const int n = n0 + r16 + lrow;
            const bf16_t* gp = Wb + (long long)n * a.K + k0 + gchunk * 8;
            const void* ph = n < a.N ? static_cast<const void*>(gp) : static_cast<const void*>(g_zero_row);
            const void* pl = n < a.N ? static_cast<const void*>(gp + a.w_plane) : static_cast<const void*>(g_zero_row);
            __builtin_amdgcn_global_load_lds((glb_ptr)ph, (lds_ptr)(Bh(buf) + r16 * BK), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)pl, (lds_ptr)(Bl(buf) + r16 * BK), 16, 0, 0);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int fr = lane & 15, fg = lane >> 4;
    const int fsw = (fg ^ ((-(fr >> 2)) & 3)) * 8;               // slot of this lane's fragment chunk in its row
    auto compute = [&](int cur) {
        bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            ah[i] = load8(Ah(cur) + (wm0 + i * 16 + fr) * BK + fsw);
            al[i] = load8(Al(cur) + (wm0 + i * 16 + fr) * BK + fsw);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bh[j] = load8(Bh(cur) + (wn0 + j * 16 + fr) * BK + fsw);
            bl[j] = load8(Bl(cur) + (wn0 + j * 16 + fr) * BK + fsw);
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
    issue(0, 0);
    for (int kc = 0; kc < nk; ++kc) {
        __syncthreads();                                 // tile kc has landed (the barrier drains the DMA queue); buffer (kc + 1) & 1 is free
        if (kc + 1 < nk) issue(kc + 1, (kc + 1) & 1);
        compute(kc & 1);
    }
    __syncthreads();                                     // the epilogue reuses the tile memory as staging
    gemm_epilogue<float, MT, NT, WN, EPI_GENERIC, 2>(a, acc, reinterpret_cast<float*>(smem) + wave * ROWS_PASS * SLD, lane, m0 + wm0, n0 + wn0, bz, g);
}


// ---- the plane-pair product on a 128 x 128 x 64 tile with EIGHT waves (round 5) -----------------------------------------------------------
// gemm_x3p_kernel<128,128> above is bound by what its loop does between two barriers: a 32-wide K-step is 64-byte row pieces for the LDS-DMA (the
// L2 -> LDS rate of 64-byte pieces is ~42 GB/s per CU, of 128-byte pieces 51-67: tools/ingest_lab.hip) and one barrier per 48 MFMAs of a wave.
// Here a K-tile is 64 wide — 128-byte row pieces, half the barriers per k — which needs 64 KB per stage (4 planes x 128 rows x 128 B), i.e. ONE
// workgroup per CU; eight waves (4 x 2, a wave owns 32 x 64 of the tile) keep two waves on every SIMD.  LDS image of a plane tile: [128 rows][64 bf16],
// lane-linear as the DMA deposits it (a wave instruction = 8 rows of 128 B); bank spread by a swizzle on the SOURCE address, chunk ^ ((row >> 1) & 7),
// undone by the fragment reads (gemm_big.hip's scheme: the 16 lanes of every ds_read_b128 group touch 16 distinct 16-byte slots).
// Same (hi, lo) pairs, same k order, same three MFMAs per step: results are bit-identical to the 4-wave form.
__global__ __launch_bounds__(512) void gemm_x3p8_kernel(GemmArgs a) {
    constexpr int BM = 128, BN = 128, BK = 64;
    constexpr int WM = 32, WN = 64, MT = WM / 16, NT = WN / 16, WAVES_N = BN / WN;
    constexpr int SLD = WN + 4, ROWS_PASS = (64 / WN) * 16;
    constexpr int PLANE = 128 * BK;                         // bf16 elements of one plane tile (A and W tiles are both 128 rows)
    constexpr int STAGE = 4 * PLANE;                        // Ah | Al | Bh | Bl
    static_assert(8 * ROWS_PASS * SLD * 4 <= 2 * STAGE * 2, "epilogue staging fits the tile memory");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE * 2];
    bf16_t* const tiles = reinterpret_cast<bf16_t*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int bz = blockIdx.z / a.groups, g = blockIdx.z % a.groups;
    const bf16_t* __restrict__ Ab = reinterpret_cast<const bf16_t*>(a.A) + (long long)bz * a.a_bs + (long long)g * a.a_gs;
    const bf16_t* __restrict__ Wb = reinterpret_cast<const bf16_t*>(a.W) + (long long)g * a.w_gs;
    const int nk = a.K / BK;
    const long long in_span = (long long)a.rows_in * a.up;

    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int lrow = lane >> 3, lslot = lane & 7;           // a DMA instruction: 8 rows x 8 slots of 16 B
    auto issue = [&](int kc, int buf) {
        const int k0 = kc * BK;
        const int tap = k0 / a.cin_pad;                      // (cin_pad is a multiple of 64: a K-tile never straddles two taps)
        const int ci = k0 - tap * a.cin_pad;
        bf16_t* const st = tiles + buf * STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r8 = wave * 16 + q * 8;                // first tile row of this instruction
            const int gchunk = lslot ^ ((q * 4 + (lrow >> 1)) & 7);       // ((r >> 1) & 7) with r = r8 + lrow, r8 a multiple of 8
            {
                const int m = m0 + r8 + lrow;
                const long long idx = (long long)m * a.conv_stride + (long long)tap * a.conv_dil - a.pad_left;
                const bool ok = m < a.M && idx >= 0 && idx < in_span;
                const long long src = (a.up == 1) ? idx : idx / a.up;
                const bf16_t* gp = Ab + src * a.lda + ci + gchunk * 8;
                const void* ph = ok ? static_cast<const void*>(gp) : static_cast<const void*>(g_zero_row);
                const void* pl = ok ? static_cast<const void*>(gp + a.a_plane) : static_cast<const void*>(g_zero_row);
                __builtin_amdgcn_global_load_lds((glb_ptr)ph, (lds_ptr)(st + r8 * BK), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_ptr)pl, (lds_ptr)(st + PLANE + r8 * BK), 16, 0, 0);
            }
            {
                const int n = n0 + r8 + lrow;
                const bf16_t* gp = Wb + (long long)n * a.K + k0 + gchunk * 8;
                const void* ph = n < a.N ? static_cast<const void*>(gp) : static_cast<const void*>(g_zero_row);
                const void* pl = n < a.N ? static_cast<const void*>(gp + a.w_plane) : static_cast<const void*>(g_zero_row);
                __builtin_amdgcn_global_load_lds((glb_ptr)ph, (lds_ptr)(st + 2 * PLANE + r8 * BK), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_ptr)pl, (lds_ptr)(st + 3 * PLANE + r8 * BK), 16, 0, 0);
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int fr = lane & 15, fg = lane >> 4;
    const int sw = (fr >> 1) & 7;                            // ((row >> 1) & 7) of every fragment row of this lane (rows = 16 i + fr)
    auto compute = [&](int cur) {
        const bf16_t* const st = tiles + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = ((ks * 4 + fg) ^ sw) * 8;
            bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = load8(st + (wm0 + i * 16 + fr) * BK + off);
                al[i] = load8(st + PLANE + (wm0 + i * 16 + fr) * BK + off);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bh[j] = load8(st + 2 * PLANE + (wn0 + j * 16 + fr) * BK + off);
                bl[j] = load8(st + 3 * PLANE + (wn0 + j * 16 + fr) * BK + off);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    mma32(acc[i][j], al[i], bh[j]);
                    mma32(acc[i][j], ah[i], bl[j]);
                    mma32(acc[i][j], ah[i], bh[j]);
                }
        }
    };
    issue(0, 0);
    for (int kc = 0; kc < nk; ++kc) {
        __syncthreads();                                 // tile kc has landed (the barrier drains the DMA queue); buffer (kc + 1) & 1 is free
        if (kc + 1 < nk) issue(kc + 1, (kc + 1) & 1);
        compute(kc & 1);
    }
    __syncthreads();                                     // the epilogue reuses the tile memory as staging
#if defined(HVX_LAB_X3_EPI) && HVX_LAB_X3_EPI == 1
    {   // (lab: the K-loop alone — every accumulator stays observable through one value per lane)
        f32x4 t = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) t += acc[i][j];
        if (t[0] + t[1] + t[2] + t[3] == 123.456f) reinterpret_cast<float*>(a.out)[lane] = t[0];
        return;
    }
#elif defined(HVX_LAB_X3_EPI) && HVX_LAB_X3_EPI == 2
    {   // (lab: the epilogue's instructions without its memory traffic: only row 0 is stored / read)
        GemmArgs a2 = a;
        a2.M = 1;
        gemm_epilogue<float, MT, NT, WN, EPI_GENERIC, 2>(a2, acc, reinterpret_cast<float*>(smem) + wave * ROWS_PASS * SLD, lane, m0 + wm0, n0 + wn0, bz, g);
        return;
    }
#endif
    gemm_epilogue<float, MT, NT, WN, EPI_GENERIC, 2>(a, acc, reinterpret_cast<float*>(smem) + wave * ROWS_PASS * SLD, lane, m0 + wm0, n0 + wn0, bz, g);
}

// ---- plane-pair convolution with the INPUT ROWS RESIDENT IN LDS (64 channels in, <= 64 out, stride 1): the last vocoder stage -------------------
// A k-tap convolution read through gemm_x3p_kernel fetches the same input rows k times (tap t is the tile shifted by t * dil rows) and passes
// 2 k barriers with 24 MFMAs per wave between them; at 64 channels that, not the matrix cores or HBM, sets the time (252 us for 350-690 MB of
// traffic at 675 841 rows).  Here the workgroup's 128 output rows + their (k - 1) * dil halo rows are brought into LDS ONCE (both planes, one
// barrier), every tap reads its fragments from that image at a row offset, and only the weight tile of a tap (16 KB) is staged per step:
// k barriers with 48 MFMAs per wave between them and a tenth of the L2 -> LDS traffic.
template <int MAXR>
__global__ __launch_bounds__(256) void conv64_x3p_kernel(GemmArgs a) {
    constexpr int BM = 128, MT = 2, NT = 4, WN = 64;
    constexpr int SLD = WN + 4, ROWS_PASS = 16;
    constexpr int AIMG = MAXR * 32;                        // elements of one (plane, k-half) image of the input rows: [MAXR][32]
    constexpr int BIMG = 64 * 32;                          // one (plane, k-half) image of a tap's weights: [64][32]
    constexpr int SCR_BYTES = 4 * ROWS_PASS * SLD * 4;
    static_assert(2 * 4 * BIMG * 2 >= SCR_BYTES, "the epilogue staging fits the weight buffers");
    __shared__ __attribute__((aligned(16))) bf16_t As[4 * AIMG];      // [plane][half][row][32]
    __shared__ __attribute__((aligned(16))) bf16_t Bs[2 * 4 * BIMG];  // [buffer][plane][half][n][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM;
    const int taps = a.K / 64;
    const int rows_tile = BM + (taps - 1) * a.conv_dil;               // <= MAXR (checked by the launcher)
    const bf16_t* __restrict__ Ab = reinterpret_cast<const bf16_t*>(a.A);
    const bf16_t* __restrict__ Wb = reinterpret_cast<const bf16_t*>(a.W);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int lrow = lane >> 2, lslot = lane & 3;
    const int gchunk = lslot ^ ((-(lrow >> 2)) & 3);                  // (16-row groups start at multiples of 16: the swizzle of a row is that of row & 15)
    // ---- input rows: tile row j = global row m0 + j - pad_left -----------------------------------------------------------------------------
    const int groups = (rows_tile + 15) >> 4;
    for (int q = wave; q < 4 * groups; q += 4) {
        const int img = q / groups, g16 = q - img * groups;           // img = plane * 2 + half
        const long long grow = (long long)m0 + g16 * 16 + lrow - a.pad_left;
        const bool ok = grow >= 0 && grow < a.rows_in && (g16 * 16 + lrow) < rows_tile;
        const bf16_t* gp = Ab + (img >> 1) * a.a_plane + grow * a.lda + (img & 1) * 32 + gchunk * 8;
        const void* src = ok ? static_cast<const void*>(gp) : static_cast<const void*>(g_zero_row);
        __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(As + img * AIMG + g16 * 16 * 32), 16, 0, 0);
    }
    auto issue_w = [&](int tap, int buf) {
        // 16 instructions per tap (4 images x 64 rows), 4 per wave: wave w fills image w
        const int img = wave;
#pragma unroll
        for (int g16 = 0; g16 < 4; ++g16) {
            const int n = g16 * 16 + lrow;
            const bf16_t* gp = Wb + (img >> 1) * a.w_plane + (long long)n * a.K + tap * 64 + (img & 1) * 32 + gchunk * 8;
            const void* src = n < a.N ? static_cast<const void*>(gp) : static_cast<const void*>(g_zero_row);
            __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(Bs + (buf * 4 + img) * BIMG + g16 * 16 * 32), 16, 0, 0);
        }
    };
    issue_w(0, 0);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int fr = lane & 15, fg = lane >> 4;
    const int wm0 = wave * 32;
    const int bsw = (fg ^ ((-(fr >> 2)) & 3)) * 8;                    // weight rows n = 16 j + fr
    for (int tap = 0; tap < taps; ++tap) {
        __syncthreads();                                 // the weights of `tap` (and, the first time, the input rows) have landed; buffer (tap + 1) & 1 is free
        if (tap + 1 < taps) issue_w(tap + 1, (tap + 1) & 1);
        const int j0 = wm0 + fr + tap * a.conv_dil;      // this lane's row of the first row tile at this tap (the second is 16 rows further: same swizzle)
        const int asw = (fg ^ ((-((j0 & 15) >> 2)) & 3)) * 8;
        const bf16_t* const bt = Bs + (tap & 1) * 4 * BIMG;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = load8(As + (0 + h) * AIMG + (j0 + i * 16) * 32 + asw);
                al[i] = load8(As + (2 + h) * AIMG + (j0 + i * 16) * 32 + asw);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bh[j] = load8(bt + (0 + h) * BIMG + (j * 16 + fr) * 32 + bsw);
                bl[j] = load8(bt + (2 + h) * BIMG + (j * 16 + fr) * 32 + bsw);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    mma32(acc[i][j], al[i], bh[j]);
                    mma32(acc[i][j], ah[i], bl[j]);
                    mma32(acc[i][j], ah[i], bh[j]);
                }
        }
    }
    __syncthreads();                                     // the epilogue stages through the weight buffers
#if defined(HVX_LAB_X3_EPI) && HVX_LAB_X3_EPI == 1
    {   // (lab: the K-loop alone — every accumulator stays observable through one value per lane)
        f32x4 t = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) t += acc[i][j];
        if (t[0] + t[1] + t[2] + t[3] == 123.456f) reinterpret_cast<float*>(a.out)[lane] = t[0];
        return;
    }
#elif defined(HVX_LAB_X3_EPI) && HVX_LAB_X3_EPI == 2
    {   // (lab: the epilogue's instructions without its memory traffic: only row 0 is stored / read)
        GemmArgs a2 = a;
        a2.M = 1;
        gemm_epilogue<float, MT, NT, WN, EPI_GENERIC, 2>(a2, acc, reinterpret_cast<float*>(Bs) + wave * ROWS_PASS * SLD, lane, m0 + wm0, 0, 0, 0);
        return;
    }
#endif
    gemm_epilogue<float, MT, NT, WN, EPI_GENERIC, 2>(a, acc, reinterpret_cast<float*>(Bs) + wave * ROWS_PASS * SLD, lane, m0 + wm0, 0, 0, 0);
}

// (Round 5, measured and removed: the resident-row form of conv64_x3p_kernel below for the 128-channel stage — 128 + halo input rows of both planes in LDS
// once (94 KB), only the 32 KB weight tile of a (tap, 64-channel half) staged per step, 8 waves.  Bit-identical, and SLOWER: vocoder 17.5 vs 17.1 ms per
// 5632-frame utterance.  At 158 KB a CU holds ONE workgroup, so the 94 KB prologue of every tile is exposed; the 64-channel form works because two
// 81 KB workgroups cover each other's prologues.)

template <int BM, int BN, int WM, int WN>
int launch_cfg_p(const GemmArgs& a, hipStream_t s) {
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.batch * a.groups);
    const int slot = prof_begin(PK_GEMM_F32, 2.0 * a.M * a.N * (double)a.K * a.batch * a.groups, s);
    hipLaunchKernelGGL((gemm_x3p_kernel<BM, BN, WM, WN>), grid, dim3(256), 0, s, a);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 1 : (set_error("gemm (plane-pair form) launch failed"), -1);
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
    if (a.a_planes || a.w_planes) {
        // plane pairs cannot fall back to an fp32 kernel: every shape takes one of the two tile forms
        if (!a.a_planes || !a.w_planes || a.dtype != DT_F32 || a.epi != EPI_GENERIC) return set_error("gemm (plane-pair form): both operands must be plane pairs of an fp32 generic-epilogue GEMM"), -1;
        if ((a.lda & 7) || (a.a_plane & 7) || (a.w_plane & 7)) return set_error("gemm (plane-pair form): 16-byte alignment"), -1;
        const int resident = (int)opt(OPT_CONV64_RESIDENT);      // (A/B option conv64_resident)
        if (resident && a.N <= 64 && a.cin_pad == 64 && a.conv_stride == 1 && a.up == 1 && a.groups == 1 && a.batch == 1 && a.M >= 4096 &&
            128 + (a.K / 64 - 1) * a.conv_dil <= 192) {
            const int slot = prof_begin(PK_GEMM_F32, 2.0 * a.M * a.N * (double)a.K, s);
            // (Round 5, measured: 160 resident rows instead of 192 — 73 728 B of LDS, every convolution of the stage but (k = 11, dilation 5) — 213.9 vs
            // 214.7 us per launch, and a start offset for the second workgroup of a CU changes nothing either: the two workgroups of a CU already overlap.)
            hipLaunchKernelGGL((conv64_x3p_kernel<192>), dim3((a.M + 127) / 128), dim3(256), 0, s, a);
            prof_end(slot, s);
            return hipGetLastError() == hipSuccess ? 1 : (set_error("conv (resident-row plane-pair form) launch failed"), -1);
        }
        if (a.N <= 64) return launch_cfg_p<128, 64, 32, 64>(a, s);
        const int x3p8 = (int)opt(OPT_X3P8);                                  // (A/B option x3p8)
        if (x3p8 && (a.K & 63) == 0 && (a.cin_pad & 63) == 0) {
            dim3 grid((a.M + 127) / 128, (a.N + 127) / 128, a.batch * a.groups);
            const int slot = prof_begin(PK_GEMM_F32, 2.0 * a.M * a.N * (double)a.K * a.batch * a.groups, s);
            hipLaunchKernelGGL(gemm_x3p8_kernel, grid, dim3(512), 0, s, a);
            prof_end(slot, s);
            return hipGetLastError() == hipSuccess ? 1 : (set_error("gemm (plane-pair form, 8 waves) launch failed"), -1);
        }
        return launch_cfg_p<128, 128, 64, 64>(a, s);
    }
    if (!a.x3 || a.dtype != DT_F32 || a.epi != EPI_GENERIC) return 0;
    const long long blocks128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch * a.groups;
    if (a.N <= 64) return a.M <= 4096 ? 0 : launch_cfg<128, 64, 32, 64>(a, s);
    if (blocks128 < 256) return 0;
    return launch_cfg<128, 128, 64, 64>(a, s);
}

}  // namespace hvx
