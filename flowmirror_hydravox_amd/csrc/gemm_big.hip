// gemm_big.hip — the 256 x 256 x 64 tile form of the bf16 Linear GEMM (DiT: M = frames x CFG x utterances in the tens of thousands,
// N = 1024..3072, K = 1024..2048), for gfx950.  out = epi(A[M][K] . W[N][K]^T), same GemmArgs / epilogues as gemm_tiled.hip.
//
// Workgroup = 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of the tile = 8 x 4 accumulator tiles of 16 x 16 (128 VGPRs),
// so every A fragment read from LDS feeds 4 MFMAs and every B fragment 8: half the LDS bytes per flop of the 128 x 128 form, and two waves
// per SIMD (one workgroup per CU, 128 KiB of LDS) so one wave's fragment reads sit under the other's MFMAs.
//
// Staging: LDS-DMA (global_load_lds_dwordx4), double-buffered K-tiles of 64, ONE barrier per K-tile (it drains the DMA queue: the tile
// issued one iteration ago has had a whole MFMA phase to land).  A wave instruction deposits 64 x 16 B lane-linearly = 8 rows of 128 B;
// the bank spread comes from a swizzle on the SOURCE address: the lane that fills 16-byte slot p of row r fetches chunk p ^ ((r >> 1) & 7),
// and the fragment reads apply the same involution — the 16 lanes of every ds_read_b128 group then touch 16 distinct slots of the 256-byte
// bank row (rows of equal parity differ in (r >> 1) & 7; the two k-groups of a lane group differ in bit 0 of the chunk).
//
// Tile order: the linear workgroup id is remapped so that each XCD (id % 8) walks a contiguous run of tiles, column tile fastest — the 32
// CUs of an XCD then share A row panels and the W panel in their private L2 (guide T1, bijective form).
#include <stdlib.h>

#include "gemm_epilogue.h"

namespace hvx {

namespace {

constexpr int BM = 256, WM = 128, WN = 64, MT = WM / 16, NT = WN / 16;

// Two forms share the code:
//   <BN 256, BK 64, 8 waves>: one workgroup per CU (128 KiB of LDS), 2 x 4 waves
//   <BN 128, BK 32, 4 waves>: TWO independent workgroups per CU (48 KiB each), 2 x 2 waves — the same 128 x 64 wave tile and two waves per
//       SIMD, but the two waves of a SIMD belong to different workgroups: one's epilogue and barrier waits run under the other's K-loop.
//       K-tile rows are 64 B: a DMA instruction deposits 16 rows, swizzle chunk ^ ((row >> 1) & 3) (conflict-free ds_read_b128, checked by
//       enumeration of the lane groups).  Correct (GPU tests pass with it) but SLOWER than the first form — 1.5x the L2 -> LDS bytes per
//       flop and twice the barriers cost more than the overlap returns (out_proj 154 vs 149 us, FF1 276 vs 258, FF2 279 vs 254 at
//       M = 45056) — so it is not instantiated.
template <int EPI, int BN, int BK, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64, 2) void gemm_big_kernel(GemmArgs a, int tiles_m, int tiles_n) {
    typedef bf16_t T;
    constexpr int TILE_ELEMS = (BM + BN) * BK;               // one K-tile of A and W (bf16 elements)
    constexpr int WAVES_N = BN / WN;
    constexpr int SLOTS = BK / 8;                            // 16-byte chunks per K-tile row
    constexpr int RPI = 64 / SLOTS;                          // rows per DMA instruction (1 KiB)
    constexpr int QA = BM / RPI / NWAVE, QB = BN / RPI / NWAVE;
    constexpr int SCR_FLOATS = 32 * (WN + 4);                // 32 staging rows per wave (the lean V^T epilogue transposes two row tiles at a time)
    static_assert((BM / WM) * WAVES_N == NWAVE && QA * RPI * NWAVE == BM && QB * RPI * NWAVE == BN, "tile geometry");
    static_assert(NWAVE * SCR_FLOATS * 4 <= 2 * TILE_ELEMS * 2, "epilogue staging fits the tile memory");
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE_ELEMS * sizeof(T)];        // the epilogue stages through it
    T* const lds = reinterpret_cast<T*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

    // XCD-aware tile order (bijective for any grid size)
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qn = nwg >> 3, rn = nwg & 7;
    const int wgid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int tn = wgid % tiles_n, rt = wgid / tiles_n;
    const int bz = rt / tiles_m, tm = rt - bz * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const T* __restrict__ Ab = reinterpret_cast<const T*>(a.A) + (long long)bz * a.a_bs;
    const T* __restrict__ Wb = reinterpret_cast<const T*>(a.W);

    // ---- LDS-DMA sources: QA instructions of A and QB of W per wave and K-tile, RPI rows each ------------------------------------------
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int lrow = lane / SLOTS, lslot = lane % SLOTS;
    const char* srcA[QA];
    const char* srcW[QB];
    bool okA[QA], okW[QB];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int r = (wave * QA + q) * RPI + lrow;               // tile row filled by this lane
        const int chunk = lslot ^ ((r >> 1) & (SLOTS - 1));
        okA[q] = (m0 + r) < a.M;
        srcA[q] = reinterpret_cast<const char*>(Ab + (long long)(m0 + r) * a.lda + chunk * 8);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int r = (wave * QB + q) * RPI + lrow;
        const int chunk = lslot ^ ((r >> 1) & (SLOTS - 1));
        okW[q] = (n0 + r) < a.N;
        srcW[q] = reinterpret_cast<const char*>(Wb + (long long)(n0 + r) * a.K + chunk * 8);
    }
    auto issue = [&](int kc, int buf) {
        const long long kb = (long long)kc * BK * sizeof(T);
        T* const As = lds + buf * TILE_ELEMS;
        T* const Bs = As + BM * BK;
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const void* gp = okA[q] ? static_cast<const void*>(srcA[q] + kb) : static_cast<const void*>(g_zero_row);
            __builtin_amdgcn_global_load_lds((glb_ptr)gp, (lds_ptr)(As + (wave * QA + q) * RPI * BK), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const void* gp = okW[q] ? static_cast<const void*>(srcW[q] + kb) : static_cast<const void*>(g_zero_row);
            __builtin_amdgcn_global_load_lds((glb_ptr)gp, (lds_ptr)(Bs + (wave * QB + q) * RPI * BK), 16, 0, 0);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    const int fr = lane & 15, fg = lane >> 4;
    const int sw = (fr >> 1) & (SLOTS - 1);                      // ((row >> 1) & (SLOTS - 1)) of every fragment row of this lane (rows = 16 i + fr)
    auto compute = [&](int buf) {
        const T* const As = lds + buf * TILE_ELEMS + (wm0 + fr) * BK;
        const T* const Bs = lds + buf * TILE_ELEMS + BM * BK + (wn0 + fr) * BK;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int off = ((kk * 4 + fg) ^ sw) * 8;
            bf16x8 af[MT], bf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = load8(Bs + j * 16 * BK + off);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = load8(As + i * 16 * BK + off);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) mma32(acc[i][j], af[i], bf[j]);
        }
    };

    const int nk = a.K / BK;
    issue(0, 0);
    for (int kc = 0; kc < nk; ++kc) {
        __syncthreads();                                 // tile kc has landed (the barrier drains the DMA queue); buffer (kc+1)&1 is free
        if (kc + 1 < nk) issue(kc + 1, (kc + 1) & 1);
        compute(kc & 1);
    }
    __syncthreads();                                     // the epilogue reuses the tile memory as staging

    gemm_epilogue<T, MT, NT, WN, EPI, 1>(a, acc, reinterpret_cast<float*>(smem) + wave * SCR_FLOATS, lane, m0 + wm0, n0 + wn0, bz, 0);
}

template <int BN, int BK, int NWAVE>
int launch_form(const GemmArgs& a, hipStream_t s, long long min_tiles) {
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const long long tiles = (long long)tiles_m * tiles_n * a.batch;
    if (tiles < min_tiles || tiles > 0x7fffffffLL) return 0;
    const int slot = prof_begin(PK_GEMM, 2.0 * a.M * a.N * (double)a.K * a.batch, s);
    if (a.epi == EPI_GENERIC) hipLaunchKernelGGL((gemm_big_kernel<EPI_GENERIC, BN, BK, NWAVE>), dim3((unsigned)tiles), dim3(NWAVE * 64), 0, s, a, tiles_m, tiles_n);
    else hipLaunchKernelGGL((gemm_big_kernel<EPI_QKV_DIT, BN, BK, NWAVE>), dim3((unsigned)tiles), dim3(NWAVE * 64), 0, s, a, tiles_m, tiles_n);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 1 : (set_error("gemm (256-tile form) launch failed"), -1);
}

}  // namespace

// Eligible: bf16 plain Linear (one tap, one group, no stride / up-sampling / padding), K a multiple of 64, wide N and enough tiles to fill
// the chip.  Returns 1 when the launch was taken, 0 when the caller should use the generic tile forms, -1 on error.
int launch_gemm_big(const GemmArgs& a, hipStream_t s) {
    if (a.dtype != DT_BF16 || a.groups != 1 || a.cin_pad != a.K || a.conv_stride != 1 || a.conv_dil != 1 || a.pad_left != 0 || a.up != 1 ||
        a.rows_in != a.M || (a.K & 63) || a.N < 256 || (a.N & 63))
        return 0;
    // below half a round of the 256 CUs the 128 x 128 form (2-3 workgroups per CU, four times as many tiles) balances better (measured: M = 11264,
    // N = 1024: 176 tiles tie; M = 2816: 44 tiles lose 177 vs 267 TF/s)
    static const long long min_tiles = [] { const char* e = getenv("HVX_GEMM_BIG_MIN_TILES"); return e ? atoll(e) : 128LL; }();   // (tuning knob)
    // (launch_form<128, 32, 4> — two workgroups per CU — measured 3-10 % slower on every DiT Linear at M = 45056: flow solve 450 vs 433 ms)
    return launch_form<256, 64, 8>(a, s, min_tiles);
}

}  // namespace hvx
