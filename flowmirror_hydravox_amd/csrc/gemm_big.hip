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

constexpr int BM = 256, BN = 256, BK = 64, WM = 128, WN = 64, MT = WM / 16, NT = WN / 16;
constexpr int TILE_ELEMS = (BM + BN) * BK;               // one K-tile of A and W (bf16 elements)

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(GemmArgs a, int tiles_m, int tiles_n) {
    typedef bf16_t T;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE_ELEMS * sizeof(T)];        // 128 KiB; the epilogue stages through it
    T* const lds = reinterpret_cast<T*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 2) * WM, wn0 = (wave & 3) * WN;

    // XCD-aware tile order (bijective for any grid size)
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qn = nwg >> 3, rn = nwg & 7;
    const int wgid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int tn = wgid % tiles_n, rt = wgid / tiles_n;
    const int bz = rt / tiles_m, tm = rt - bz * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const T* __restrict__ Ab = reinterpret_cast<const T*>(a.A) + (long long)bz * a.a_bs;
    const T* __restrict__ Wb = reinterpret_cast<const T*>(a.W);

    // ---- LDS-DMA sources: 4 instructions of A and 4 of W per wave and K-tile, 8 rows each -------------------------------------------
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int lrow = lane >> 3, lslot = lane & 7;
    const char* srcA[4];
    const char* srcW[4];
    bool okA[4], okW[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (wave * 4 + q) * 8 + lrow;               // tile row filled by this lane
        const int chunk = lslot ^ ((r >> 1) & 7);
        okA[q] = (m0 + r) < a.M;
        okW[q] = (n0 + r) < a.N;
        srcA[q] = reinterpret_cast<const char*>(Ab + (long long)(m0 + r) * a.lda + chunk * 8);
        srcW[q] = reinterpret_cast<const char*>(Wb + (long long)(n0 + r) * a.K + chunk * 8);
    }
    auto issue = [&](int kc, int buf) {
        const long long kb = (long long)kc * BK * sizeof(T);
        T* const As = lds + buf * TILE_ELEMS;
        T* const Bs = As + BM * BK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const void* gp = okA[q] ? static_cast<const void*>(srcA[q] + kb) : static_cast<const void*>(g_zero_row);
            __builtin_amdgcn_global_load_lds((glb_ptr)gp, (lds_ptr)(As + (wave * 4 + q) * 8 * BK), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const void* gp = okW[q] ? static_cast<const void*>(srcW[q] + kb) : static_cast<const void*>(g_zero_row);
            __builtin_amdgcn_global_load_lds((glb_ptr)gp, (lds_ptr)(Bs + (wave * 4 + q) * 8 * BK), 16, 0, 0);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    const int fr = lane & 15, fg = lane >> 4;
    const int sw = fr >> 1;                                      // ((row >> 1) & 7) of every fragment row of this lane (rows = 16 i + fr)
    auto compute = [&](int buf) {
        const T* const As = lds + buf * TILE_ELEMS + (wm0 + fr) * BK;
        const T* const Bs = lds + buf * TILE_ELEMS + BM * BK + (wn0 + fr) * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
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

    constexpr int SCR_FLOATS = 32 * (WN + 4);            // 32 staging rows per wave (the lean V^T epilogue transposes two row tiles at a time)
    gemm_epilogue<T, MT, NT, WN, EPI, 1>(a, acc, reinterpret_cast<float*>(smem) + wave * SCR_FLOATS, lane, m0 + wm0, n0 + wn0, bz, 0);
}

}  // namespace

// Eligible: bf16 plain Linear (one tap, one group, no stride / up-sampling / padding), K a multiple of 64, wide N and enough tiles to fill
// the chip.  Returns 1 when the launch was taken, 0 when the caller should use the generic tile forms, -1 on error.
int launch_gemm_big(const GemmArgs& a, hipStream_t s) {
    if (a.dtype != DT_BF16 || a.groups != 1 || a.cin_pad != a.K || a.conv_stride != 1 || a.conv_dil != 1 || a.pad_left != 0 || a.up != 1 ||
        a.rows_in != a.M || (a.K & 63) || a.N < 256 || (a.N & 63))
        return 0;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const long long tiles = (long long)tiles_m * tiles_n * a.batch;
    // below half a round of the 256 CUs the 128 x 128 form (2-3 workgroups per CU, four times as many tiles) balances better (measured: M = 11264,
    // N = 1024: 176 tiles tie; M = 2816: 44 tiles lose 177 vs 267 TF/s)
    static const long long min_tiles = [] { const char* e = getenv("HVX_GEMM_BIG_MIN_TILES"); return e ? atoll(e) : 128LL; }();   // (tuning knob)
    if (tiles < min_tiles || tiles > 0x7fffffffLL) return 0;
    const int slot = prof_begin(PK_GEMM, 2.0 * a.M * a.N * (double)a.K * a.batch, s);
    if (a.epi == EPI_GENERIC) hipLaunchKernelGGL((gemm_big_kernel<EPI_GENERIC>), dim3((unsigned)tiles), dim3(512), 0, s, a, tiles_m, tiles_n);
    else hipLaunchKernelGGL((gemm_big_kernel<EPI_QKV_DIT>), dim3((unsigned)tiles), dim3(512), 0, s, a, tiles_m, tiles_n);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 1 : (set_error("gemm (256-tile form) launch failed"), -1);
}

}  // namespace hvx
