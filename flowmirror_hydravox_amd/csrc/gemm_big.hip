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

// <BN 256, BK 64, 8 waves>: one workgroup per CU (128 KiB of LDS), 2 x 4 waves.  (A <BN 128, BK 32, 4 waves> form — TWO independent
// workgroups per CU, 48 KiB each, so that one's epilogue runs under the other's K-loop — was correct but 3-10 % slower on every DiT Linear,
// with or without a deliberate half-tile start offset between the two co-resident workgroups: 1.5x the L2 -> LDS bytes per flop)
// T: the operand type — bf16, or IEEE fp16 for the DiT block Linears of a handle that runs them in the reference's deployed dtype (DT_F16); the QKV
// epilogue writes q / k / V^T for the attention in bf16 either way
// MS: the MFMA shape of the K-loop — 16 (v_mfma_f32_16x16x32: 64 instructions of 16 cycles per K-tile and wave) or 32 (v_mfma_f32_32x32x16: 32 instructions of
// 32 cycles, the shape the matrix pipe sustains 12-15 % faster and that leaves twice the issue slots per matrix cycle for the fragment reads and the DMA; the
// accumulators are then 4 x 2 tiles of 32 x 32, the same 128 registers, and the epilogue stages them through acc_stage's second form).  Same sums per output
// element, another order of the k index inside a K-tile: results differ in the last fp32 bits of the accumulation only.
__device__ __forceinline__ f32x16_epi mfma32(const bf16x8& x, const bf16x8& y, const f32x16_epi& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0); }
__device__ __forceinline__ f32x16_epi mfma32(const f16x8& x, const f16x8& y, const f32x16_epi& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0); }

template <int EPI, int BN, int BK, int NWAVE, class T, int MS = 16>
__global__ __launch_bounds__(NWAVE * 64, 2) void gemm_big_kernel(GemmArgs a, int tiles_m, int tiles_n, int gw) {
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
    // column tiles in groups of gw (gw divides tiles_n): a group's W panels (gw x 256 rows x K) stay in the XCD's 4 MiB L2 while the row tiles go by
    const int per = tiles_m * a.batch * gw, sc = wgid / per, rr = wgid - sc * per;
    const int rt = rr / gw, tn = sc * gw + (rr - rt * gw);
    const int bz = rt / tiles_m, tm = rt - bz * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const T* __restrict__ Ab = reinterpret_cast<const T*>(a.A) + (long long)bz * a.a_bs;
    const T* __restrict__ Wb = reinterpret_cast<const T*>(a.W);

    // ---- LDS-DMA sources: QA instructions of A and QB of W per wave and K-tile, RPI rows each ------------------------------------------
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    // (a uniform base + one 32-bit byte offset per instruction: the accumulators and the two fragment sets leave no room for eight 64-bit
    // pointers.  Rows past M / N are clamped to the last row: their products land in accumulator rows / columns the epilogue never stores.)
    const int lrow = lane / SLOTS, lslot = lane % SLOTS;
    unsigned offA[QA], offW[QB];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int r = (wave * QA + q) * RPI + lrow;               // tile row filled by this lane
        const int chunk = lslot ^ ((r >> 1) & (SLOTS - 1));
        const int row = (m0 + r) < a.M ? (m0 + r) : a.M - 1;
        offA[q] = (unsigned)(((long long)row * a.lda + chunk * 8) * (long long)sizeof(T));
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int r = (wave * QB + q) * RPI + lrow;
        const int chunk = lslot ^ ((r >> 1) & (SLOTS - 1));
        const int row = (n0 + r) < a.N ? (n0 + r) : a.N - 1;
        offW[q] = (unsigned)(((long long)row * a.K + chunk * 8) * (long long)sizeof(T));
    }
    auto issue = [&](int kc, int buf) {
        const char* const pa = reinterpret_cast<const char*>(Ab) + (long long)kc * BK * sizeof(T);
        const char* const pw = reinterpret_cast<const char*>(Wb) + (long long)kc * BK * sizeof(T);
        T* const As = lds + buf * TILE_ELEMS;
        T* const Bs = As + BM * BK;
#pragma unroll
        for (int q = 0; q < QA; ++q) __builtin_amdgcn_global_load_lds((glb_ptr)(pa + offA[q]), (lds_ptr)(As + (wave * QA + q) * RPI * BK), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < QB; ++q) __builtin_amdgcn_global_load_lds((glb_ptr)(pw + offW[q]), (lds_ptr)(Bs + (wave * QB + q) * RPI * BK), 16, 0, 0);
    };

    const int nk = a.K / BK;
    if constexpr (MS == 32) {
        // ---- K-loop on 32 x 32 x 16 MFMAs: four k-steps of 16 per K-tile, 4 A + 2 B fragments (one ds_read_b128 each) feed the 8 MFMAs of a step; the two fragment
        // sets alternate, the barrier sits behind the third step: after it the first step of tile kc + 1 is read under the last step of tile kc -------------------
        static_assert(BK == 64 && MT == 8 && NT == 4, "written for the 128 x 64 wave tile");
        constexpr int MT2 = MT / 2, NT2 = NT / 2;
        f32x16_epi acc[MT2][NT2];
#pragma unroll
        for (int i = 0; i < MT2; ++i)
#pragma unroll
            for (int j = 0; j < NT2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        typedef typename Vec8<T>::type frag_t;
        const int c31 = lane & 31, hi = lane >> 5;
        const int sw = (c31 >> 1) & (SLOTS - 1);                     // ((row >> 1) & 7) of every fragment row of this lane (rows = 32 i + c31)
        frag_t a0[MT2], b0[NT2], a1[MT2], b1[NT2];
        auto loadf = [&](int buf, int step, frag_t (&af)[MT2], frag_t (&bf)[NT2]) __attribute__((always_inline)) {
            const int off = ((2 * step + hi) ^ sw) * 8;
            const T* const As = lds + buf * TILE_ELEMS + (wm0 + c31) * BK + off;
            const T* const Bs = lds + buf * TILE_ELEMS + BM * BK + (wn0 + c31) * BK + off;
#pragma unroll
            for (int j = 0; j < NT2; ++j) bf[j] = load8(Bs + j * 32 * BK);
#pragma unroll
            for (int i = 0; i < MT2; ++i) af[i] = load8(As + i * 32 * BK);
        };
        auto mfmas = [&](const frag_t (&af)[MT2], const frag_t (&bf)[NT2]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < MT2; ++i)
#pragma unroll
                for (int j = 0; j < NT2; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
        };
        auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < MT2 + NT2; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // one LDS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MT2 * NT2 - (MT2 + NT2), 0);
        };
        auto iter = [&](int kc, auto DMA, auto NEXT) __attribute__((always_inline)) {
            loadf(kc & 1, 1, a1, b1);
            mfmas(a0, b0);
            interleave();
            loadf(kc & 1, 2, a0, b0);
            mfmas(a1, b1);
            interleave();
            loadf(kc & 1, 3, a1, b1);
            mfmas(a0, b0);
            interleave();
            __syncthreads();                             // every wave has read all of tile kc (its buffer is free); tile kc + 1 has landed
            if constexpr (decltype(DMA)::value) issue(kc + 2, kc & 1);
            if constexpr (decltype(NEXT)::value) loadf((kc + 1) & 1, 0, a0, b0);
            mfmas(a1, b1);
            if constexpr (decltype(NEXT)::value) interleave();
        };
        issue(0, 0);
        __syncthreads();
        if (nk > 1) issue(1, 1);
        loadf(0, 0, a0, b0);
        for (int kc = 0; kc < nk - 2; ++kc) iter(kc, std::true_type{}, std::true_type{});
        if (nk >= 2) iter(nk - 2, std::false_type{}, std::true_type{});
        iter(nk - 1, std::false_type{}, std::false_type{});
        __syncthreads();                                 // the epilogue reuses the tile memory as staging
        gemm_epilogue<T, MT, NT, WN, EPI, 1, bf16_t>(a, acc, reinterpret_cast<float*>(smem) + wave * SCR_FLOATS, lane, m0 + wm0, n0 + wn0, bz, 0);
        return;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    // ---- K-loop, fragments double-buffered in REGISTERS.  The compiler's own schedule of the plain "read 12 fragments, 32 MFMAs" loop keeps two A
    // fragments live and waits on LDS every eight MFMAs (tools/gemm_lab.hip: 1.60 PF/s with no global traffic at all).  Here the twelve
    // ds_read_b128 of the NEXT k-step are issued one per two MFMAs of the current one (sched_group_barrier pins that order), and the barrier sits
    // between the two k-steps of a K-tile: after it, the first k-step of tile kc + 1 is read under the second k-step of tile kc, and tile kc + 2 is
    // requested into the buffer tile kc has just left.  Same loop without the epilogue: 2.0 PF/s from LDS alone, 1.5-1.68 PF/s with the DMA.
    static_assert(BK == 64, "two k-steps per K-tile");
    const int fr = lane & 15, fg = lane >> 4;
    const int sw = (fr >> 1) & (SLOTS - 1);                      // ((row >> 1) & (SLOTS - 1)) of every fragment row of this lane (rows = 16 i + fr)
    const int off0 = (fg ^ sw) * 8, off1 = ((4 + fg) ^ sw) * 8;
    typename Vec8<T>::type a0[MT], b0[NT], a1[MT], b1[NT];
    typedef typename Vec8<T>::type frag_t;
    auto loadf = [&](int buf, int off, frag_t (&af)[MT], frag_t (&bf)[NT]) __attribute__((always_inline)) {
        const T* const As = lds + buf * TILE_ELEMS + (wm0 + fr) * BK + off;
        const T* const Bs = lds + buf * TILE_ELEMS + BM * BK + (wn0 + fr) * BK + off;
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = load8(Bs + j * 16 * BK);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = load8(As + i * 16 * BK);
    };
    auto mfmas = [&](const frag_t (&af)[MT], const frag_t (&bf)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) mma32(acc[i][j], af[i], bf[j]);
    };
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < MT + NT; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);           // two MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // one LDS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - 2 * (MT + NT), 0);
    };
    auto iter = [&](int kc, auto DMA, auto NEXT) __attribute__((always_inline)) {
        loadf(kc & 1, off1, a1, b1);
        mfmas(a0, b0);
        interleave();
        __syncthreads();                                 // every wave has read all of tile kc (its buffer is free); tile kc + 1 has landed
        if constexpr (decltype(DMA)::value) issue(kc + 2, kc & 1);
        if constexpr (decltype(NEXT)::value) loadf((kc + 1) & 1, off0, a0, b0);
        mfmas(a1, b1);
        if constexpr (decltype(NEXT)::value) interleave();
    };
    issue(0, 0);
    __syncthreads();
    if (nk > 1) issue(1, 1);
    loadf(0, off0, a0, b0);
    for (int kc = 0; kc < nk - 2; ++kc) iter(kc, std::true_type{}, std::true_type{});
    if (nk >= 2) iter(nk - 2, std::false_type{}, std::true_type{});
    iter(nk - 1, std::false_type{}, std::false_type{});
    __syncthreads();                                     // the epilogue reuses the tile memory as staging

#if defined(HVX_LAB_GEMM_EPI) && HVX_LAB_GEMM_EPI == 1
    {   // (lab: the K-loop alone — every accumulator stays observable through one value per lane)
        f32x4 t = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) t += acc[i][j];
        if (t[0] + t[1] + t[2] + t[3] == 123.456f) reinterpret_cast<float*>(smem)[lane] = t[0];
        return;
    }
#elif defined(HVX_LAB_GEMM_EPI) && HVX_LAB_GEMM_EPI == 2
    {   // (lab: the whole epilogue's instructions, but only row 0 is stored and every residual row read is row 0: no memory traffic to speak of)
        GemmArgs a2 = a;
        a2.M = 1;
        gemm_epilogue<T, MT, NT, WN, EPI, 1, bf16_t>(a2, acc, reinterpret_cast<float*>(smem) + wave * SCR_FLOATS, lane, m0 + wm0, n0 + wn0, bz, 0);
        return;
    }
#endif
    gemm_epilogue<T, MT, NT, WN, EPI, 1, bf16_t>(a, acc, reinterpret_cast<float*>(smem) + wave * SCR_FLOATS, lane, m0 + wm0, n0 + wn0, bz, 0);
}

template <int BN, int BK, int NWAVE>
int launch_form(const GemmArgs& a, hipStream_t s, long long min_tiles) {
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const long long tiles = (long long)tiles_m * tiles_n * a.batch;
    if (tiles < min_tiles || tiles > 0x7fffffffLL) return 0;
    // Column groups of 4 tiles: the W panels of a group (4 x 256 rows x K bf16 = 2-4 MiB at K = 1024-2048) fit the XCD's L2; with all 8-12 column
    // tiles in one sweep W (4-6 MiB) is re-fetched through the fabric for every row tile (FETCH_SIZE 530 MB per QKV launch for 98 MB of operands).
    // Measured at M = 45056: N = 3072 330 -> 302 us, N = 2048 214 -> 204 us; N = 1024 has four column tiles anyway.
    const int gw_env = (int)opt(OPT_GEMM_BIG_GW);     // (tuning option; 0 = all columns in one group)
    int gw = tiles_n;
    if (gw_env > 0 && gw_env < tiles_n && tiles_n % gw_env == 0) gw = gw_env;
    const int slot = prof_begin(PK_GEMM, 2.0 * a.M * a.N * (double)a.K * a.batch, s);
    const dim3 grid((unsigned)tiles), block(NWAVE * 64);
    // (lab option gemm_big_mfma: 16 | 32.  The 32x32x16 K-loop is bit-identical and 4 % slower at flow level — profiles/r06_attn_tile_ab.md — and doubles this file's
    // compile time, so only lab builds carry it)
#ifdef HVX_LAB
    const bool m32 = opt(OPT_GEMM_BIG_MFMA) == 32;
#define HVX_BIG(EPI_, T_)                                                                                                            \
    do {                                                                                                                            \
        if (m32) hipLaunchKernelGGL((gemm_big_kernel<EPI_, BN, BK, NWAVE, T_, 32>), grid, block, 0, s, a, tiles_m, tiles_n, gw);    \
        else hipLaunchKernelGGL((gemm_big_kernel<EPI_, BN, BK, NWAVE, T_, 16>), grid, block, 0, s, a, tiles_m, tiles_n, gw);        \
    } while (0)
#else
#define HVX_BIG(EPI_, T_) hipLaunchKernelGGL((gemm_big_kernel<EPI_, BN, BK, NWAVE, T_, 16>), grid, block, 0, s, a, tiles_m, tiles_n, gw)
#endif
    if (a.dtype == DT_F16) {
        if (a.epi == EPI_GENERIC) HVX_BIG(EPI_GENERIC, f16_t);
        else HVX_BIG(EPI_QKV_DIT, f16_t);
    } else {
        if (a.epi == EPI_GENERIC) HVX_BIG(EPI_GENERIC, bf16_t);
        else HVX_BIG(EPI_QKV_DIT, bf16_t);
    }
#undef HVX_BIG
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 1 : (set_error("gemm (256-tile form) launch failed"), -1);
}

}  // namespace

// Eligible: bf16 plain Linear (one tap, one group, no stride / up-sampling / padding), K a multiple of 64, wide N and enough tiles to fill
// the chip.  Returns 1 when the launch was taken, 0 when the caller should use the generic tile forms, -1 on error.
int launch_gemm_big(const GemmArgs& a, hipStream_t s) {
    const bool shape_ok = !(a.groups != 1 || a.cin_pad != a.K || a.conv_stride != 1 || a.conv_dil != 1 || a.pad_left != 0 || a.up != 1 ||
                            a.rows_in != a.M || (a.K & 63) || a.N < 256 || (a.N & 63));
    if (a.dtype == DT_F16) {
        // fp16 operands exist in this form only (the DiT block Linears): every size takes it, however few tiles
        if (!shape_ok) return set_error("gemm: fp16 operands are supported for plain Linears with K %% 64 == 0, N >= 256, N %% 64 == 0 only"), -1;
        return launch_form<256, 64, 8>(a, s, 0);
    }
    if (a.dtype != DT_BF16 || !shape_ok) return 0;
    // below half a round of the 256 CUs the 128 x 128 form (2-3 workgroups per CU, four times as many tiles) balances better (measured: M = 11264,
    // N = 1024: 176 tiles tie; M = 2816: 44 tiles lose 177 vs 267 TF/s)
    const long long min_tiles = opt(OPT_GEMM_BIG_MIN_TILES);   // (tuning option)
    // (launch_form<128, 32, 4> — two workgroups per CU — measured 3-10 % slower on every DiT Linear at M = 45056: flow solve 450 vs 433 ms)
    return launch_form<256, 64, 8>(a, s, min_tiles);
}

}  // namespace hvx
