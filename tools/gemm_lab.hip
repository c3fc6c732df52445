// gemm_lab.hip — where does the K-loop of gemm_big_kernel (csrc/gemm_big.hip) spend its time?  A standalone bench of the 256 x 256 x 64 tile
// loop with the epilogue reduced to one float per lane, in variants that remove one ingredient each (development aid, not the product):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -o tools/bin/gemm_lab tools/gemm_lab.hip
//   tools/bin/gemm_lab [M]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, WM = 128, WN = 64, MT = WM / 16, NT = WN / 16, NWAVE = 8;

enum { F_DMA = 1, F_COMPUTE = 2, F_SAMEK = 4, F_PRIO = 8 };

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

// ---- form 1: the product loop — two K-tiles of 64 in LDS, one barrier per K-tile -----------------------------------------------------
template <int FLAGS>
__global__ __launch_bounds__(512, 2) void lab2_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* sink, int M, int N, int K,
                                                      int tiles_m, int tiles_n) {
    constexpr int BK = 64, TILE_ELEMS = (BM + BN) * BK, SLOTS = 8, RPI = 8, QA = BM / RPI / NWAVE, QB = BN / RPI / NWAVE;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE_ELEMS * 2];
    bf16_t* const lds = reinterpret_cast<bf16_t*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / 4) * WM, wn0 = (wave % 4) * WN;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qn = nwg >> 3, rn = nwg & 7;
    const int wgid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int tn = wgid % tiles_n, tm = wgid / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int lrow = lane / SLOTS, lslot = lane % SLOTS;
    const char* srcA[QA];
    const char* srcW[QB];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int r = (wave * QA + q) * RPI + lrow;
        const int chunk = lslot ^ ((r >> 1) & 7);
        int row = m0 + r;
        if (row >= M) row = M - 1;
        srcA[q] = reinterpret_cast<const char*>(A + (long long)row * K + chunk * 8);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int r = (wave * QB + q) * RPI + lrow;
        const int chunk = lslot ^ ((r >> 1) & 7);
        srcW[q] = reinterpret_cast<const char*>(W + (long long)(n0 + r) * K + chunk * 8);
    }
    auto issue = [&](int kc, int buf) {
        if (!(FLAGS & F_DMA)) return;
        const long long kb = (FLAGS & F_SAMEK) ? 0 : (long long)kc * BK * 2;
        bf16_t* const As = lds + buf * TILE_ELEMS;
        bf16_t* const Bs = As + BM * BK;
#pragma unroll
        for (int q = 0; q < QA; ++q) __builtin_amdgcn_global_load_lds((glb_ptr)(srcA[q] + kb), (lds_ptr)(As + (wave * QA + q) * RPI * BK), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < QB; ++q) __builtin_amdgcn_global_load_lds((glb_ptr)(srcW[q] + kb), (lds_ptr)(Bs + (wave * QB + q) * RPI * BK), 16, 0, 0);
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int fr = lane & 15, fg = lane >> 4, sw = (fr >> 1) & 7;
    auto compute = [&](int buf) {
        if (!(FLAGS & F_COMPUTE)) return;
        const bf16_t* const As = lds + buf * TILE_ELEMS + (wm0 + fr) * BK;
        const bf16_t* const Bs = lds + buf * TILE_ELEMS + BM * BK + (wn0 + fr) * BK;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int off = ((kk * 4 + fg) ^ sw) * 8;
            bf16x8 af[MT], bf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 16 * BK + off);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + i * 16 * BK + off);
            if (FLAGS & F_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            if (FLAGS & F_PRIO) __builtin_amdgcn_s_setprio(0);
        }
    };
    const int nk = K / BK;
    issue(0, 0);
    for (int kc = 0; kc < nk; ++kc) {
        __syncthreads();
        if (kc + 1 < nk) issue(kc + 1, (kc + 1) & 1);
        compute(kc & 1);
    }
    f32x4 s = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) s += acc[i][j];
    sink[(long long)blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}

// ---- form 2: a ring of NS K-tiles of 32 (NS x 32 KiB), tile kc + NS - 1 issued while kc is computed: NS - 1 tiles in flight, counted waits ----
// K-tile rows are 64 B: one DMA instruction deposits 16 rows; swizzle chunk ^ ((row >> 1) & 3).
template <int NS, int FLAGS>
__global__ __launch_bounds__(512, 2) void labring_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* sink, int M, int N, int K,
                                                         int tiles_m, int tiles_n) {
    constexpr int BK = 32, TILE_ELEMS = (BM + BN) * BK, SLOTS = 4, RPI = 16, QA = BM / RPI / NWAVE, QB = BN / RPI / NWAVE;   // 2 + 2 instructions per wave and tile
    __shared__ __attribute__((aligned(16))) char smem[NS * TILE_ELEMS * 2];
    bf16_t* const lds = reinterpret_cast<bf16_t*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / 4) * WM, wn0 = (wave % 4) * WN;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qn = nwg >> 3, rn = nwg & 7;
    const int wgid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int tn = wgid % tiles_n, tm = wgid / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int lrow = lane / SLOTS, lslot = lane % SLOTS;
    const char* srcA[QA];
    const char* srcW[QB];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int r = (wave * QA + q) * RPI + lrow;
        const int chunk = lslot ^ ((r >> 1) & 3);
        int row = m0 + r;
        if (row >= M) row = M - 1;
        srcA[q] = reinterpret_cast<const char*>(A + (long long)row * K + chunk * 8);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int r = (wave * QB + q) * RPI + lrow;
        const int chunk = lslot ^ ((r >> 1) & 3);
        srcW[q] = reinterpret_cast<const char*>(W + (long long)(n0 + r) * K + chunk * 8);
    }
    auto issue = [&](int kc, int buf) {
        const long long kb = (FLAGS & F_SAMEK) ? 0 : (long long)kc * BK * 2;
        bf16_t* const As = lds + buf * TILE_ELEMS;
        bf16_t* const Bs = As + BM * BK;
#pragma unroll
        for (int q = 0; q < QA; ++q) __builtin_amdgcn_global_load_lds((glb_ptr)(srcA[q] + kb), (lds_ptr)(As + (wave * QA + q) * RPI * BK), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < QB; ++q) __builtin_amdgcn_global_load_lds((glb_ptr)(srcW[q] + kb), (lds_ptr)(Bs + (wave * QB + q) * RPI * BK), 16, 0, 0);
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int fr = lane & 15, fg = lane >> 4, sw = (fr >> 1) & 3;
    auto compute = [&](int buf) {
        if (!(FLAGS & F_COMPUTE)) return;
        const bf16_t* const As = lds + buf * TILE_ELEMS + (wm0 + fr) * BK;
        const bf16_t* const Bs = lds + buf * TILE_ELEMS + BM * BK + (wn0 + fr) * BK;
        const int off = (fg ^ sw) * 8;
        bf16x8 af[MT], bf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 16 * BK + off);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + i * 16 * BK + off);
        if (FLAGS & F_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        if (FLAGS & F_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    const int nk = K / BK;
    constexpr int PER = QA + QB;                             // DMA instructions per wave and tile
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t, t);
    for (int kc = 0; kc < nk; ++kc) {
        // tile kc landed: at most (NS - 2) younger tiles of this wave may still be in flight (the tail issues nothing: wait for all)
        if (kc + NS - 1 <= nk) __builtin_amdgcn_s_waitcnt(0x0F70 | ((NS - 2) * PER));      // vmcnt((NS-2)*PER), lgkm/exp untouched
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();                        // every wave's part of tile kc landed; everybody is done with tile kc - 1
        if (kc + NS - 1 < nk) issue(kc + NS - 1, (kc + NS - 1) % NS);
        compute(kc % NS);
    }
    f32x4 s = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) s += acc[i][j];
    sink[(long long)blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}

// ---- form 3: the same two K-tiles of 64, fragments double-buffered in registers: the ds_reads of the NEXT k-step (of the next tile after the
// barrier) are interleaved one per two MFMAs with the current k-step's MFMAs; the barrier sits between the two k-steps of a tile ----
template <int FLAGS>
__global__ __launch_bounds__(512, 2) void lab3_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* sink, int M, int N, int K,
                                                      int tiles_m, int tiles_n) {
    constexpr int BK = 64, TILE_ELEMS = (BM + BN) * BK, SLOTS = 8, RPI = 8, QA = BM / RPI / NWAVE, QB = BN / RPI / NWAVE;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE_ELEMS * 2];
    bf16_t* const lds = reinterpret_cast<bf16_t*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / 4) * WM, wn0 = (wave % 4) * WN;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qn = nwg >> 3, rn = nwg & 7;
    const int wgid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int tn = wgid % tiles_n, tm = wgid / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int lrow = lane / SLOTS, lslot = lane % SLOTS;
    const char* srcA[QA];
    const char* srcW[QB];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int r = (wave * QA + q) * RPI + lrow;
        const int chunk = lslot ^ ((r >> 1) & 7);
        int row = m0 + r;
        if (row >= M) row = M - 1;
        srcA[q] = reinterpret_cast<const char*>(A + (long long)row * K + chunk * 8);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int r = (wave * QB + q) * RPI + lrow;
        const int chunk = lslot ^ ((r >> 1) & 7);
        srcW[q] = reinterpret_cast<const char*>(W + (long long)(n0 + r) * K + chunk * 8);
    }
    auto issue = [&](int kc, int buf) {
        if (!(FLAGS & F_DMA)) return;
        const long long kb = (FLAGS & F_SAMEK) ? 0 : (long long)kc * BK * 2;
        bf16_t* const As = lds + buf * TILE_ELEMS;
        bf16_t* const Bs = As + BM * BK;
#pragma unroll
        for (int q = 0; q < QA; ++q) __builtin_amdgcn_global_load_lds((glb_ptr)(srcA[q] + kb), (lds_ptr)(As + (wave * QA + q) * RPI * BK), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < QB; ++q) __builtin_amdgcn_global_load_lds((glb_ptr)(srcW[q] + kb), (lds_ptr)(Bs + (wave * QB + q) * RPI * BK), 16, 0, 0);
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int fr = lane & 15, fg = lane >> 4, sw = (fr >> 1) & 7;
    const int off0 = ((0 * 4 + fg) ^ sw) * 8, off1 = ((1 * 4 + fg) ^ sw) * 8;
    bf16x8 a0[MT], b0[NT], a1[MT], b1[NT];
    auto loadf = [&](int buf, int off, bf16x8 (&af)[MT], bf16x8 (&bf)[NT]) __attribute__((always_inline)) {
        const bf16_t* const As = lds + buf * TILE_ELEMS + (wm0 + fr) * BK + off;
        const bf16_t* const Bs = lds + buf * TILE_ELEMS + BM * BK + (wn0 + fr) * BK + off;
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 16 * BK);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + i * 16 * BK);
    };
    auto mfmas = [&](const bf16x8 (&af)[MT], const bf16x8 (&bf)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < MT + NT; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - 2 * (MT + NT), 0);
    };
    const int nk = K / BK;
    issue(0, 0);
    __syncthreads();
    if (nk > 1) issue(1, 1);
    loadf(0, off0, a0, b0);
    auto iter = [&](int kc, auto DMA, auto NEXT) __attribute__((always_inline)) {
        if (FLAGS & F_COMPUTE) {
            loadf(kc & 1, off1, a1, b1);
            mfmas(a0, b0);
            interleave();
        }
        __syncthreads();                                 // every wave has read all of tile kc; tile kc + 1 has landed
        if constexpr (decltype(DMA)::value) issue(kc + 2, kc & 1);
        if (FLAGS & F_COMPUTE) {
            if constexpr (decltype(NEXT)::value) loadf((kc + 1) & 1, off0, a0, b0);
            mfmas(a1, b1);
            if constexpr (decltype(NEXT)::value) interleave();
        }
    };
    for (int kc = 0; kc < nk - 2; ++kc) iter(kc, std::true_type{}, std::true_type{});
    if (nk >= 2) iter(nk - 2, std::false_type{}, std::true_type{});
    iter(nk - 1, std::false_type{}, std::false_type{});
    f32x4 s = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) s += acc[i][j];
    sink[(long long)blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}

// ---- form 4: 256 x 128 tiles, FOUR waves (2 x 2 of 128 x 64), two workgroups per CU; K-tiles of 32 in a ring of NS, fragments double-buffered
// in registers: iteration kc reads the fragments of tile kc + 1 under the MFMAs of tile kc; tile kc + NS - 1 is requested right after the barrier ----
template <int NS, int FLAGS>
__global__ __launch_bounds__(256, 2) void lab4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* sink, int M, int N, int K,
                                                      int tiles_m, int tiles_n) {
    constexpr int BN4 = 128, NW = 4, BK = 32, TILE_ELEMS = (BM + BN4) * BK, SLOTS = 4, RPI = 16, QA = BM / RPI / NW, QB = BN4 / RPI / NW;   // 4 + 2 per wave
    __shared__ __attribute__((aligned(16))) char smem[NS * TILE_ELEMS * 2];
    bf16_t* const lds = reinterpret_cast<bf16_t*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / 2) * WM, wn0 = (wave % 2) * WN;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qn = nwg >> 3, rn = nwg & 7;
    const int wgid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int tn = wgid % tiles_n, tm = wgid / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN4;
    const int lrow = lane / SLOTS, lslot = lane % SLOTS;
    const char* srcA[QA];
    const char* srcW[QB];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int r = (wave * QA + q) * RPI + lrow;
        const int chunk = lslot ^ ((r >> 1) & 3);
        int row = m0 + r;
        if (row >= M) row = M - 1;
        srcA[q] = reinterpret_cast<const char*>(A + (long long)row * K + chunk * 8);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int r = (wave * QB + q) * RPI + lrow;
        const int chunk = lslot ^ ((r >> 1) & 3);
        srcW[q] = reinterpret_cast<const char*>(W + (long long)(n0 + r) * K + chunk * 8);
    }
    auto issue = [&](int kc, int buf) {
        if (!(FLAGS & F_DMA)) return;
        const long long kb = (long long)kc * BK * 2;
        bf16_t* const As = lds + buf * TILE_ELEMS;
        bf16_t* const Bs = As + BM * BK;
#pragma unroll
        for (int q = 0; q < QA; ++q) __builtin_amdgcn_global_load_lds((glb_ptr)(srcA[q] + kb), (lds_ptr)(As + (wave * QA + q) * RPI * BK), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < QB; ++q) __builtin_amdgcn_global_load_lds((glb_ptr)(srcW[q] + kb), (lds_ptr)(Bs + (wave * QB + q) * RPI * BK), 16, 0, 0);
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int fr = lane & 15, fg = lane >> 4, sw = (fr >> 1) & 3;
    const int off = (fg ^ sw) * 8;
    bf16x8 a0[MT], b0[NT], a1[MT], b1[NT];
    auto loadf = [&](int buf, bf16x8 (&af)[MT], bf16x8 (&bf)[NT]) __attribute__((always_inline)) {
        const bf16_t* const As = lds + buf * TILE_ELEMS + (wm0 + fr) * BK + off;
        const bf16_t* const Bs = lds + buf * TILE_ELEMS + BM * BK + (wn0 + fr) * BK + off;
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 16 * BK);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + i * 16 * BK);
    };
    auto mfmas = [&](const bf16x8 (&af)[MT], const bf16x8 (&bf)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < MT + NT; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - 2 * (MT + NT), 0);
    };
    const int nk = K / BK;                                   // even, >= NS
    constexpr int PER = QA + QB;
    // half-iteration: tile kc is in (af, bf); read tile kc + 1 into (an, bn) under its MFMAs
    auto half = [&](int kc, bf16x8 (&af)[MT], bf16x8 (&bf)[NT], bf16x8 (&an)[MT], bf16x8 (&bn)[NT]) __attribute__((always_inline)) {
        // tile kc + 1 landed: only the (NS - 2) youngest batches of this wave may still be in flight (near the tail fewer were issued: wait for all)
        if (kc + NS - 1 <= nk) __builtin_amdgcn_s_waitcnt(0x0070 | ((NS - 2) * PER));      // vmcnt(..) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        if (kc + NS - 1 < nk) issue(kc + NS - 1, (kc + NS - 1) % NS);
        if (FLAGS & F_COMPUTE) {
            if (kc + 1 < nk) loadf((kc + 1) % NS, an, bn);
            mfmas(af, bf);
            interleave();
        }
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t, t);
    __builtin_amdgcn_s_waitcnt(0x0070 | ((NS - 2) * PER));
    __builtin_amdgcn_s_barrier();
    loadf(0, a0, b0);
    for (int kc = 0; kc < nk; kc += 2) {
        half(kc, a0, b0, a1, b1);
        half(kc + 1, a1, b1, a0, b0);
    }
    f32x4 s = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) s += acc[i][j];
    sink[(long long)blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

__global__ void fill_random(unsigned short* p, long long n, unsigned seed) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u + seed * 40503u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (unsigned short)((x & 0x80ffu) | 0x3f00u);          // sign, exponent 0x7e/0x7f, 7 mantissa bits: |v| in [0.5, 2)
    }
}

template <class F>
static float time_us(F launch, int reps = 20) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("  (launch error)\n");
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 45056;
    const int Kmax = 2048, Nmax = 3072;
    bf16_t *A, *W;
    float* sink;
    hipMalloc(&A, (size_t)M * Kmax * 2);
    hipMalloc(&W, (size_t)Nmax * Kmax * 2);
    hipMalloc(&sink, (size_t)8192 * 512 * 4);
    {   // random bf16 in (-2, 2): the matrix cores' power draw (and with it the clock) depends on the operand bits
        const char* cst = getenv("LAB_CONST");
        if (cst) { hipMemset(A, 0x3c, (size_t)M * Kmax * 2); hipMemset(W, 0x3c, (size_t)Nmax * Kmax * 2); }
        else {
            hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned short*>(A), (long long)M * Kmax, 1u);
            hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned short*>(W), (long long)Nmax * Kmax, 2u);
        }
    }
    const int shapes[][2] = {{1024, 1024}, {1024, 2048}, {3072, 1024}};
    for (auto& sh : shapes) {
        const int N = sh[0], K = sh[1];
        const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN, tiles = tiles_m * tiles_n;
        const double fl = 2.0 * M * N * K;
        printf("M=%d N=%d K=%d  (%d tiles = %.2f rounds of 256; MFMA floor %.1f us)\n", M, N, K, tiles, tiles / 256.0, fl / 2.5e15 * 1e6);
#define RUN(name, kern)                                                                                                  \
    {                                                                                                                    \
        const float us = time_us([&] { hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), 0, 0, A, W, sink, M, N, K, tiles_m, tiles_n); }); \
        printf("  %-52s %8.1f us  %7.1f TF/s\n", name, us, fl / us / 1e6);                                               \
    }
        RUN("2 x 64: DMA + MFMA (the product loop)", (lab2_kernel<F_DMA | F_COMPUTE>));
        RUN("2 x 64: DMA + MFMA, s_setprio around the MFMAs", (lab2_kernel<F_DMA | F_COMPUTE | F_PRIO>));
        RUN("2 x 64: MFMA only (no DMA)", (lab2_kernel<F_COMPUTE>));
        RUN("2 x 64: DMA only (no fragment reads, no MFMA)", (lab2_kernel<F_DMA>));
        RUN("2 x 64: DMA of one L2-hot K-tile + MFMA", (lab2_kernel<F_DMA | F_COMPUTE | F_SAMEK>));
        RUN("2 x 64 register-pipelined: DMA + MFMA", (lab3_kernel<F_DMA | F_COMPUTE>));
        RUN("2 x 64 register-pipelined: MFMA only", (lab3_kernel<F_COMPUTE>));
        RUN("2 x 64 register-pipelined: L2-hot tile + MFMA", (lab3_kernel<F_DMA | F_COMPUTE | F_SAMEK>));
#define RUN4(name, kern)                                                                                                 \
    {                                                                                                                    \
        const int t4 = tiles_m * (N / 128);                                                                              \
        const float us = time_us([&] { hipLaunchKernelGGL(kern, dim3(t4), dim3(256), 0, 0, A, W, sink, M, N, K, tiles_m, N / 128); }); \
        printf("  %-52s %8.1f us  %7.1f TF/s\n", name, us, fl / us / 1e6);                                               \
    }
        RUN4("256x128, 4 waves, 2 WG/CU, ring 3 x 32: DMA + MFMA", (lab4_kernel<3, F_DMA | F_COMPUTE>));
        RUN4("256x128, 4 waves, 2 WG/CU, ring 3 x 32: MFMA only", (lab4_kernel<3, F_COMPUTE>));
        RUN4("256x128, 4 waves, 2 WG/CU, ring 3 x 32: DMA only", (lab4_kernel<3, F_DMA>));
        RUN("ring 3 x 32: DMA + MFMA", (labring_kernel<3, F_DMA | F_COMPUTE>));
        RUN("ring 4 x 32: DMA + MFMA", (labring_kernel<4, F_DMA | F_COMPUTE>));
        RUN("ring 4 x 32: DMA + MFMA, s_setprio", (labring_kernel<4, F_DMA | F_COMPUTE | F_PRIO>));
        RUN("ring 4 x 32: DMA only", (labring_kernel<4, F_DMA>));
        RUN("ring 4 x 32: L2-hot tile + MFMA", (labring_kernel<4, F_DMA | F_COMPUTE | F_SAMEK>));
    }
    return 0;
}
