// gemm_tiled.hip — LDS-tiled implicit-GEMM on MFMA for gfx950 (see hvx_kernels.h: GemmArgs).
//
// Workgroup = 256 threads = 4 waves.  Block tile BM x BN, K-step 32.  Both operands are
// K-contiguous, so one 16-byte (bf16) / 32-byte (f32) vector per lane is exactly an MFMA
// fragment row.  bf16 tiles are staged by LDS-DMA (global_load_lds_dwordx4, double-buffered,
// swizzle on the source address); fp32 tiles go global -> registers -> LDS with the next
// K-step's loads issued before the current step's MFMAs and LDS rows padded (144 B) so that
// the 16 rows read by one ds_read_b128 lane group fall on 16 distinct 16-byte slots.  The conv index map (tap, dilation, stride, nearest-upsample,
// zero padding) lives in the A-tile loader; every epilogue variant is fused.
#include <stdlib.h>

#include <type_traits>

#include "hvx_device.h"
#include "hvx_kernels.h"
#include "gemm_epilogue.h"

namespace hvx {

template <class T> struct LdsPad { static constexpr int value = (sizeof(T) == 2) ? 8 : 4; };


template <class T, int BM, int BN, int WM, int WN, int EPI, int BK, int NBUF = 1, int GLDS = 0>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(GemmArgs a) {
    // BK = 64 (bf16, 144-byte LDS rows) halves the barriers and global-load round trips per flop; BK = 32 serves fp32 and the
    // convs whose padded channel count is not a multiple of 64 (a K-step must not straddle two taps)
    constexpr int LDK = GLDS ? BK : BK + LdsPad<T>::value;      // the LDS-DMA image is lane-linear: no padding, swizzled instead
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    constexpr int CPR = BK / 8;                      // 8-element chunks per tile row
    constexpr int RPP = 256 / CPR;                   // tile rows covered by one pass of the 256 threads
    constexpr int A_VECS = BM / RPP, B_VECS = BN / RPP;
    typedef typename Vec8<T>::type V8;

    // one LDS allocation: K-step tiles during the main loop, per-wave fp32 staging tiles in the epilogue
    constexpr int SLD = WN + 4;                      // staging row stride (floats)
    constexpr int ROWS_PASS = (64 / WN) * 16;        // rows a wave stages per pass: 64 lanes x 16 columns each
    constexpr int TILE_BYTES = NBUF * (BM + BN) * LDK * (int)sizeof(T);
    constexpr int SCR_BYTES = 4 * ROWS_PASS * SLD * 4;
    __shared__ __attribute__((aligned(16))) char smem[TILE_BYTES > SCR_BYTES ? TILE_BYTES : SCR_BYTES];
    T (*As)[BM * LDK] = reinterpret_cast<T (*)[BM * LDK]>(smem);
    T (*Bs)[BN * LDK] = reinterpret_cast<T (*)[BN * LDK]>(smem + NBUF * BM * LDK * sizeof(T));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int bz = blockIdx.z / a.groups, g = blockIdx.z % a.groups;

    const T* __restrict__ Ab = reinterpret_cast<const T*>(a.A) + (long long)bz * a.a_bs + (long long)g * a.a_gs;
    const T* __restrict__ Wb = reinterpret_cast<const T*>(a.W) + (long long)g * a.w_gs;

    // ---- per-thread tile-load coordinates ------------------------------------------------------
    int a_row[A_VECS], a_m[A_VECS];
    int b_row[B_VECS];
    const int chunk = tid % CPR;                    // which 8-element chunk of the K-step
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) {
        a_row[i] = (tid / CPR) + i * RPP;
        a_m[i] = m0 + a_row[i];
    }
#pragma unroll
    for (int i = 0; i < B_VECS; ++i) b_row[i] = (tid / CPR) + i * RPP;

    const int nk = a.K / BK;
    const long long in_span = (long long)a.rows_in * a.up;

    auto load_a = [&](int kc, V8 (&ra)[A_VECS]) {
        const int k0 = kc * BK;
        const int tap = k0 / a.cin_pad;
        const int ci = k0 - tap * a.cin_pad + chunk * 8;
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
            const long long idx = (long long)a_m[i] * a.conv_stride + (long long)tap * a.conv_dil - a.pad_left;
            if (a_m[i] < a.M && idx >= 0 && idx < in_span) {
                const long long src = (a.up == 1) ? idx : idx / a.up;
                ra[i] = load8(Ab + src * a.lda + ci);
            } else {
                ra[i] = zero8<T>();
            }
        }
    };
    auto load_b = [&](int kc, V8 (&rb)[B_VECS]) {
        const int k0 = kc * BK + chunk * 8;
#pragma unroll
        for (int i = 0; i < B_VECS; ++i) {
            const int n = n0 + b_row[i];
            if (n < a.N) rb[i] = load8(Wb + (long long)n * a.K + k0);
            else rb[i] = zero8<T>();
        }
    };
    auto stash = [&](int buf, const V8 (&ra)[A_VECS], const V8 (&rb)[B_VECS]) {
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) store8(&As[buf][a_row[i] * LDK + chunk * 8], ra[i]);
#pragma unroll
        for (int i = 0; i < B_VECS; ++i) store8(&Bs[buf][b_row[i] * LDK + chunk * 8], rb[i]);
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    V8 ra[A_VECS], rb[B_VECS];
    if constexpr (!GLDS) {
        load_a(0, ra);
        load_b(0, rb);
        stash(0, ra, rb);
    }
    const int fr = lane & 15, fg = lane >> 4;
    auto compute = [&](int cur) {
#pragma unroll
        for (int kk = 0; kk < BK; kk += 32) {
            V8 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = load8(&As[cur][(wm0 + i * 16 + fr) * LDK + kk + fg * 8]);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = load8(&Bs[cur][(wn0 + j * 16 + fr) * LDK + kk + fg * 8]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) mma32(acc[i][j], af[i], bf[j]);
        }
    };
    if constexpr (GLDS) {
        // LDS-DMA staging (global_load_lds_dwordx4): no staging registers, no ds_write pass.  A wave instruction deposits its 64 lanes'
        // 16 bytes back to back, so the LDS image is lane-linear: [row][4 chunks of 8 bf16], 16 rows per instruction, and the bank
        // spread that padding gives the register form comes from a swizzle applied to the SOURCE address instead: the lane that
        // fills slot s of row r fetches chunk s ^ f(r), f(r) = (-(r >> 2)) & 3, which puts the 16 rows of every ds_read_b128 lane
        // group on 16 distinct 16-byte slots; the four lanes of a row still fetch the same 64 contiguous bytes.
        static_assert(!GLDS || (sizeof(T) == 2 && BK == 32 && NBUF == 2 && BM % 32 == 0 && BN % 32 == 0), "LDS-DMA form: bf16, BK = 32");
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* glb_ptr;
        const int lrow = lane >> 2, lslot = lane & 3;
        const int gchunk = lslot ^ ((-(lrow >> 2)) & 3);
        constexpr int A_INS = BM / 64, B_INS = BN / 64;            // instructions per wave and tile: 16 rows each, 4 waves
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
                const void* gp = ok ? static_cast<const void*>(Ab + src * a.lda + ci) : static_cast<const void*>(g_zero_row);
                __builtin_amdgcn_global_load_lds((glb_ptr)gp, (lds_ptr)(&As[buf][r16 * LDK]), 16, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < B_INS; ++q) {
                const int r16 = (wave * B_INS + q) * 16;
                const int n = n0 + r16 + lrow;
                const void* gp = n < a.N ? static_cast<const void*>(Wb + (long long)n * a.K + k0 + gchunk * 8) : static_cast<const void*>(g_zero_row);
                __builtin_amdgcn_global_load_lds((glb_ptr)gp, (lds_ptr)(&Bs[buf][r16 * LDK]), 16, 0, 0);
            }
        };
        const int fsw = (fg ^ ((-(fr >> 2)) & 3)) * 8;               // slot of this lane's fragment chunk in its row
        auto compute_swz = [&](int cur) {
            V8 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = load8(&As[cur][(wm0 + i * 16 + fr) * LDK + fsw]);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = load8(&Bs[cur][(wn0 + j * 16 + fr) * LDK + fsw]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) mma32(acc[i][j], af[i], bf[j]);
        };
        issue(0, 0);
        for (int kc = 0; kc < nk; ++kc) {
            __syncthreads();                                 // tile kc has landed (the barrier drains the DMA queue); buffer (kc+1)&1 is free
            if (kc + 1 < nk) issue(kc + 1, (kc + 1) & 1);
            compute_swz(kc & 1);
        }
        __syncthreads();                                     // the epilogue reuses the tile memory as staging
    } else if constexpr (NBUF == 2) {
        // One barrier per K-step, one register set: after the barrier (every wave has finished reading the buffer about to be
        // overwritten, and the tile staged last step is visible) the registers holding tile t+1 go to LDS and are re-issued at once
        // for tile t+2, whose global loads then have the whole MFMA phase of tile t and the next barrier to land.
        if (nk > 1) {
            load_a(1, ra);
            load_b(1, rb);
        }
        for (int kc = 0; kc < nk; ++kc) {
            __syncthreads();
            if (kc + 1 < nk) {
                stash((kc + 1) & 1, ra, rb);
                if (kc + 2 < nk) {
                    load_a(kc + 2, ra);
                    load_b(kc + 2, rb);
                }
            }
            compute(kc & 1);
        }
        __syncthreads();                                 // the epilogue reuses the tile memory as staging
    } else {
        __syncthreads();
        for (int kc = 0; kc < nk; ++kc) {
            const bool more = (kc + 1) < nk;
            if (more) {
                load_a(kc + 1, ra);
                load_b(kc + 1, rb);
            }
            compute(0);
            __syncthreads();
            if (more) {
                stash(0, ra, rb);
                __syncthreads();
            }
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------------
    gemm_epilogue<T, MT, NT, WN, EPI>(a, acc, reinterpret_cast<float*>(smem) + wave * ROWS_PASS * SLD, lane, m0 + wm0, n0 + wn0, bz, g);
}

template <class T, int BM, int BN, int WM, int WN, int BK, int NBUF = 1, int GLDS = 0>
static int launch_cfg_bk(const GemmArgs& a, hipStream_t s) {
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.batch * a.groups);
    const int slot = prof_begin(sizeof(T) == 2 ? PK_GEMM : PK_GEMM_F32, 2.0 * a.M * a.N * (double)a.K * a.batch * a.groups, s);
    if (a.epi == EPI_GENERIC)
        hipLaunchKernelGGL((gemm_tiled_kernel<T, BM, BN, WM, WN, EPI_GENERIC, BK, NBUF, GLDS>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_tiled_kernel<T, BM, BN, WM, WN, EPI_QKV_DIT, BK, NBUF, GLDS>), grid, dim3(256), 0, s, a);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("gemm launch failed"), -1);
}

template <class T, int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmArgs& a, hipStream_t s) {
    // K-loop forms, measured on MI355X with tools/bench_ops.py on the DiT shapes (M = 11264, K = 1024..2048, 128x128 tiles):
    //   registers -> LDS, two barriers per K-step            470-550 TF/s
    //   registers -> LDS, double buffer, one barrier          480-575 TF/s   (write after the barrier, re-issue at once)
    //   LDS-DMA (global_load_lds), double buffer              530-650 TF/s   <- bf16 default
    //   BK = 64 with either register form                     340-460 TF/s   (LDS footprint halves the resident workgroups)
    // fp32 (HiFT shapes): the two-barrier register form is the fastest of the three (the double buffer costs 10-25 % there).
    // (only the chosen form of each dtype is instantiated: every extra one costs a minute of build time over the tile configurations)
    if constexpr (sizeof(T) == 2) return launch_cfg_bk<T, BM, BN, WM, WN, 32, 2, 1>(a, s);
    else return launch_cfg_bk<T, BM, BN, WM, WN, 32>(a, s);
}

template <class T>
static int launch_t(const GemmArgs& a, hipStream_t s) {
    // tile choice: wide N -> 128x128; narrow N (<= 64 per group) -> 128x64; small problems -> 64x64 for more workgroups
    const long long blocks128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch * a.groups;
    if (a.N <= 64) {
        if (a.M <= 4096) return launch_cfg<T, 64, 64, 32, 32>(a, s);
        return launch_cfg<T, 128, 64, 32, 64>(a, s);
    }
    if (blocks128 < 256) return launch_cfg<T, 64, 64, 32, 32>(a, s);
    // (256x128 tiles with 128x64 per wave — a third less LDS traffic per flop — were measured on the DiT shapes, M = 11264: 380-540 TF/s
    // against 535-657 TF/s for 128x128; the 128 accumulator registers per lane leave two waves per SIMD)
    return launch_cfg<T, 128, 128, 64, 64>(a, s);
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return 0;
    if (a.K <= 0 || (a.K & 31) || a.cin_pad <= 0 || (a.cin_pad & 31) || (a.K % a.cin_pad) || (a.lda & 7) || a.up < 1 || a.groups < 1) {
        set_error("launch_gemm: bad geometry M=%d N=%d K=%d cin_pad=%d lda=%d up=%d", a.M, a.N, a.K, a.cin_pad, a.lda, a.up);
        return -1;
    }
    if (a.epi == EPI_QKV_DIT && (a.N != 3 * a.heads * 64 || (a.t_pad & 3) || a.groups != 1)) {
        set_error("launch_gemm: bad QKV geometry N=%d heads=%d t_pad=%d", a.N, a.heads, a.t_pad);
        return -1;
    }
    const int x3 = launch_gemm_x3(a, s);                   // fp32 operands split into bf16 pairs (gemm_x3.hip), where the caller allows it
    if (x3) return x3 < 0 ? -1 : 0;
    const int resident = (int)opt(OPT_CONV_RESIDENT);      // (A/B option conv_resident)
    const int rc = resident ? launch_conv_resident(a, s) : 0;   // 64-channel-per-group bf16 convolutions with their input rows resident in LDS
    if (rc) return rc < 0 ? -1 : 0;
    const int big = launch_gemm_big(a, s);                 // the 256 x 256 tile form takes the large bf16 Linears (gemm_big.hip)
    if (big) return big < 0 ? -1 : 0;
    return a.dtype == DT_BF16 ? launch_t<bf16_t>(a, s) : launch_t<float>(a, s);
}

}  // namespace hvx
