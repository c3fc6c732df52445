// conv_resident.hip — bf16 convolutions over 64 input channels per group with the INPUT ROWS RESIDENT IN LDS (gfx950): the DiT's causal grouped
// position-embedding convolutions (flow/DiT/modules.py:115-144: two Conv1d(1024, 1024, k = 31, groups = 16) + Mish; 64 channels per group).
//
// Read through the tiled GEMM a k-tap convolution is K = k x 64: the workgroup fetches its 128 input rows 31 times (tap t is the tile shifted by t
// rows), one 32-wide K-step at a time with a barrier and 8 MFMAs per wave each — 62 barriers per tile, 405 TF/s at M = 45056.  Here (the bf16, grouped
// and batched sibling of gemm_x3.hip's conv64_x3p_kernel) the workgroup's 128 output rows + their (k - 1) x dil halo are brought into LDS ONCE by
// LDS-DMA, every tap reads its fragments from that image at a row offset, and only the weights are staged per step, two taps (16 KB) at a time,
// double-buffered: 16 barriers per tile with 32 MFMAs per wave between them.  Same epilogues as every other GEMM form (gemm_epilogue.h).
//
// LDS images are [k-half][row][32 bf16] (64-byte rows, lane-linear as the DMA deposits them); the lane that fills 16-byte slot s of row r fetches chunk
// s ^ f(r), f(r) = (-(r >> 2)) & 3 (r within its 16-row group), which puts the 16 lanes of every ds_read_b128 lane group on 16 distinct slots.
#include "gemm_epilogue.h"

namespace hvx {

namespace {

template <int MAXR, int TPS>
__global__ __launch_bounds__(256) void gconv64_kernel(GemmArgs a) {
    typedef bf16_t T;
    constexpr int BM = 128, MT = 2, NT = 4, WN = 64;
    constexpr int SLD = WN + 4, ROWS_PASS = 16;
    constexpr int AIMG = MAXR * 32;                        // one k-half image of the input rows: [MAXR][32]
    constexpr int BIMG = 64 * 32;                          // one k-half image of a tap's weights: [64][32]
    constexpr int SCR_BYTES = 4 * ROWS_PASS * SLD * 4;
    static_assert(2 * TPS * 2 * BIMG * 2 >= SCR_BYTES, "the epilogue staging fits the weight buffers");
    static_assert(TPS * 2 * 4 == 16, "16 weight DMA instructions per stage, 4 per wave");
    __shared__ __attribute__((aligned(16))) T As[2 * AIMG];             // [half][row][32]
    __shared__ __attribute__((aligned(16))) T Bs[2 * TPS * 2 * BIMG];   // [buffer][tap of the stage][half][n][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM;
    const int bz = blockIdx.z / a.groups, g = blockIdx.z - bz * a.groups;
    const int taps = a.K / 64;
    const int rows_tile = BM + (taps - 1) * a.conv_dil;               // <= MAXR (checked by the launcher)
    const T* __restrict__ Ab = reinterpret_cast<const T*>(a.A) + (long long)bz * a.a_bs + (long long)g * a.a_gs;
    const T* __restrict__ Wb = reinterpret_cast<const T*>(a.W) + (long long)g * a.w_gs;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int lrow = lane >> 2, lslot = lane & 3;
    const int gchunk = lslot ^ ((-(lrow >> 2)) & 3);
    // ---- input rows: tile row j = row m0 + j - pad_left of this (batch entry, group) ---------------------------------------------------------
    const int groups16 = (rows_tile + 15) >> 4;
    for (int q = wave; q < 2 * groups16; q += 4) {
        const int half = q / groups16, g16 = q - half * groups16;
        const long long grow = (long long)m0 + g16 * 16 + lrow - a.pad_left;
        const bool ok = grow >= 0 && grow < a.rows_in && (g16 * 16 + lrow) < rows_tile;
        const T* gp = Ab + grow * a.lda + half * 32 + gchunk * 8;
        const void* src = ok ? static_cast<const void*>(gp) : static_cast<const void*>(g_zero_row);
        __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(As + half * AIMG + g16 * 16 * 32), 16, 0, 0);
    }
    const int nst = (taps + TPS - 1) / TPS;
    auto issue_w = [&](int st, int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = wave * 4 + u;                            // (tap of the stage, half, 16-row group)
            const int ts = idx >> 3, half = (idx >> 2) & 1, g16 = idx & 3;
            const int tap = st * TPS + ts;
            const int n = g16 * 16 + lrow;
            const T* gp = Wb + (long long)n * a.K + tap * 64 + half * 32 + gchunk * 8;
            const void* src = (n < a.N && tap < taps) ? static_cast<const void*>(gp) : static_cast<const void*>(g_zero_row);
            __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(Bs + ((buf * TPS + ts) * 2 + half) * BIMG + g16 * 16 * 32), 16, 0, 0);
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
    for (int st = 0; st < nst; ++st) {
        __syncthreads();                                 // stage st (and, the first time, the input rows) has landed; buffer (st + 1) & 1 is free
        if (st + 1 < nst) issue_w(st + 1, (st + 1) & 1);
#pragma unroll
        for (int ts = 0; ts < TPS; ++ts) {
            const int tap = st * TPS + ts;
            if (tap >= taps) break;                      // (uniform)
            const int j0 = wm0 + fr + tap * a.conv_dil;  // this lane's row of the first row tile at this tap (the second is 16 rows further: same swizzle)
            const int asw = (fg ^ ((-((j0 & 15) >> 2)) & 3)) * 8;
            const T* const bt = Bs + (((st & 1) * TPS + ts) * 2) * BIMG;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x8 af[MT], bf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = load8(As + h * AIMG + (j0 + i * 16) * 32 + asw);
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[j] = load8(bt + h * BIMG + (j * 16 + fr) * 32 + bsw);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) mma32(acc[i][j], af[i], bf[j]);
            }
        }
    }
    __syncthreads();                                     // the epilogue stages through the weight buffers
    gemm_epilogue<T, MT, NT, WN, EPI_GENERIC, 0>(a, acc, reinterpret_cast<float*>(Bs) + wave * ROWS_PASS * SLD, lane, m0 + wm0, 0, bz, g);
}

}  // namespace

// 1 = launched, 0 = not eligible (the caller falls through to the tiled forms), -1 = error
int launch_conv_resident(const GemmArgs& a, hipStream_t s) {
    constexpr int MAXR = 160;
    if (a.dtype != DT_BF16 || a.epi != EPI_GENERIC || a.cin_pad != 64 || a.N > 64 || a.conv_stride != 1 || a.up != 1 || a.K < 128 || a.M < 2048 ||
        (a.lda & 7) || (a.a_gs & 7) || (a.a_bs & 7) || 128 + (a.K / 64 - 1) * a.conv_dil > MAXR || a.a_planes || a.w_planes)
        return 0;
    const long long gz = (long long)a.batch * a.groups;
    if (gz > 65535) return 0;
    const int slot = prof_begin(PK_GEMM, 2.0 * a.M * a.N * (double)a.K * a.batch * a.groups, s);
    hipLaunchKernelGGL((gconv64_kernel<MAXR, 2>), dim3((a.M + 127) / 128, 1, (unsigned)gz), dim3(256), 0, s, a);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 1 : (set_error("conv (resident-row form) launch failed"), -1);
}

}  // namespace hvx
