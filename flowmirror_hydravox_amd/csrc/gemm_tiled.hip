// gemm_tiled.hip — LDS-tiled implicit-GEMM on MFMA for gfx950 (see hvx_kernels.h: GemmArgs).
//
// Workgroup = 256 threads = 4 waves.  Block tile BM x BN, K-step 32.  Both operands are
// K-contiguous, so one 16-byte (bf16) / 32-byte (f32) vector per lane is exactly an MFMA
// fragment row; tiles are staged global -> registers -> LDS with the next K-step's global
// loads issued before the current step's MFMAs (register prefetch).  LDS rows are padded
// (bf16: 80 B, f32: 144 B) so that the 16 rows read by one ds_read_b128 lane group fall on
// 16 distinct 16-byte slots.  The conv index map (tap, dilation, stride, nearest-upsample,
// zero padding) lives in the A-tile loader; every epilogue variant is fused.
#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

template <class T> struct LdsPad { static constexpr int value = (sizeof(T) == 2) ? 8 : 4; };

template <class T, int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(GemmArgs a) {
    constexpr int BK = 32;
    constexpr int LDK = BK + LdsPad<T>::value;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    constexpr int A_VECS = BM * 4 / 256, B_VECS = BN * 4 / 256;
    typedef typename Vec8<T>::type V8;

    __shared__ __attribute__((aligned(16))) T As[BM * LDK];
    __shared__ __attribute__((aligned(16))) T Bs[BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int bz = blockIdx.z / a.groups, g = blockIdx.z % a.groups;

    const T* __restrict__ Ab = reinterpret_cast<const T*>(a.A) + (long long)bz * a.a_bs + (long long)g * a.a_gs;
    const T* __restrict__ Wb = reinterpret_cast<const T*>(a.W) + (long long)g * a.w_gs;

    // ---- per-thread tile-load coordinates ------------------------------------------------------
    int a_row[A_VECS], a_m[A_VECS];
    int b_row[B_VECS];
    const int chunk = tid & 3;                      // which 8-element chunk of the 32-wide K-step
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) {
        a_row[i] = (tid >> 2) + i * 64;
        a_m[i] = m0 + a_row[i];
    }
#pragma unroll
    for (int i = 0; i < B_VECS; ++i) b_row[i] = (tid >> 2) + i * 64;

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
    auto stash = [&](const V8 (&ra)[A_VECS], const V8 (&rb)[B_VECS]) {
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) store8(&As[a_row[i] * LDK + chunk * 8], ra[i]);
#pragma unroll
        for (int i = 0; i < B_VECS; ++i) store8(&Bs[b_row[i] * LDK + chunk * 8], rb[i]);
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    V8 ra[A_VECS], rb[B_VECS];
    load_a(0, ra);
    load_b(0, rb);
    stash(ra, rb);
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    for (int kc = 0; kc < nk; ++kc) {
        const bool more = (kc + 1) < nk;
        if (more) {
            load_a(kc + 1, ra);
            load_b(kc + 1, rb);
        }
        V8 af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = load8(&As[(wm0 + i * 16 + fr) * LDK + fg * 8]);
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = load8(&Bs[(wn0 + j * 16 + fr) * LDK + fg * 8]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) mma32(acc[i][j], af[i], bf[j]);
        __syncthreads();
        if (more) {
            stash(ra, rb);
            __syncthreads();
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------------
    if constexpr (EPI == EPI_GENERIC) {
        const long long ob = (long long)bz * a.out_bs;
        const int total_cols = a.groups * a.N;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn0 + j * 16 + fr;
            const int gc = g * a.N + col;
            const bool col_ok = col < a.N;
            const bool col_pad = (!col_ok) && (g == a.groups - 1) && (gc < a.out_cols);     // zero-fill padded channels
            const bool col_pad2 = (!col_ok) && (g == a.groups - 1) && (gc < a.out2_cols);
            if (!col_ok && !col_pad && !col_pad2) continue;
            const float bias = (col_ok && a.bias) ? a.bias[gc] : 0.0f;
            const float alpha = (col_ok && a.act_alpha) ? a.act_alpha[gc] : 1.0f;
            const float alpha2 = (col_ok && a.act2_alpha) ? a.act2_alpha[gc] : 1.0f;
            const float gate = (col_ok && a.gate) ? a.gate[(long long)bz * a.gate_bs + gc] : 1.0f;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm0 + i * 16 + fg * 4 + r;
                    if (row >= a.M || (row + a.out_row_off) < 0) continue;
                    float v = 0.0f;
                    if (col_ok) {
                        v = act_apply(a.act, acc[i][j][r] + bias, a.act_param, alpha) * gate;
                        if (a.res) v += a.res[(long long)bz * a.res_bs + (long long)(row + a.res_row_off) * a.ldres + gc];
                        if (a.res2) v += a.res2[(long long)bz * a.res2_bs + (long long)row * a.ldres2 + gc];
                        v *= a.scale;
                        if (a.div != 0.0f) v = v / a.div;
                    }
                    if (a.out && (col_ok || col_pad)) {
                        const long long o = ob + (long long)(row + a.out_row_off) * a.ldo + gc;
                        if (a.out_f32) reinterpret_cast<float*>(a.out)[o] = v;
                        else reinterpret_cast<T*>(a.out)[o] = from_f32<T>(v);
                    }
                    if (a.out2 && (col_ok || col_pad2)) {
                        const long long o2 = (long long)bz * a.out2_bs + (long long)(row + a.out2_row_off) * a.ldo2 + gc;
                        reinterpret_cast<T*>(a.out2)[o2] = from_f32<T>(col_ok ? act_apply(a.act2, v, a.act2_param, alpha2) : 0.0f);
                    }
                }
            }
        }
        (void)total_cols;
    } else {   // EPI_QKV_DIT
        const int D = a.heads * 64;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn0 + j * 16 + fr;
            const bool col_ok = col < a.N;
            const int which = col_ok ? col / D : 0;
            const int c = col_ok ? col - which * D : 0;
            const int h = c >> 6, d = c & 63;
            const float bias = (col_ok && a.bias) ? a.bias[col] : 0.0f;
            const bool rot = col_ok && which < 2 && c < 64;      // x_transformers partial rotary: first 64 channels of the row
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row0 = m0 + wm0 + i * 16 + fg * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[i][j][r] + bias;
                    const float partner = __shfl_xor(x, 1, 64);          // neighbouring channel, same rows
                    if (rot && (row0 + r) < a.M) {
                        const float cs = a.rope_cos[(long long)(row0 + r) * 32 + (d >> 1)];
                        const float sn = a.rope_sin[(long long)(row0 + r) * 32 + (d >> 1)];
                        x = (d & 1) ? (x * cs + partner * sn) : (x * cs - partner * sn);
                    }
                    v[r] = x;
                }
                if (!col_ok) continue;
                if (which < 2) {
                    T* dst = reinterpret_cast<T*>(which == 0 ? a.q : a.k) + (((long long)bz * a.heads + h) * a.t_pad) * 64 + d;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row0 + r < a.M) dst[(long long)(row0 + r) * 64] = from_f32<T>(v[r]);
                } else {
                    T* dst = reinterpret_cast<T*>(a.vT) + (((long long)bz * a.heads + h) * 64 + d) * a.t_pad + row0;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row0 + r < a.M) dst[r] = from_f32<T>(v[r]);
                }
            }
        }
    }
}

template <class T, int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmArgs& a, hipStream_t s) {
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.batch * a.groups);
    const int slot = prof_begin(sizeof(T) == 2 ? PK_GEMM : PK_GEMM_F32, 2.0 * a.M * a.N * (double)a.K * a.batch * a.groups, s);
    if (a.epi == EPI_GENERIC)
        hipLaunchKernelGGL((gemm_tiled_kernel<T, BM, BN, WM, WN, EPI_GENERIC>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_tiled_kernel<T, BM, BN, WM, WN, EPI_QKV_DIT>), grid, dim3(256), 0, s, a);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("gemm launch failed"), -1);
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
    return a.dtype == DT_BF16 ? launch_t<bf16_t>(a, s) : launch_t<float>(a, s);
}

}  // namespace hvx
