// aggressors.hip — micro-kernels for tools/platform_probe.py: each isolates ONE ingredient of gemm_x3_kernel (the kernel whose presence on
// the device changes the results of kernels running beside it) so that the ingredient that matters can be named.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -o tools/bin/libagg.so tools/aggressors.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (1) the operand split: hi = bf16(x), lo = bf16(x - hi)  -> v_cvt_pk_bf16_f32 + VALU, no LDS, no MFMA
__global__ __launch_bounds__(256) void agg_split(float* buf, int iters) {
    float x = buf[blockIdx.x * 256 + threadIdx.x];
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        const __bf16 h = (__bf16)x;
        const __bf16 l = (__bf16)(x - (float)h);
        acc += (unsigned)__builtin_bit_cast(unsigned short, h) + ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
        x = x * 1.0001f + 0.37f;
    }
    buf[blockIdx.x * 256 + threadIdx.x] = x + (float)(acc & 1);
}

// (2) the LDS pattern: LDS_BYTES of static LDS, 16-byte writes at an 80-byte row pitch, barrier, 16-byte fragment reads, barrier
template <int LDS_BYTES>
__global__ __launch_bounds__(256) void agg_lds(float* buf, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    const int tid = threadIdx.x, chunk = tid & 3, lrow = tid >> 2, lane = tid & 63;
    const int rows = LDS_BYTES / 80;
    f32x4 v = {buf[blockIdx.x * 256 + tid], 1.0f, 2.0f, 3.0f};
    f32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        for (int r = lrow; r < rows; r += 64) *reinterpret_cast<f32x4*>(smem + r * 80 + chunk * 16) = v;
        __syncthreads();
        for (int r = (lane & 15); r < rows; r += 16) acc += *reinterpret_cast<const f32x4*>(smem + r * 80 + (lane >> 4) * 16);
        __syncthreads();
        v += acc * 1e-9f;
    }
    buf[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

// (3) the matrix-core pattern: three dependent bf16 MFMAs per accumulator, NACC accumulators (AGPRs), no LDS
template <int NACC>
__global__ __launch_bounds__(256) void agg_mfma(float* buf, int iters) {
    const float x = buf[blockIdx.x * 256 + threadIdx.x];
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(x + e); b[e] = (__bf16)(x - e); }
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc[j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    buf[blockIdx.x * 256 + threadIdx.x] = s;
}

// (4) register footprint only: a long dependent VALU chain over NREG live registers, no LDS, no MFMA, no conversions
template <int NREG>
__global__ __launch_bounds__(256) void agg_regs(float* buf, int iters) {
    float r[NREG];
    const float x = buf[blockIdx.x * 256 + threadIdx.x];
    for (int j = 0; j < NREG; ++j) r[j] = x + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NREG; ++j) r[j] = r[j] * 1.0001f + r[(j + 1) % NREG] * 1e-6f;
    }
    float s = 0;
    for (int j = 0; j < NREG; ++j) s += r[j];
    buf[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- victims: deterministic kernels, one instruction class each; out[] must be bit-identical from launch to launch ------------------------
// (a) scalar fp32 FMA chain   (b) packed fp32 FMA chain (v_pk_fma_f32)   (c) fp32 MFMA chain   (d) bf16 MFMA chain
// (e) LDS table broadcast reads feeding FMAs (the shape of hift_stft_kernel)   (f) v_exp_f32 / v_sin_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(128) void vic_kernel(int kind, const float* in, float* out, int iters) {
    const int gid = blockIdx.x * 128 + threadIdx.x;
    const float x = in[gid];
    __shared__ float tab[48];
    if (threadIdx.x < 48) tab[threadIdx.x] = 0.01f * (float)(threadIdx.x * 7 % 31) - 0.1f;
    __syncthreads();
    float r = 0.0f;
    if (kind == 0) {
        float a = x, b = x * 0.5f + 1.0f;
        for (int i = 0; i < iters; ++i) { a = __builtin_fmaf(a, 0.999f, b); b = __builtin_fmaf(b, 0.998f, a * 1e-3f); }
        r = a + b;
    } else if (kind == 1) {
        f32x2 a = {x, x + 1.0f}, b = {x * 0.5f, 2.0f - x}, c = {0.999f, 0.998f};
        for (int i = 0; i < iters; ++i) { a = __builtin_elementwise_fma(a, c, b); b = __builtin_elementwise_fma(b, c, a * 1e-3f); }
        r = a[0] + a[1] + b[0] + b[1];
    } else if (kind == 2) {
        f32x4 acc = {0, 0, 0, 0};
        float a = x, b = 1.0f - x;
        for (int i = 0; i < iters; ++i) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0); a = a * 0.999f + 1e-3f; }
        r = acc[0] + acc[1] + acc[2] + acc[3];
    } else if (kind == 3) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(x + 0.1f * e); b[e] = (__bf16)(1.0f - x * e); }
        f32x4 acc = {0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0) * 0.5f;
        r = acc[0] + acc[1] + acc[2] + acc[3];
    } else if (kind == 4) {
        float a = x, b = 0.0f;
        for (int i = 0; i < iters; ++i) {
            const volatile float* t = tab;
#pragma unroll
            for (int j = 0; j < 16; ++j) b = __builtin_fmaf(a + j, t[(j * 5 + i) % 48], b);
            a = a * 0.999f + 1e-3f;
        }
        r = b;
    } else {
        float a = x;
        for (int i = 0; i < iters; ++i) a = __builtin_amdgcn_exp2f(-a) + __builtin_amdgcn_sinf(a) * 0.25f + 0.5f;
        r = a;
    }
    out[gid] = r;
}
extern "C" int vic_launch(int kind, const float* in, float* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(vic_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, kind, in, out, iters);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- ONE kernel holding both roles: does the interference need two kernels / two queues, or only two waves on one SIMD? -------------------
// mode 0: 512-thread workgroups, waves 0-3 stream MFMAs, waves 4-7 (one per SIMD, beside an MFMA wave) compute a 16-point DFT per thread
//         with fp32 operations the compiler packs (the arithmetic of hift_stft_kernel);  mode 1: the roles alternate by WORKGROUP (256 threads).
__device__ __forceinline__ void dft_role(const float* x, float* spec, int f, int frames, const float* cs, const float* sn, const float* wnd) {
    if (f >= frames) return;
    float xs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) xs[j] = x[4 * f + j] * wnd[j];
    float* o = spec + (long long)f * 32;
    // bins in pairs through explicit two-wide fp32 vectors: v_pk_fma_f32 with a broadcast sample and a (table, table) pair, as in the product kernel
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        f32x2 re = {0.0f, 0.0f}, im = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const f32x2 xx = {xs[j], xs[j]};
            const f32x2 c = {cs[(j * k) & 15], cs[(j * (k + 1)) & 15]}, s2 = {sn[(j * k) & 15], sn[(j * (k + 1)) & 15]};
            re = __builtin_elementwise_fma(xx, c, re);
            im = __builtin_elementwise_fma(-xx, s2, im);
        }
        o[k] = re[0]; o[k + 1] = re[1];
        o[9 + k] = im[0]; o[10 + k] = im[1];
    }
}
__device__ __forceinline__ float mfma_role(float x, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(x + e); b[e] = (__bf16)(x - e); }
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc[j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    return s;
}
__global__ __launch_bounds__(512) void mix_kernel(int mode, const float* x, float* spec, float* sink, int frames, int iters) {
    __shared__ float cs[16], sn[16], wnd[16];
    if (threadIdx.x < 16) {
        const float ang = 6.28318530717958647692f * (float)threadIdx.x / 16.0f;
        cs[threadIdx.x] = cosf(ang);
        sn[threadIdx.x] = sinf(ang);
        wnd[threadIdx.x] = 0.5f - 0.5f * cosf(ang);
    }
    __syncthreads();
    if (mode == 0) {
        if (threadIdx.x < 256) sink[blockIdx.x * 256 + threadIdx.x] = mfma_role(x[threadIdx.x], iters);
        else dft_role(x, spec, blockIdx.x * 256 + (threadIdx.x - 256), frames, cs, sn, wnd);
    } else {
        if (blockIdx.x & 1) sink[(blockIdx.x >> 1) * 256 + threadIdx.x] = mfma_role(x[threadIdx.x], iters);
        else dft_role(x, spec, (blockIdx.x >> 1) * 256 + threadIdx.x, frames, cs, sn, wnd);
    }
}
extern "C" int mix_launch(int mode, const float* x, float* spec, float* sink, int frames, int iters, void* stream) {
    const int per = 256, blocks = (frames + per - 1) / per;
    if (mode == 0) hipLaunchKernelGGL(mix_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, mode, x, spec, sink, frames, iters);
    else hipLaunchKernelGGL(mix_kernel, dim3(2 * blocks), dim3(256), 0, (hipStream_t)stream, mode, x, spec, sink, frames, iters);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int agg_launch(int kind, float* buf, int blocks, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    dim3 g(blocks), b(256);
    switch (kind) {
        case 1: hipLaunchKernelGGL(agg_split, g, b, 0, s, buf, iters); break;
        case 2: hipLaunchKernelGGL(agg_lds<30720>, g, b, 0, s, buf, iters); break;
        case 3: hipLaunchKernelGGL(agg_lds<27648>, g, b, 0, s, buf, iters); break;
        case 4: hipLaunchKernelGGL(agg_mfma<8>, g, b, 0, s, buf, iters); break;
        case 5: hipLaunchKernelGGL(agg_regs<160>, g, b, 0, s, buf, iters); break;
        case 6: hipLaunchKernelGGL(agg_lds<40960>, g, b, 0, s, buf, iters); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
