// hvx_device.h — device-side building blocks shared by every HIP kernel of libhvx (gfx950 only).
//
// One abstraction carries both arithmetic types of the hot path:
//   T = bf16  -> v_mfma_f32_16x16x32_bf16        (production LLM / DiT)
//   T = float -> 8 x v_mfma_f32_16x16x4_f32      (HiFT is fp32 in the reference; fp32 parity mode)
// A "k-step" is 32 consecutive k values.  Within a k-step lane l of a wave64 owns row/col (l & 15)
// of the 16-wide operand and the 8 consecutive k values starting at (l >> 4) * 8 ("slot (g, j)",
// g = l >> 4, j = 0..7).  For bf16 that is exactly the hardware A/B fragment of 16x16x32; for f32
// the j-th 16x16x4 instruction consumes slot (g, j) of both operands, so the same loads feed it.
// C/D layout (both): col = l & 15, rows = (l >> 4) * 4 + r, r = 0..3   (MI355X guide §3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hvx {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// fp32 -> IEEE fp16, round-to-nearest-even, SATURATING at +-65504 (one v_med3_f32): a residual stream that outgrows fp16 clips instead of turning
// the rest of the network into NaNs (the reference's own fp16 run would overflow to inf at the same place)
__device__ __forceinline__ f16_t f32_to_f16_sat(float v) { return (f16_t)__builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f); }
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f32_to_bf16(float v) { return (bf16_t)v; }   // round-to-nearest-even

template <class T> struct Vec8;
template <> struct Vec8<bf16_t> { typedef bf16x8 type; };
template <> struct Vec8<float> { typedef f32x8 type; };
template <> struct Vec8<f16_t> { typedef f16x8 type; };
template <class T> struct Vec4;
template <> struct Vec4<bf16_t> { typedef bf16x4 type; };
template <> struct Vec4<f16_t> { typedef f16x4 type; };
template <> struct Vec4<float> { typedef f32x4 type; };

template <class T> __device__ __forceinline__ typename Vec8<T>::type zero8();
template <> __device__ __forceinline__ bf16x8 zero8<bf16_t>() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.0f;
    return z;
}
template <> __device__ __forceinline__ f16x8 zero8<f16_t>() {
    f16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (f16_t)0.0f;
    return z;
}
template <> __device__ __forceinline__ f32x8 zero8<float>() {
    f32x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return z;
}

// 8 consecutive elements from global / LDS (address must be 16-byte aligned)
__device__ __forceinline__ bf16x8 load8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ f16x8 load8(const f16_t* p) { return *reinterpret_cast<const f16x8*>(p); }
__device__ __forceinline__ f32x8 load8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    f32x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return r;
}
// the same for data read exactly once by one CU (streamed weights): non-temporal, does not displace the L2-resident activations
__device__ __forceinline__ bf16x8 load8_nt(const bf16_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p)); }
__device__ __forceinline__ f32x8 load8_nt(const float* p) {
    const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
    f32x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return r;
}
__device__ __forceinline__ void store8(bf16_t* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }
__device__ __forceinline__ void store8(f16_t* p, f16x8 v) { *reinterpret_cast<f16x8*>(p) = v; }
__device__ __forceinline__ void store8(float* p, f32x8 v) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}

// acc(16x16) += A(16x32) * B(32x16) over one k-step, operands in the slot layout described above
__device__ __forceinline__ void mma32(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
// IEEE fp16 operands (the DiT's Linears when the handle runs them in the reference's own deployed dtype): same fragment layout, same rate
__device__ __forceinline__ void mma32(f32x4& acc, const f16x8& a, const f16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x4& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
}

// ---- "fragment order" of a 16-bit activation matrix [rows][K] (K % 32 == 0, rows padded to 16) ---------------------------------------
// The matrix is stored as the MFMA operand fragments its consumer will load: [row tile of 16][k-step of 32][64 lanes][8], lane
// (row & 15) + 16 * ((k & 31) >> 3), element k & 7.  A wave then fetches a whole 16 x 32 fragment with ONE contiguous 1 KiB load
// instead of 16 row pieces of 64 B (tools/dec_lab.hip: the decode GEMMs run 1.2-1.5x faster on it).  KT = K / 32.
__device__ __forceinline__ long long frag_index(int row, int k, int KT) {
    return ((long long)(row >> 4) * KT + (k >> 5)) * 512 + (((row & 15) + ((k & 24) << 1)) << 3) + (k & 7);
}

// The LLM's KV cache in the same spirit (per (slot, kv head); element indices, any element type): K [ctx][64] is a fragment-order matrix with two
// k-steps per row (frag_index(pos, d, 2): a 16-key tile = the two A fragments of the score MFMAs, 1 KiB each at bf16); V is stored as the B fragments
// of the PV MFMAs: per 32-key block four d-tiles of [64 lanes][8], lane (d & 15) + 16 g holding keys {4g..4g+3, 16+4g..16+4g+3} of the block.
__device__ __forceinline__ long long vfrag_index(int pos, int d) {
    const int kk = pos & 31;
    return (long long)(pos >> 5) * 2048 + ((d >> 4) << 9) + (((d & 15) + ((kk & 12) << 2)) << 3) + ((kk >> 4) << 2) + (kk & 3);
}

template <class T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return f32_to_f16_sat(v); }
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }

// ---- activations (fp32 math) ---------------------------------------------------------------------
enum Act : int {
    ACT_NONE = 0,
    ACT_GELU_TANH = 1,   // nn.GELU(approximate="tanh")            (DiT/modules.py:277)
    ACT_SILU = 2,
    ACT_MISH = 3,        // x * tanh(softplus(x))                  (DiT/modules.py:122-127)
    ACT_ELU = 4,         // alpha = 1                               (f0_predictor.py:64-83)
    ACT_LRELU = 5,       // slope = param                           (generator.py:683,702)
    ACT_SNAKE = 6,       // x + sin^2(a x) / (a + 1e-9), a per col  (activation.py:73-84)
    ACT_TANH = 7,
    ACT_ABS = 8,         // |x|                                     (f0_predictor.py:103)
    ACT_LOG_CLAMP = 10,  // log(max(x, param))                      (matcha/utils/audio.py:23-24)
    ACT_SNAKEBETA = 9,   // x + sin^2(a x) / (b + 1e-9); a = alpha[col], b = alpha[n_cols + col] (matcha transformer.py:17-80, exp() applied at load)
};

// sin^2(y) for the Snake activations.  sin^2 has period pi and no sign: y = k pi + r with k = rint(y / pi), the reduction in two fused steps against a
// two-part pi (exact for |k| < 2^13, far beyond any activation argument), then the odd Taylor polynomial of sin through r^11 on |r| <= pi / 2
// (truncation 6e-8): 1.5e-7 absolute against libm's sinf squared, with 13 full-rate instructions and no branch — libm's sinf (range-reduction ladder,
// ~45 instructions) was 9 % of the vocoder once its convolutions stopped waiting on memory (round 5: 17.0 ms per 5632-frame utterance, 15.5 without the sines).
// Beyond |y| ~ 2.5e4 (k near 2^13) the two-part reduction stops being exact and r would leave [-pi / 2, pi / 2], where the polynomial is unbounded: r is clamped
// to that interval (ONE v_med3_f32), so a stray huge activation gives a value in [0, 1] — a bounded sample, like the reference's torch.sin — instead of a huge or
// infinite one; the value itself is then arbitrary, as any fp32 sine of an argument whose ulp exceeds pi is.  (A libm sinf fall-back on that branch was tried first: inlined
// into every epilogue instantiation it multiplied the compile time of the GEMM files by six and cost registers in kernels that never take it.)
__device__ __forceinline__ float sin_sq(float y) {
    const float k = __builtin_rintf(y * 0.3183098861837907f);
    float r = __builtin_fmaf(-k, 3.140625f, y);                    // pi = 3.140625 (exact in 8 bits) + 9.67653589793e-4
    r = __builtin_fmaf(-k, 9.67653589793e-4f, r);
    r = __builtin_amdgcn_fmed3f(r, -1.5707964f, 1.5707964f);
    const float r2 = r * r;
    float p = -2.5052108385441720e-8f;                             // -1 / 11!
    p = __builtin_fmaf(p, r2, 2.7557319223985893e-6f);            //  1 / 9!
    p = __builtin_fmaf(p, r2, -1.9841269841269841e-4f);           // -1 / 7!
    p = __builtin_fmaf(p, r2, 8.3333333333333332e-3f);            //  1 / 5!
    p = __builtin_fmaf(p, r2, -1.6666666666666666e-1f);           // -1 / 3!
    const float s = __builtin_fmaf(r * r2, p, r);
    return s * s;
}

__device__ __forceinline__ float act_apply(int act, float x, float param, float alpha, float beta = 1.0f) {
    switch (act) {
        case ACT_GELU_TANH: {
            // 0.5 x (1 + tanh(u)) == x sigmoid(2u) == x / (1 + 2^(-2 u log2 e)),  u = k0 (x + k1 x^3): one v_exp_f32 and one v_rcp_f32
            // (1 ulp each) instead of libm's tanhf — the epilogue of the DiT's first feed-forward Linear ran 2.7x longer than its K-loop
            const float c = 2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f;
            const float e = __builtin_amdgcn_exp2f(-c * x * __builtin_fmaf(k1 * x, x, 1.0f));    // +inf for very negative x: x * rcp(inf) = -0
            return x * __builtin_amdgcn_rcpf(1.0f + e);
        }
        case ACT_SILU: return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
        case ACT_MISH: {
            // x tanh(softplus(x)) with tanh(ln(1 + e)) = ((1 + e)^2 - 1) / ((1 + e)^2 + 1) = n / (n + 2), n = e (e + 2), e = exp(x): one v_exp_f32 and
            // one v_rcp_f32 instead of libm's expf + log1pf + tanhf (which made the Mish epilogue of the DiT's position-embedding convolutions a
            // third of the launch).  No cancellation anywhere (n > 0); above torch's softplus threshold (x > 20) tanh(x) is 1 to 1e-17: the branch
            // also keeps e (e + 2) finite.
            if (x > 20.0f) return x;
            const float e = __builtin_amdgcn_exp2f(1.4426950408889634f * x);
            const float n = e * (e + 2.0f);
            return x * n * __builtin_amdgcn_rcpf(n + 2.0f);
        }
        case ACT_ELU: return x > 0.0f ? x : expm1f(x);
        case ACT_LRELU: return x > 0.0f ? x : x * param;
        case ACT_SNAKE: return x + (1.0f / (alpha + 1e-9f)) * sin_sq(x * alpha);
        case ACT_SNAKEBETA: return x + (1.0f / (beta + 1e-9f)) * sin_sq(x * alpha);
        case ACT_LOG_CLAMP: return logf(fmaxf(x, param));
        case ACT_TANH: return tanhf(x);
        case ACT_ABS: return fabsf(x);
        default: return x;
    }
}

// four values at once with the dispatch OUTSIDE the element loop: the epilogues call this per 4-column chunk, and a per-element
// `switch` on a runtime activation code (with libm bodies inlined into every arm) cost more than the K-loop of a K = 1024 Linear
__device__ __forceinline__ f32x4 act_apply4(int act, f32x4 x, float param, f32x4 alpha, f32x4 beta) {
    f32x4 y;
    switch (act) {
#define HVX_ACT4(A) case A: _Pragma("unroll") for (int e = 0; e < 4; ++e) y[e] = act_apply(A, x[e], param, alpha[e], beta[e]); return y;
        HVX_ACT4(ACT_GELU_TANH) HVX_ACT4(ACT_SILU) HVX_ACT4(ACT_MISH) HVX_ACT4(ACT_ELU) HVX_ACT4(ACT_LRELU) HVX_ACT4(ACT_SNAKE)
        HVX_ACT4(ACT_SNAKEBETA) HVX_ACT4(ACT_LOG_CLAMP) HVX_ACT4(ACT_TANH) HVX_ACT4(ACT_ABS)
#undef HVX_ACT4
        default: return x;
    }
}

// ---- wave / block reductions (wave = 64 lanes) ---------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace hvx
