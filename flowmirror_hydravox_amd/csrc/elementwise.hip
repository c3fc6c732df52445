// elementwise.hip — small fused HBM-bound kernels of the hot path (norms, gathers, CFG/Euler update).
// All of them are one pass over their data with coalesced row-contiguous accesses; statistics are
// reduced with wave64 shuffles (+ one LDS hop across the 4 waves of a workgroup).
#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, red[w]);
    return t;
}

// ---- split-K reduce + residual + RMSNorm (Qwen2 RMSNorm: fp32 statistics, eps inside rsqrt) ----------
template <class T>
__global__ __launch_bounds__(256) void reduce_rmsnorm_kernel(ReduceNormArgs a) {
    __shared__ float red[4];
    const int m = blockIdx.x, H = a.H;
    const int z = m / a.rows_per_z, mz = m - z * a.rows_per_z;
    const int mf = a.y_frag_zrows > 0 ? z * a.y_frag_zrows + mz : m;          // row of the fragment-order output
    float* xr = a.x + (long long)m * a.ldx;
    const float* part = a.part ? a.part + (long long)z * a.part_zs + (long long)mz * H : nullptr;
    const float* bias = a.bias ? a.bias + (long long)z * a.bias_zs : nullptr;
    const float* gain = a.gain ? a.gain + (long long)z * a.gain_zs : nullptr;
    float ss = 0.0f;
    bool y_done = false;
    if (part && (H & 3) == 0 && (a.part_stride & 3) == 0 && (a.ldx & 3) == 0) {
        // plain copy (no statistics needed): y leaves from the registers that hold the new x — no barrier, no re-read of x through memory
        const bool y_inline = a.y && !a.do_norm && (a.ldy & 3) == 0;
        const int KTy = H >> 5;
        T* const yr = reinterpret_cast<T*>(a.y) + (a.y_frag ? 0 : (long long)m * a.ldy);
        y_done = y_inline;
        // 16-byte columns, all split partials of a column in flight at once (the loads are independent; only the adds are ordered)
        for (int c = threadIdx.x * 4; c < H; c += 1024) {
            f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
            f32x4 acc = {0, 0, 0, 0};
            int s = 0;
            for (; s + 4 <= a.split_k; s += 4) {
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(part + (long long)(s + 0) * a.part_stride + c);
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(part + (long long)(s + 1) * a.part_stride + c);
                const f32x4 p2 = *reinterpret_cast<const f32x4*>(part + (long long)(s + 2) * a.part_stride + c);
                const f32x4 p3 = *reinterpret_cast<const f32x4*>(part + (long long)(s + 3) * a.part_stride + c);
                acc = (((acc + p0) + p1) + p2) + p3;                                              // fixed order: deterministic
            }
            for (; s < a.split_k; ++s) acc += *reinterpret_cast<const f32x4*>(part + (long long)s * a.part_stride + c);
            v += acc;
            if (bias) v += *reinterpret_cast<const f32x4*>(bias + c);
            *reinterpret_cast<f32x4*>(xr + c) = v;
            if (y_inline) {
                // (fragment order keeps 4 consecutive columns of a row contiguous: c % 4 == 0)
                if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x4*>(yr + (a.y_frag ? frag_index(mf, c, KTy) : (long long)c)) = bf16x4{f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])};
                else *reinterpret_cast<f32x4*>(yr + c) = v;
            }
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        if (y_done) return;
    } else {
        for (int c = threadIdx.x; c < H; c += 256) {
            float v = xr[c];
            if (part) {
                float acc = 0.0f;
                for (int s = 0; s < a.split_k; ++s) acc += part[(long long)s * a.part_stride + c];   // fixed order: deterministic
                v += acc;
                if (bias) v += bias[c];
                xr[c] = v;
            }
            ss += v * v;
        }
        __syncthreads();
    }
    __syncthreads();             // the normalisation pass below re-reads columns written by other threads
    if (!a.y) return;
    T* y = reinterpret_cast<T*>(a.y) + (long long)m * a.ldy;
    if (!a.do_norm) {
        if (a.y_frag) {
            for (int c = threadIdx.x; c < H; c += 256) reinterpret_cast<T*>(a.y)[frag_index(mf, c, H >> 5)] = from_f32<T>(xr[c]);
            return;
        }
        for (int c = threadIdx.x; c < H; c += 256) y[c] = from_f32<T>(xr[c]);
        return;
    }
    ss = block_sum(ss, red);
    const float inv = rsqrtf(ss / (float)H + a.eps);
    for (int c = threadIdx.x; c < H; c += 256) {
        const float v = xr[c] * inv;
        const T o = from_f32<T>(gain ? gain[c] * to_f32(from_f32<T>(v)) : v);
        if (a.y_frag) reinterpret_cast<T*>(a.y)[frag_index(mf, c, H >> 5)] = o;
        else y[c] = o;
    }
}

int launch_reduce_rmsnorm(const ReduceNormArgs& a_in, hipStream_t s) {
    ReduceNormArgs a = a_in;
    if (a.M <= 0) return 0;
    if (a.rows_per_z <= 0) a.rows_per_z = a.M;
    if (a.y_frag && (a.dtype == DT_F32 || (a.H & 31) || (a.y_frag_zrows & 15))) return set_error("reduce_rmsnorm: fragment-order output is 16-bit with H %% 32 == 0"), -1;
    if (a.dtype == DT_BF16) hipLaunchKernelGGL(reduce_rmsnorm_kernel<bf16_t>, dim3(a.M), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(reduce_rmsnorm_kernel<float>, dim3(a.M), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("reduce_rmsnorm launch failed"), -1);
}

template <class T>
__global__ void act_rows_kernel(const float* x, int ldx, T* y, int ldy, int act, float param, const float* alpha, long long rows, int cols) {
    const long long total = rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cols;
        const int c = (int)(i - r * cols);
        y[r * ldy + c] = from_f32<T>(act_apply(act, x[r * ldx + c], param, alpha ? alpha[c] : 1.0f));
    }
}
// the same with the result as a (hi, lo) bf16 plane pair (GemmArgs.a_planes: the operand format of gemm_x3p_kernel)
__global__ void act_rows_planes_kernel(const float* x, int ldx, bf16_t* y, int ldy, long long plane, int act, float param, const float* alpha, long long rows, int cols4) {
    const long long total = rows * cols4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cols4;
        const int c = (int)(i - r * cols4) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
        const f32x4 al = alpha ? *reinterpret_cast<const f32x4*>(alpha + c) : f32x4{1, 1, 1, 1};
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = act_apply(act, v[e], param, al[e]);
            h[e] = f32_to_bf16(u);
            l[e] = f32_to_bf16(u - bf16_to_f32(h[e]));
        }
        *reinterpret_cast<bf16x4*>(y + r * ldy + c) = h;
        *reinterpret_cast<bf16x4*>(y + plane + r * ldy + c) = l;
    }
}
int launch_act_rows_planes(const float* x, int ldx, void* y, int ldy, long long plane, int act, float param, const float* alpha, long long rows, int cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return 0;
    if ((cols & 3) || (ldx & 3) || (ldy & 3) || (plane & 3)) return set_error("act_rows (plane pair): columns and strides must be multiples of 4"), -1;
    const long long want = (rows * (cols / 4) + 255) / 256;
    const int blocks = (int)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(act_rows_planes_kernel, dim3(blocks), dim3(256), 0, s, x, ldx, (bf16_t*)y, ldy, plane, act, param, alpha, rows, cols / 4);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("act_rows (plane pair) launch failed"), -1);
}
int launch_act_rows(const float* x, int ldx, void* y, int ldy, int dtype, int act, float param, const float* alpha, long long rows, int cols,
                    hipStream_t s) {
    if (rows <= 0 || cols <= 0) return 0;
    const long long want = (rows * cols + 255) / 256;
    const int blocks = (int)(want < 8192 ? want : 8192);
    if (dtype == DT_BF16) hipLaunchKernelGGL(act_rows_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, x, ldx, (bf16_t*)y, ldy, act, param, alpha, rows, cols);
    else hipLaunchKernelGGL(act_rows_kernel<float>, dim3(blocks), dim3(256), 0, s, x, ldx, (float*)y, ldy, act, param, alpha, rows, cols);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("act_rows launch failed"), -1);
}

// ---- row gathers ----------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* x, int ldx, const int* idx, float* y, int ldy, int H) {
    const int r = blockIdx.x;
    const int src = idx[r];
    for (int c = threadIdx.x; c < H; c += blockDim.x) y[(long long)r * ldy + c] = src >= 0 ? x[(long long)src * ldx + c] : 0.0f;
}
int launch_gather_rows_f32(const float* x, int ldx, const int* idx, float* y, int ldy, int rows, int H, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, s, x, ldx, idx, y, ldy, H);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("gather_rows launch failed"), -1);
}

// ---- head prologue: last rows -> final RMSNorm y (hidden_states[-1]) -> per MTP head: residual copy and input RMSNorm ---------------
// One launch instead of gather + norm + K copies + norm (llm_multi_head_v3.py:248-260, 886-887): y = gain_f * (x * rsqrt(mean x^2 + eps));
// hx[j] = y; ha[j] = T(gain_j * T(y * rsqrt(mean y^2 + eps_mtp))) — the mean square of y is shared by the K heads.
template <class T>
__global__ __launch_bounds__(256) void heads_prologue_kernel(const float* x, int ldx, const int* idx, const float* gain_f, float eps_f,
                                                             const float* gain_h, float eps_h, int K, int S, int H, float* ylast, float* hx, T* ha) {
    __shared__ float red[4];
    const int sq = blockIdx.x;
    const int src = idx[sq];
    const float* xr = x + (long long)(src >= 0 ? src : 0) * ldx;
    float ss = 0.0f;
    for (int c = threadIdx.x; c < H; c += 256) {
        const float v = src >= 0 ? xr[c] : 0.0f;
        ss += v * v;
    }
    ss = block_sum(ss, red);
    const float inv = rsqrtf(ss / (float)H + eps_f);
    float ss2 = 0.0f;
    for (int c = threadIdx.x; c < H; c += 256) {
        const float v = (src >= 0 ? xr[c] : 0.0f) * inv;
        const float y = gain_f[c] * v;
        ylast[(long long)sq * H + c] = y;
        ss2 += y * y;
    }
    __syncthreads();
    ss2 = block_sum(ss2, red);
    const float inv2 = rsqrtf(ss2 / (float)H + eps_h);
    for (int c = threadIdx.x; c < H; c += 256) {
        const float y = gain_f[c] * ((src >= 0 ? xr[c] : 0.0f) * inv);
        const float v = y * inv2;
        for (int j = 0; j < K; ++j) {
            hx[((long long)j * S + sq) * H + c] = y;
            ha[((long long)j * S + sq) * H + c] = from_f32<T>(gain_h[(long long)j * H + c] * to_f32(from_f32<T>(v)));
        }
    }
}
int launch_heads_prologue(const float* x, int ldx, const int* idx, const float* gain_f, float eps_f, const float* gain_h, float eps_h, int K,
                          int S, int H, float* ylast, float* hx, void* ha, int dtype, hipStream_t s) {
    if (S <= 0 || K <= 0) return 0;
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(heads_prologue_kernel<bf16_t>, dim3(S), dim3(256), 0, s, x, ldx, idx, gain_f, eps_f, gain_h, eps_h, K, S, H, ylast, hx, (bf16_t*)ha);
    else
        hipLaunchKernelGGL(heads_prologue_kernel<float>, dim3(S), dim3(256), 0, s, x, ldx, idx, gain_f, eps_f, gain_h, eps_h, K, S, H, ylast, hx, (float*)ha);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("heads_prologue launch failed"), -1);
}

template <class T>
__global__ void embed_kernel(const T* table, const int* tok, float* x, int ldx, int H) {
    const int r = blockIdx.x;
    const int t = tok[r];
    for (int c = threadIdx.x; c < H; c += blockDim.x) x[(long long)r * ldx + c] = t >= 0 ? to_f32(table[(long long)t * H + c]) : 0.0f;
}
int launch_embed(const void* table, int table_dtype, const int* tok, float* x, int ldx, int rows, int H, hipStream_t s) {
    if (rows <= 0) return 0;
    if (table_dtype == DT_BF16)
        hipLaunchKernelGGL(embed_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(table), tok, x, ldx, H);
    else
        hipLaunchKernelGGL(embed_kernel<float>, dim3(rows), dim3(256), 0, s, reinterpret_cast<const float*>(table), tok, x, ldx, H);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("embed launch failed"), -1);
}

template <class T>
__global__ void embed2_kernel(const T* speech, const T* text, const int* tok, float* x, int ldx, T* x_copy, int H, int copy_frag) {
    const int r = blockIdx.x;
    const int t = tok[r];
    const T* src = t >= 0 ? speech + (long long)t * H : (t <= -2 ? text + (long long)(-t - 2) * H : nullptr);
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        x[(long long)r * ldx + c] = src ? to_f32(src[c]) : 0.0f;
        if (x_copy) x_copy[copy_frag ? frag_index(r, c, H >> 5) : (long long)r * ldx + c] = src ? src[c] : from_f32<T>(0.0f);      // the table row itself
    }
}
int launch_embed2(const void* speech, const void* text, int dtype, const int* tok, float* x, int ldx, void* x_copy, int rows, int H, hipStream_t s,
                  int copy_frag) {
    if (rows <= 0) return 0;
    if (copy_frag && (dtype != DT_BF16 || (H & 31))) return set_error("embed2: fragment-order copy needs bf16 and H %% 32 == 0"), -1;
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(embed2_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, (const bf16_t*)speech, (const bf16_t*)text, tok, x, ldx, (bf16_t*)x_copy, H, copy_frag);
    else
        hipLaunchKernelGGL(embed2_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)speech, (const float*)text, tok, x, ldx, (float*)x_copy, H, 0);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("embed2 launch failed"), -1);
}

// ---- log_softmax over V (llm_multi_head_v3.py:888) ------------------------------------------------------
__global__ __launch_bounds__(256) void log_softmax_kernel(float* x, int ld, int V) {
    extern __shared__ __attribute__((aligned(16))) float row[];      // the row is read from global memory once
    __shared__ float red[4];
    float* xr = x + (long long)blockIdx.x * ld;
    float mx = -INFINITY;
    for (int base = 0; base < V; base += 256 * 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = base + u * 256 + threadIdx.x;
            t[u] = c < V ? xr[c] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = base + u * 256 + threadIdx.x;
            if (c < V) row[c] = t[u];
            mx = fmaxf(mx, t[u]);
        }
    }
    mx = block_max(mx, red);
    float sum = 0.0f;
    for (int c = threadIdx.x; c < V; c += 256) sum += expf(row[c] - mx);
    sum = block_sum(sum, red);
    const float lse = mx + logf(sum);
    for (int c = threadIdx.x; c < V; c += 256) xr[c] = row[c] - lse;
}
int launch_log_softmax(float* x, int ld, int rows, int V, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(log_softmax_kernel, dim3(rows), dim3(256), (size_t)V * 4, s, x, ld, V);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("log_softmax launch failed"), -1);
}

// ---- DiT adaLN: LayerNorm (no affine, biased variance, eps) then (1 + scale) * . + shift -----------------
// (XT: the residual stream's storage type, float or IEEE fp16 — GemmArgs.res_f16 / out_f16)
__device__ __forceinline__ f32x4 ln_load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ln_load4(const f16_t* p) {
    const f16x4 h = *reinterpret_cast<const f16x4*>(p);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
template <class T, class XT>
__global__ __launch_bounds__(256) void layernorm_mod_kernel(const XT* x, const float* shift, const float* scale, long long mod_bs, float eps,
                                                            T* y, int T_, int D) {
    // one wave per row, 4 rows per workgroup
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (row >= T_) return;
    const int lane = threadIdx.x & 63;
    const XT* xr = x + ((long long)b * T_ + row) * D;
    const float* sh = shift + (long long)b * mod_bs;
    const float* sc = scale + (long long)b * mod_bs;
    T* yr = y + ((long long)b * T_ + row) * D;
    if ((D & 255) == 0 && D <= 2048) {
        // the row is read from memory once: 4 consecutive channels per lane and step (16-byte loads), kept in registers
        f32x4 v[8];
        const int nq = D >> 8;
        float s1 = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < nq) {
                v[q] = ln_load4(xr + q * 256 + lane * 4);
                s1 += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
            }
        const float mean = wave_sum(s1) / (float)D;
        float s2 = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < nq) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[q][e] - mean;
                    s2 += d * d;
                }
            }
        const float inv = rsqrtf(wave_sum(s2) / (float)D + eps);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < nq) {
                const int c = q * 256 + lane * 4;
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc + c), h4 = *reinterpret_cast<const f32x4*>(sh + c);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[q][e] - mean) * inv * (1.0f + s4[e]) + h4[e];
                if constexpr (sizeof(T) == 2) {
                    *reinterpret_cast<typename Vec4<T>::type*>(yr + c) = typename Vec4<T>::type{from_f32<T>(o[0]), from_f32<T>(o[1]), from_f32<T>(o[2]), from_f32<T>(o[3])};
                } else {
                    *reinterpret_cast<f32x4*>(yr + c) = o;
                }
            }
        return;
    }
    float s1 = 0.0f;
    for (int c = lane; c < D; c += 64) s1 += (float)xr[c];
    const float mean = wave_sum(s1) / (float)D;
    float s2 = 0.0f;
    for (int c = lane; c < D; c += 64) {
        const float d = (float)xr[c] - mean;
        s2 += d * d;
    }
    const float inv = rsqrtf(wave_sum(s2) / (float)D + eps);
    for (int c = lane; c < D; c += 64) yr[c] = from_f32<T>(((float)xr[c] - mean) * inv * (1.0f + sc[c]) + sh[c]);
}
// D == 1024 (the DiT width): a wave owns TWO rows, 8 consecutive channels per lane and step, every load of both rows in flight before the
// first reduction (a row is 4 KB: one row per wave leaves the memory pipe idle during the two dependent wave reductions), 16-byte stores.
template <class T, class XT>
__global__ __launch_bounds__(256) void layernorm_mod1024_kernel(const XT* __restrict__ x, const float* __restrict__ shift,
                                                                const float* __restrict__ scale, long long mod_bs, float eps, T* __restrict__ y, int T_) {
    constexpr int D = 1024;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    const int b = blockIdx.y;
    if (row0 >= T_) return;
    const bool two = row0 + 1 < T_;
    const XT* xr = x + ((long long)b * T_ + row0) * D;
    f32x4 v[2][4];
    if constexpr (sizeof(XT) == 2) {
        f16x8 h[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q) h[r][q] = *reinterpret_cast<const f16x8*>(xr + (r && two ? D : 0) + q * 512 + lane * 8);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[r][2 * q][e] = (float)h[r][q][e]; v[r][2 * q + 1][e] = (float)h[r][q][4 + e]; }
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const XT* p = xr + (r && two ? D : 0) + q * 512 + lane * 8;
                v[r][2 * q] = *reinterpret_cast<const f32x4*>(p);
                v[r][2 * q + 1] = *reinterpret_cast<const f32x4*>(p + 4);
            }
    }
    const float* sh = shift + (long long)b * mod_bs;
    const float* sc = scale + (long long)b * mod_bs;
    f32x4 s4[4], h4[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        s4[2 * q] = *reinterpret_cast<const f32x4*>(sc + q * 512 + lane * 8);
        s4[2 * q + 1] = *reinterpret_cast<const f32x4*>(sc + q * 512 + lane * 8 + 4);
        h4[2 * q] = *reinterpret_cast<const f32x4*>(sh + q * 512 + lane * 8);
        h4[2 * q + 1] = *reinterpret_cast<const f32x4*>(sh + q * 512 + lane * 8 + 4);
    }
    float mean[2], inv[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float s1 = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s1 += (v[r][q][0] + v[r][q][1]) + (v[r][q][2] + v[r][q][3]);
        mean[r] = s1;
    }
    // the two rows' reductions interleave (independent shuffle chains)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mean[0] += __shfl_xor(mean[0], o, 64);
        mean[1] += __shfl_xor(mean[1], o, 64);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        mean[r] *= (1.0f / D);
        float s2 = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[r][q][e] - mean[r];
                s2 += d * d;
            }
        inv[r] = s2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        inv[0] += __shfl_xor(inv[0], o, 64);
        inv[1] += __shfl_xor(inv[1], o, 64);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r && !two) break;
        const float iv = rsqrtf(inv[r] * (1.0f / D) + eps);
        T* yr = y + ((long long)b * T_ + row0 + r) * D;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (v[r][2 * q][e] - mean[r]) * iv * (1.0f + s4[2 * q][e]) + h4[2 * q][e];
                o[4 + e] = (v[r][2 * q + 1][e] - mean[r]) * iv * (1.0f + s4[2 * q + 1][e]) + h4[2 * q + 1][e];
            }
            typename Vec8<T>::type w8;
#pragma unroll
            for (int e = 0; e < 8; ++e) w8[e] = from_f32<T>(o[e]);
            store8(yr + q * 512 + lane * 8, w8);
        }
    }
}
template <class T>
static void launch_ln_t(const void* x, int x_f16, const float* shift, const float* scale, long long mod_bs, float eps, void* y, int B, int T_, int D, bool fast, hipStream_t s) {
    const float* xf = reinterpret_cast<const float*>(x);
    const f16_t* xh = reinterpret_cast<const f16_t*>(x);
    T* yt = reinterpret_cast<T*>(y);
    if (fast) {
        dim3 grid((T_ + 7) / 8, B);
        if (x_f16) hipLaunchKernelGGL((layernorm_mod1024_kernel<T, f16_t>), grid, dim3(256), 0, s, xh, shift, scale, mod_bs, eps, yt, T_);
        else hipLaunchKernelGGL((layernorm_mod1024_kernel<T, float>), grid, dim3(256), 0, s, xf, shift, scale, mod_bs, eps, yt, T_);
    } else {
        dim3 grid((T_ + 3) / 4, B);
        if (x_f16) hipLaunchKernelGGL((layernorm_mod_kernel<T, f16_t>), grid, dim3(256), 0, s, xh, shift, scale, mod_bs, eps, yt, T_, D);
        else hipLaunchKernelGGL((layernorm_mod_kernel<T, float>), grid, dim3(256), 0, s, xf, shift, scale, mod_bs, eps, yt, T_, D);
    }
}
// dtype: the OUTPUT type (DT_F32 / DT_BF16 / DT_F16); x_f16: the residual stream is stored as fp16
int launch_layernorm_mod(const void* x, int x_f16, const float* shift, const float* scale, long long mod_bs, float eps, void* y, int dtype, int B, int T_,
                         int D, hipStream_t s) {
    if (B <= 0 || T_ <= 0) return 0;
    const bool fast = D == 1024 && ((mod_bs & 3) == 0) && ((((unsigned long long)shift | (unsigned long long)scale | (unsigned long long)x | (unsigned long long)y)) & 15) == 0;
    if (dtype == DT_BF16) launch_ln_t<bf16_t>(x, x_f16, shift, scale, mod_bs, eps, y, B, T_, D, fast, s);
    else if (dtype == DT_F16) launch_ln_t<f16_t>(x, x_f16, shift, scale, mod_bs, eps, y, B, T_, D, fast, s);
    else launch_ln_t<float>(x, x_f16, shift, scale, mod_bs, eps, y, B, T_, D, fast, s);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("layernorm_mod launch failed"), -1);
}

// ---- DiT input: cat[x, cond, mu, spks] as time-major rows (dit.py:91-97) ---------------------------------
template <class T>
__global__ void dit_concat_kernel(const float* x, const float* cond, const float* mu, const float* spk, T* y, int T_, int mel) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.y + threadIdx.y;
    if (t >= T_) return;
    const int W = 4 * mel;
    for (int c = threadIdx.x; c < W; c += blockDim.x) {
        const int which = c / mel, ch = c - which * mel;
        float v;
        if (which == 0) v = x[((long long)b * mel + ch) * T_ + t];
        else if (which == 1) v = cond[((long long)b * mel + ch) * T_ + t];
        else if (which == 2) v = mu[((long long)b * mel + ch) * T_ + t];
        else v = spk[(long long)b * mel + ch];
        y[((long long)b * T_ + t) * W + c] = from_f32<T>(v);
    }
}
int launch_dit_concat(const float* x, const float* cond, const float* mu, const float* spk, void* y, int dtype, int B, int T_, int mel,
                      hipStream_t s) {
    if (B <= 0 || T_ <= 0) return 0;
    dim3 block(64, 4), grid((T_ + 3) / 4, B);
    if (dtype == DT_BF16) hipLaunchKernelGGL(dit_concat_kernel<bf16_t>, grid, block, 0, s, x, cond, mu, spk, reinterpret_cast<bf16_t*>(y), T_, mel);
    else hipLaunchKernelGGL(dit_concat_kernel<float>, grid, block, 0, s, x, cond, mu, spk, reinterpret_cast<float*>(y), T_, mel);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("dit_concat launch failed"), -1);
}

// ---- sinusoidal timestep embedding (modules.py:71-83, scale 1000) ----------------------------------------
template <class T>
__global__ void time_sinus_kernel(const float* t, T* y, int dim) {
    const int b = blockIdx.x, half = dim / 2;
    const float step = logf(10000.0f) / (float)(half - 1);
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        const float e = expf((float)i * -step);
        const float v = 1000.0f * t[b] * e;
        y[(long long)b * dim + i] = from_f32<T>(sinf(v));
        y[(long long)b * dim + half + i] = from_f32<T>(cosf(v));
    }
}
int launch_time_sinus(const float* t, void* y, int dtype, int B, int dim, hipStream_t s) {
    if (B <= 0) return 0;
    if (dtype == DT_BF16) hipLaunchKernelGGL(time_sinus_kernel<bf16_t>, dim3(B), dim3(128), 0, s, t, reinterpret_cast<bf16_t*>(y), dim);
    else hipLaunchKernelGGL(time_sinus_kernel<float>, dim3(B), dim3(128), 0, s, t, reinterpret_cast<float*>(y), dim);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("time_sinus launch failed"), -1);
}

// ---- classifier-free guidance + Euler update (flow_matching.py:116-120) -----------------------------------
__global__ void cfg_euler_kernel(float* x, const float* v, int ldv, long long v_bs, float dt, float rate, int T_, int mel) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int ch = blockIdx.y;
    if (t >= T_) return;
    const float vc = v[(long long)t * ldv + ch], vu = v[v_bs + (long long)t * ldv + ch];
    const float d = (1.0f + rate) * vc - rate * vu;
    x[(long long)ch * T_ + t] = x[(long long)ch * T_ + t] + dt * d;
}
int launch_cfg_euler(float* x, const float* v, int ldv, long long v_bs, float dt, float rate, int T_, int mel, hipStream_t s) {
    if (T_ <= 0) return 0;
    hipLaunchKernelGGL(cfg_euler_kernel, dim3((T_ + 255) / 256, mel), dim3(256), 0, s, x, v, ldv, v_bs, dt, rate, T_, mel);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("cfg_euler launch failed"), -1);
}

// ---- casts / layout helpers --------------------------------------------------------------------------------
template <class S, class D>
__global__ void cast_kernel(const S* src, D* dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = from_f32<D>(to_f32(src[i]));
}
int launch_cast(const void* src, int sd, void* dst, int dd, long long n, hipStream_t s) {
    if (n <= 0) return 0;
    const long long want = (n + 255) / 256;
    const int blocks = (int)(want < 4096 ? want : 4096);
    if (sd == DT_F32 && dd == DT_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, n);
    else if (sd == DT_BF16 && dd == DT_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, n);
    else if (sd == DT_F32 && dd == DT_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, s, (const float*)src, (float*)dst, n);
    else hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("cast launch failed"), -1);
}

__global__ void transpose_kernel(const float* src, float* dst, int rows, int cols, int ld_src, int ld_dst) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? src[(long long)r * ld_src + c] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (c < cols && r < rows) dst[(long long)c * ld_dst + r] = tile[threadIdx.x][i];
    }
}
int launch_transpose_f32(const float* src, float* dst, int rows, int cols, int ld_src, int ld_dst, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, s, src, dst, rows, cols, ld_src, ld_dst);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("transpose launch failed"), -1);
}

// ---- F.interpolate(x, size=T_out, mode='linear') over the last axis of (rows, T_in): speed change of a mel (infer_speech_model.py:583-588,
// cli/model.py:424-426).  align_corners=False: src = (j + 0.5) * T_in / T_out - 0.5 clamped at 0, two-tap blend in fp32.
__global__ void resample_linear_kernel(const float* x, int t_in, float* y, int t_out, float scale) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (j >= t_out) return;
    float src = scale * ((float)j + 0.5f) - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    const int i0 = min((int)src, t_in - 1);
    const int i1 = min(i0 + 1, t_in - 1);
    const float l1 = src - (float)i0, l0 = 1.0f - l1;
    const float* xr = x + (long long)r * t_in;
    y[(long long)r * t_out + j] = l0 * xr[i0] + l1 * xr[i1];
}
int launch_resample_linear(const float* x, int rows, int t_in, float* y, int t_out, hipStream_t s) {
    if (rows <= 0 || t_out <= 0) return 0;
    if (t_in <= 0) return set_error("resample_linear: empty input"), -1;
    hipLaunchKernelGGL(resample_linear_kernel, dim3((t_out + 255) / 256, rows), dim3(256), 0, s, x, t_in, y, t_out, (float)t_in / (float)t_out);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("resample_linear launch failed"), -1);
}

template <class D>
__global__ void rows_to_dtype_kernel(const float* src, int ld_src, D* dst, int ld_dst, int cols, int cols_pad) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols_pad; c += blockDim.x)
        dst[(long long)r * ld_dst + c] = from_f32<D>(c < cols ? src[(long long)r * ld_src + c] : 0.0f);
}
int launch_rows_to_dtype(const float* src, int ld_src, void* dst, int dd, int ld_dst, int rows, int cols, int cols_pad, hipStream_t s) {
    if (rows <= 0) return 0;
    if (dd == DT_BF16) hipLaunchKernelGGL(rows_to_dtype_kernel<bf16_t>, dim3(rows), dim3(128), 0, s, src, ld_src, (bf16_t*)dst, ld_dst, cols, cols_pad);
    else hipLaunchKernelGGL(rows_to_dtype_kernel<float>, dim3(rows), dim3(128), 0, s, src, ld_src, (float*)dst, ld_dst, cols, cols_pad);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("rows_to_dtype launch failed"), -1);
}

// =====================================================================================================================
// Matcha-TTS / HiFi-GAN v1 family
// =====================================================================================================================

// ---- decoder input: pack [x, mu, spks, cond] (decoder.py:380-384; cv/flow/decoder.py:232-238) as time-major rows -------
__global__ void pack_rows_kernel(PackRowsArgs a) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.y + threadIdx.y;
    if (t >= a.T) return;
    float* y = a.dst + ((long long)b * a.T + t) * a.ld;
    for (int c = threadIdx.x; c < a.ld; c += blockDim.x) {
        float v = 0.0f;
        int base = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ck = a.channels[k];
            if (ck > 0 && c >= base && c < base + ck) {
                const int ch = c - base;
                v = a.broadcast[k] ? a.src[k][(long long)b * ck + ch] : a.src[k][((long long)b * ck + ch) * a.T + t];
            }
            base += ck;
        }
        y[c] = v;
    }
}
int launch_pack_rows(const PackRowsArgs& a, int B, hipStream_t s) {
    if (B <= 0 || a.T <= 0) return 0;
    dim3 block(64, 4), grid((a.T + 3) / 4, B);
    hipLaunchKernelGGL(pack_rows_kernel, grid, block, 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("pack_rows launch failed"), -1);
}

// ---- GroupNorm over (C / G channels x T rows) + Mish + row mask + per-(batch, channel) bias (decoder.py:40-75) ---------
// pass 1: partial (sum, sum of squares) per (batch, group, 64-row chunk); pass 2: every workgroup folds the partials of its
// batch in double (fixed order) and normalises its rows.  y = (act((x - mean) * rstd * gamma + beta) + tbias[b][c]) * [t < len]
constexpr int GN_CHUNK = 64;
__global__ __launch_bounds__(256) void groupnorm_partial_kernel(const float* x, int ld, int T, int C, int G, double* part) {
    const int chunk = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int Cg = C / G, n_chunks = gridDim.x;
    // the statistics run over ALL T rows of the conv output, padded rows included: the reference masks the conv's input and the
    // block's output, not what GroupNorm sees (decoder.py:52-54)
    const int r0 = chunk * GN_CHUNK, r1 = min(r0 + GN_CHUNK, T);
    float s1 = 0.0f, s2 = 0.0f;
    const int n = max(r1 - r0, 0) * Cg;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = r0 + i / Cg, c = g * Cg + i % Cg;
        const float v = x[((long long)b * T + r) * ld + c];
        s1 += v;
        s2 += v * v;
    }
    __shared__ float red[2][4];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s1;
        red[1][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* p = part + (((long long)b * G + g) * n_chunks + chunk) * 2;
        p[0] = (double)red[0][0] + red[0][1] + red[0][2] + red[0][3];
        p[1] = (double)red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const float* x, int ld, int T, int C, int G, const int* len, const double* part,
                                                              int n_chunks, const float* gamma, const float* beta, float eps,
                                                              const float* tbias, int act, float* y, int ldy) {
    const int b = blockIdx.y;
    const int Cg = C / G;
    const int tlen = len ? min(len[b], T) : T;
    __shared__ float s_mean[64], s_rstd[64];
    if (threadIdx.x < G) {
        double s1 = 0.0, s2 = 0.0;
        const double* p = part + ((long long)b * G + threadIdx.x) * n_chunks * 2;
        for (int i = 0; i < n_chunks; ++i) {
            s1 += p[2 * i];
            s2 += p[2 * i + 1];
        }
        const double cnt = (double)T * Cg;
        const double mean = s1 / cnt;
        const double var = s2 / cnt - mean * mean;
        s_mean[threadIdx.x] = (float)mean;
        s_rstd[threadIdx.x] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
    }
    __syncthreads();
    const int rows_per_block = 8;
    const int r0 = blockIdx.x * rows_per_block;
    for (int i = threadIdx.x; i < rows_per_block * C; i += 256) {
        const int r = r0 + i / C, c = i % C;
        if (r >= T) break;
        const int g = c / Cg;
        const float xv = x[((long long)b * T + r) * ld + c];
        float v = (xv - s_mean[g]) * s_rstd[g] * gamma[c] + beta[c];
        v = act_apply(act, v, 0.0f, 1.0f);
        if (tbias) v += tbias[(long long)b * C + c];       // h += mlp(t): the next block masks its input again, so masked rows stay 0
        if (r >= tlen) v = 0.0f;
        y[((long long)b * T + r) * ldy + c] = v;
    }
}
size_t groupnorm_ws_bytes(int B, int T, int G) { return (size_t)B * G * ((T + GN_CHUNK - 1) / GN_CHUNK) * 2 * sizeof(double); }
int launch_groupnorm_act(const float* x, int ld, int B, int T, int C, int G, const int* len, const float* gamma, const float* beta, float eps,
                         const float* tbias, int act, float* y, int ldy, void* ws, hipStream_t s) {
    if (B <= 0 || T <= 0) return 0;
    if (G < 1 || G > 64 || C % G) return set_error("groupnorm: C=%d groups=%d", C, G), -1;
    const int n_chunks = (T + GN_CHUNK - 1) / GN_CHUNK;
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(groupnorm_partial_kernel, dim3(n_chunks, G, B), dim3(256), 0, s, x, ld, T, C, G, part);
    hipLaunchKernelGGL(groupnorm_apply_kernel, dim3((T + 7) / 8, B), dim3(256), 0, s, x, ld, T, C, G, len, part, n_chunks, gamma, beta, eps, tbias,
                       act, y, ldy);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("groupnorm launch failed"), -1);
}

// ---- x * mask: zero the rows at and beyond len[b] (decoder.py:405, 441-443) ---------------------------------------------------
__global__ void mask_rows_kernel(float* x, int ld, int T, int C, const int* len) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * blockDim.y + threadIdx.y;
    if (r >= T || r < len[b]) return;
    for (int c = threadIdx.x; c < C; c += blockDim.x) x[((long long)b * T + r) * ld + c] = 0.0f;
}
int launch_mask_rows(float* x, int ld, int B, int T, int C, const int* len, hipStream_t s) {
    if (!len || B <= 0 || T <= 0) return 0;
    hipLaunchKernelGGL(mask_rows_kernel, dim3((T + 3) / 4, B), dim3(64, 4), 0, s, x, ld, T, C, len);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("mask_rows launch failed"), -1);
}

// ---- x += dt * v on (mel, T) with v given as time-major rows [T][ldv] (flow_matching.py:78-81) ----------------------------
__global__ void euler_rows_kernel(float* x, const float* v, int ldv, float dt, int T, int mel, long long x_bs, long long v_bs) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * mel) return;
    const int c = i / T, t = i - c * T;
    x[b * x_bs + i] += dt * v[b * v_bs + (long long)t * ldv + c];
}
int launch_euler_rows(float* x, const float* v, int ldv, float dt, int B, int T, int mel, hipStream_t s) {
    if (B <= 0 || T <= 0) return 0;
    hipLaunchKernelGGL(euler_rows_kernel, dim3((T * mel + 255) / 256, B), dim3(256), 0, s, x, v, ldv, dt, T, mel, (long long)mel * T, (long long)T * ldv);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("euler_rows launch failed"), -1);
}

// ---- Denoiser (denoiser.py:57-64): reflect padding of torch.stft(center=True), spectral subtraction, overlap-add ---------------
__global__ void reflect_pad_kernel(const float* x, float* y, int L, int pad, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int j = i - pad;
    if (j < 0) j = -j;
    if (j >= L) j = 2 * (L - 1) - j;
    y[i] = (j >= 0 && j < L) ? x[j] : 0.0f;
}
int launch_reflect_pad(const float* x, float* y, int L, int pad, int total, hipStream_t s) {
    hipLaunchKernelGGL(reflect_pad_kernel, dim3((total + 255) / 256), dim3(256), 0, s, x, y, L, pad, total);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("reflect_pad launch failed"), -1);
}
// spec rows: [re 0..bins) | [im 0..bins).  |S| - strength * bias, clamped at 0, phase kept: S *= max(|S| - s*b, 0) / |S|
__global__ void spectral_subtract_kernel(float* spec, int ld, int frames, int bins, const float* bias, float strength) {
    const int f = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= bins) return;
    float* row = spec + (long long)f * ld;
    const float re = row[k], im = row[bins + k];
    const float mag = sqrtf(re * re + im * im);
    const float m2 = fmaxf(mag - bias[k] * strength, 0.0f);
    // atan2(0, 0) == 0 in the reference: a zero bin comes back as (m2, 0)
    const float fr = mag > 0.0f ? re * (m2 / mag) : m2, fi = mag > 0.0f ? im * (m2 / mag) : 0.0f;
    row[k] = fr;
    row[bins + k] = fi;
}
int launch_spectral_subtract(float* spec, int ld, int frames, int bins, const float* bias, float strength, hipStream_t s) {
    if (frames <= 0) return 0;
    hipLaunchKernelGGL(spectral_subtract_kernel, dim3((bins + 255) / 256, frames), dim3(256), 0, s, spec, ld, frames, bins, bias, strength);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("spectral_subtract launch failed"), -1);
}
// mag[f][k] = sqrt(re^2 + im^2 + eps) for k < bins, 0 for the padding columns up to ld_mag
__global__ void spectral_magnitude_kernel(const float* spec, int ld, int bins, float* mag, int ld_mag, float eps) {
    const int f = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ld_mag) return;
    float v = 0.0f;
    if (k < bins) {
        const float re = spec[(long long)f * ld + k], im = spec[(long long)f * ld + bins + k];
        v = eps < 0.0f ? re * re + im * im : sqrtf(re * re + im * im + eps);     // eps < 0: the power spectrum
    }
    mag[(long long)f * ld_mag + k] = v;
}
// ---- feature post-processing of the speech frontends (cosyvoice/cli/frontend.py:92-115) --------------------------------------------
// whisper.log_mel_spectrogram: x = (max(x, max_all(x) - 8) + 4) / 4 over a [rows][cols] block with row stride ld
__global__ __launch_bounds__(1024) void whisper_range_kernel(float* x, int ld, int rows, int cols) {
    __shared__ float red[16];
    float m = -INFINITY;
    const long long n = (long long)rows * cols;
    for (long long i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, x[(i / cols) * ld + (i % cols)]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = red[0];
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    for (long long i = threadIdx.x; i < n; i += 1024) {
        float* p = x + (i / cols) * ld + (i % cols);
        *p = (fmaxf(*p, m - 8.0f) + 4.0f) * 0.25f;
    }
}
// cepstral mean normalisation of the CAM++ features: x[r][c] -= mean_r x[r][c]   (frontend.py:108), one workgroup per column
__global__ __launch_bounds__(256) void column_mean_sub_kernel(float* x, int ld, int rows) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) s += (double)x[(long long)r * ld + c];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (float)((red[0] + red[1] + red[2] + red[3]) / (double)rows);
    for (int r = threadIdx.x; r < rows; r += 256) x[(long long)r * ld + c] -= mean;
}
int launch_feature_post(float* x, int ld, int rows, int cols, int post, hipStream_t s) {
    if (rows <= 0 || cols <= 0 || post == 0) return 0;
    if (post == 1) hipLaunchKernelGGL(whisper_range_kernel, dim3(1), dim3(1024), 0, s, x, ld, rows, cols);
    else hipLaunchKernelGGL(column_mean_sub_kernel, dim3(cols), dim3(256), 0, s, x, ld, rows);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("feature post-processing launch failed"), -1);
}
int launch_spectral_magnitude(const float* spec, int ld, int frames, int bins, float* mag, int ld_mag, float eps, hipStream_t s) {
    if (frames <= 0) return 0;
    hipLaunchKernelGGL(spectral_magnitude_kernel, dim3((ld_mag + 255) / 256, frames), dim3(256), 0, s, spec, ld, bins, mag, ld_mag, eps);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("spectral_magnitude launch failed"), -1);
}
// torch.istft(center=True): y[n] = sum_f frame_f[n + n_fft/2 - f*hop] / sum_f w^2[...], n in [0, hop*(frames-1))
__global__ void overlap_add_kernel(const float* fr, int ld, int frames, int n_fft, int hop, const float* wsq, float* y, int out_len) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= out_len) return;
    const int p = n + n_fft / 2;
    const int f1 = min(p / hop, frames - 1), f0 = max((p - n_fft) / hop + ((p - n_fft) >= 0 ? 1 : 0), 0);
    float acc = 0.0f, env = 0.0f;
    for (int f = f0; f <= f1; ++f) {
        const int j = p - f * hop;
        if (j < 0 || j >= n_fft) continue;
        acc += fr[(long long)f * ld + j];
        env += wsq[j];
    }
    y[n] = env > 1e-11f ? acc / env : acc;
}
int launch_overlap_add(const float* frames_buf, int ld, int frames, int n_fft, int hop, const float* wsq, float* y, int out_len, hipStream_t s) {
    if (out_len <= 0) return 0;
    hipLaunchKernelGGL(overlap_add_kernel, dim3((out_len + 255) / 256), dim3(256), 0, s, frames_buf, ld, frames, n_fft, hop, wsq, y, out_len);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("overlap_add launch failed"), -1);
}

}  // namespace hvx
