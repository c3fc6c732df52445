// onnx_ops.hip — the small fp32 operators an ONNX graph executor needs beside the GEMM / convolution kernels (include/hvx.h: hvx_nd_*, hvx_rows_*).
//
// the zero-shot frontend of the reference runs two ONNX graphs on the prompt audio through onnxruntime
// (server/model_utils/cosyvoice/cli/frontend.py:92-115: `speech_tokenizer_v3.onnx` on the whisper log-mel, `campplus.onnx` on the kaldi fbank).
// The graphs themselves are assets of the weights repository; what can be built without them is the executor: `onnx_graph.py` walks a graph and
// maps Conv / MatMul / Gemm onto the fp32 GEMM forms of this library (hvx_op_gemm) and everything else onto the kernels of this file:
//   * hvx_nd_elementwise — unary / binary / select operators over up-to-6-D tensors with per-operand element strides (0 = broadcast): one kernel
//     covers the arithmetic of a graph AND its data movement (Transpose, Slice, Expand, Concat pieces are strided copies);
//   * hvx_rows_reduce / hvx_rows_softmax — reductions over the last axis of a [rows][cols] view (the executor permutes first);
//   * hvx_avgpool_rows — AveragePool over the last axis with ONNX's ceil_mode / count_include_pad;
//   * hvx_conv2d — direct 2-D convolution (the CAM++ front module: a handful of 3x3 convolutions over [C][80][T]).
// All of them are HBM-bound one-pass kernels: coalesced along the innermost output axis, fp32 throughout (the reference runs these graphs in fp32).
#include <math.h>

#include "hvx.h"
#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

namespace {

struct NdDesc {
    int ndim;
    int shape[6];
    long long sa[6], sb[6], sc[6];
};

__device__ __forceinline__ float ew_unary(int op, float x, float p0, float p1) {
    switch (op) {
        case HVX_EW_COPY: return x;
        case HVX_EW_RELU: return x > 0.0f ? x : 0.0f;
        case HVX_EW_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case HVX_EW_TANH: return tanhf(x);
        case HVX_EW_ERF: return erff(x);
        case HVX_EW_SQRT: return sqrtf(x);
        case HVX_EW_EXP: return expf(x);
        case HVX_EW_LOG: return logf(x);
        case HVX_EW_NEG: return -x;
        case HVX_EW_ABS: return fabsf(x);
        case HVX_EW_ROUND: return rintf(x);                       // ONNX Round: half to even
        case HVX_EW_FLOOR: return floorf(x);
        case HVX_EW_CEIL: return ceilf(x);
        case HVX_EW_RECIP: return 1.0f / x;
        case HVX_EW_CLIP: return fminf(fmaxf(x, p0), p1);
        case HVX_EW_LEAKY_RELU: return x > 0.0f ? x : x * p0;
        case HVX_EW_SOFTPLUS: return x > 20.0f ? x : log1pf(expf(x));
        case HVX_EW_SIN: return sinf(x);
        case HVX_EW_COS: return cosf(x);
        default: return x;
    }
}
__device__ __forceinline__ float ew_binary(int op, float x, float y) {
    switch (op) {
        case HVX_EW_ADD: return x + y;
        case HVX_EW_SUB: return x - y;
        case HVX_EW_MUL: return x * y;
        case HVX_EW_DIV: return x / y;
        case HVX_EW_POW: return powf(x, y);
        case HVX_EW_MAX: return fmaxf(x, y);
        case HVX_EW_MIN: return fminf(x, y);
        case HVX_EW_EQUAL: return x == y ? 1.0f : 0.0f;
        case HVX_EW_LESS: return x < y ? 1.0f : 0.0f;
        case HVX_EW_GREATER: return x > y ? 1.0f : 0.0f;
        default: return x;
    }
}

__global__ __launch_bounds__(256) void nd_elementwise_kernel(int op, NdDesc d, const float* a, const float* b, const float* c, float p0, float p1, float* out,
                                                             long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long rem = i, ia = 0, ib = 0, ic = 0;
#pragma unroll
        for (int k = 5; k >= 0; --k) {
            if (k < d.ndim) {
                const int n = d.shape[k];
                const long long q = rem / n;
                const int r = (int)(rem - q * n);
                rem = q;
                ia += r * d.sa[k];
                ib += r * d.sb[k];
                ic += r * d.sc[k];
            }
        }
        float v;
        if (op >= HVX_EW_ADD && op < HVX_EW_WHERE) v = ew_binary(op, a[ia], b[ib]);
        else if (op == HVX_EW_WHERE) v = a[ia] != 0.0f ? b[ib] : c[ic];
        else v = ew_unary(op, a[ia], p0, p1);
        out[i] = v;
    }
}

// one wave per row (cols <= a few thousand in these graphs), grid-strided over rows
__global__ __launch_bounds__(256) void rows_reduce_kernel(int op, const float* x, long long rows, long long cols, float* out) {
    const int lane = threadIdx.x & 63;
    const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = w0; r < rows; r += nw) {
        const float* xr = x + r * cols;
        float acc = (op == HVX_RED_MAX) ? -INFINITY : (op == HVX_RED_MIN ? INFINITY : 0.0f);
        for (long long c = lane; c < cols; c += 64) {
            const float v = xr[c];
            if (op == HVX_RED_MAX) acc = fmaxf(acc, v);
            else if (op == HVX_RED_MIN) acc = fminf(acc, v);
            else if (op == HVX_RED_SUMSQ) acc += v * v;
            else acc += v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float t = __shfl_xor(acc, o, 64);
            acc = (op == HVX_RED_MAX) ? fmaxf(acc, t) : (op == HVX_RED_MIN ? fminf(acc, t) : acc + t);
        }
        if (lane == 0) out[r] = (op == HVX_RED_MEAN) ? acc / (float)cols : acc;
    }
}

__global__ __launch_bounds__(256) void rows_softmax_kernel(const float* x, long long rows, long long cols, float* out) {
    const int lane = threadIdx.x & 63;
    const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = w0; r < rows; r += nw) {
        const float* xr = x + r * cols;
        float* yr = out + r * cols;
        float m = -INFINITY;
        for (long long c = lane; c < cols; c += 64) m = fmaxf(m, xr[c]);
        m = wave_max(m);
        float s = 0.0f;
        for (long long c = lane; c < cols; c += 64) s += expf(xr[c] - m);
        s = wave_sum(s);
        const float inv = 1.0f / s;
        for (long long c = lane; c < cols; c += 64) yr[c] = expf(xr[c] - m) * inv;
    }
}

__global__ __launch_bounds__(256) void avgpool_rows_kernel(const float* x, long long rows, int t_in, int kernel, int stride, int pad, int count_include_pad, float* y,
                                                           int t_out) {
    const long long total = rows * t_out;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / t_out;
        const int o = (int)(i - r * t_out);
        const int t0 = o * stride - pad;
        float acc = 0.0f;
        int n = 0;
        for (int k = 0; k < kernel; ++k) {
            const int t = t0 + k;
            if (t >= 0 && t < t_in) {
                acc += x[r * t_in + t];
                ++n;
            } else if (count_include_pad && t < t_in + pad) {
                ++n;                                               // (padding counts; positions past the padded end — ceil_mode overhang — never do)
            }
        }
        y[i] = n > 0 ? acc / (float)n : 0.0f;
    }
}

__global__ __launch_bounds__(256) void conv2d_kernel(const float* x, const float* w, const float* bias, int B, int Cin, int H, int W, int Cout, int kh, int kw, int sh,
                                                     int sw, int ph, int pw, int Ho, int Wo, float* y) {
    const long long total = (long long)B * Cout * Ho * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long rem = i;
        const int wo = (int)(rem % Wo); rem /= Wo;
        const int ho = (int)(rem % Ho); rem /= Ho;
        const int co = (int)(rem % Cout);
        const int b = (int)(rem / Cout);
        float acc = bias ? bias[co] : 0.0f;
        for (int ci = 0; ci < Cin; ++ci)
            for (int i2 = 0; i2 < kh; ++i2) {
                const int hi = ho * sh - ph + i2;
                if (hi < 0 || hi >= H) continue;
                const float* xr = x + (((long long)b * Cin + ci) * H + hi) * W;
                const float* wr = w + (((long long)co * Cin + ci) * kh + i2) * kw;
                for (int j = 0; j < kw; ++j) {
                    const int wi = wo * sw - pw + j;
                    if (wi >= 0 && wi < W) acc = fmaf(xr[wi], wr[j], acc);
                }
            }
        y[i] = acc;
    }
}

int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace
}  // namespace hvx

using namespace hvx;

extern "C" {

int hvx_nd_elementwise(int32_t op, const hvx_nd* d, const float* a, const float* b, const float* c, float p0, float p1, float* out, hvx_stream s) {
    if (!d || !a || !out || d->ndim < 1 || d->ndim > 6) return set_error("hvx_nd_elementwise: bad descriptor"), -1;
    const bool binary = op >= HVX_EW_ADD && op < HVX_EW_WHERE;
    if ((binary || op == HVX_EW_WHERE) && !b) return set_error("hvx_nd_elementwise: operator %d needs a second operand", op), -1;
    if (op == HVX_EW_WHERE && !c) return set_error("hvx_nd_elementwise: Where needs three operands"), -1;
    if (op < 0 || op > HVX_EW_WHERE) return set_error("hvx_nd_elementwise: unknown operator %d", op), -1;
    NdDesc k;
    k.ndim = d->ndim;
    long long total = 1;
    for (int i = 0; i < 6; ++i) {
        k.shape[i] = i < d->ndim ? d->shape[i] : 1;
        k.sa[i] = i < d->ndim ? d->stride_a[i] : 0;
        k.sb[i] = i < d->ndim ? d->stride_b[i] : 0;
        k.sc[i] = i < d->ndim ? d->stride_c[i] : 0;
        if (k.shape[i] < 0) return set_error("hvx_nd_elementwise: negative extent"), -1;
        total *= k.shape[i];
    }
    if (total == 0) return 0;
    hipLaunchKernelGGL(nd_elementwise_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, op, k, a, b ? b : a, c ? c : a, p0, p1, out, total);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("hvx_nd_elementwise: launch failed"), -1);
}

int hvx_rows_reduce(int32_t op, const float* x, int64_t rows, int64_t cols, float* out, hvx_stream s) {
    if (!x || !out || rows < 0 || cols < 1 || op < HVX_RED_SUM || op > HVX_RED_SUMSQ) return set_error("hvx_rows_reduce: bad arguments"), -1;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(rows_reduce_kernel, dim3(grid_for(rows * 64)), dim3(256), 0, (hipStream_t)s, op, x, (long long)rows, (long long)cols, out);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("hvx_rows_reduce: launch failed"), -1);
}

int hvx_rows_softmax(const float* x, int64_t rows, int64_t cols, float* out, hvx_stream s) {
    if (!x || !out || rows < 0 || cols < 1) return set_error("hvx_rows_softmax: bad arguments"), -1;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(rows_softmax_kernel, dim3(grid_for(rows * 64)), dim3(256), 0, (hipStream_t)s, x, (long long)rows, (long long)cols, out);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("hvx_rows_softmax: launch failed"), -1);
}

int hvx_avgpool_rows(const float* x, int64_t rows, int32_t t_in, int32_t kernel, int32_t stride, int32_t pad, int32_t count_include_pad, float* y, int32_t t_out,
                     hvx_stream s) {
    if (!x || !y || rows < 0 || t_in < 1 || kernel < 1 || stride < 1 || pad < 0 || t_out < 1) return set_error("hvx_avgpool_rows: bad arguments"), -1;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(avgpool_rows_kernel, dim3(grid_for(rows * t_out)), dim3(256), 0, (hipStream_t)s, x, (long long)rows, t_in, kernel, stride, pad, count_include_pad, y,
                       t_out);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("hvx_avgpool_rows: launch failed"), -1);
}

int hvx_conv2d(const float* x, const float* w, const float* bias, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t kh, int32_t kw, int32_t sh, int32_t sw,
               int32_t ph, int32_t pw, float* y, hvx_stream s) {
    if (!x || !w || !y || B < 1 || Cin < 1 || Cout < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1 || ph < 0 || pw < 0) return set_error("hvx_conv2d: bad arguments"), -1;
    const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    if (Ho < 1 || Wo < 1) return set_error("hvx_conv2d: empty output"), -1;
    const long long total = (long long)B * Cout * Ho * Wo;
    hipLaunchKernelGGL(conv2d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, x, w, bias, B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo, y);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("hvx_conv2d: launch failed"), -1);
}

}  // extern "C"
