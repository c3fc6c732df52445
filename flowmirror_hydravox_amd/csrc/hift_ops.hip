// hift_ops.hip — NSF source, STFT and iSTFT kernels of the causal HiFT vocoder (fp32, like the reference).
// Cites server/model_utils/cosyvoice/hifigan/generator.py.
#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

// ---- frame-level harmonic phase (SineGen2._f02sine, causal/eval: :233-260) --------------------------------
// rad = (f0*(h+1)/sr) % 1 ; linear x(1/up) downsampling of the frame-constant signal returns the frame value ;
// cumsum (torch CPU accumulates fp32 cumsum in double) ; * 2 * pi ; * up (nearest upsampling keeps it frame-constant)
// One workgroup per harmonic; thread i owns a contiguous run of frames: run sums in double, an exclusive prefix over the 256 runs, then the
// run is walked again from its prefix.  (Round 2 walked all T frames in one thread per harmonic: 0.86 ms for a 5632-frame utterance.)  The
// double-precision partial sums are re-associated, which moves a value by <= 1e-16 relative before it is rounded to fp32 — the same result
// as the sequential sum except on an fp32 rounding boundary.
__global__ __launch_bounds__(256) void hift_phase_kernel(const float* f0, float* phase, int T, int H, float sr, float up) {
    __shared__ double part[256];
    const int h = blockIdx.x, i = threadIdx.x;
    const float mult = (float)(h + 1);
    const int per = (T + 255) / 256, t0 = i * per, t1 = min(T, t0 + per);
    auto rad_of = [&](int t) {
        float rad = (f0[t] * mult) / sr;
        return rad - floorf(rad);
    };
    double sum = 0.0;
    for (int t = t0; t < t1; ++t) sum += (double)rad_of(t);
    part[i] = sum;
    __syncthreads();
    if (i == 0) {
        double run = 0.0;
        for (int j = 0; j < 256; ++j) {
            const double v = part[j];
            part[j] = run;
            run += v;
        }
    }
    __syncthreads();
    double cum = part[i];
    for (int t = t0; t < t1; ++t) {
        cum += (double)rad_of(t);
        const float c = (float)cum;
        phase[(long long)t * H + h] = ((c * 2.0f) * 3.14159274101257324f) * up;
    }
}
int launch_hift_phase(const float* f0, float* phase, int T, int H, float sr, int up, hipStream_t s) {
    if (T <= 0) return 0;
    hipLaunchKernelGGL(hift_phase_kernel, dim3(H), dim3(256), 0, s, f0, phase, T, H, sr, (float)up);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("hift_phase launch failed"), -1);
}

// ---- per-sample excitation (SineGen2.forward :289-317 + SourceModuleHnNSF.forward :358-375) -----------------
__global__ void hift_source_kernel(const float* f0, const float* phase, const float* table, const float* w, const float* b, float* s_out,
                                   int T, int H, int up, float amp, float sigma, float vthr) {
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long L = (long long)T * up;
    if (n >= L) return;
    const int t = (int)(n / up);
    const float uv = f0[t] > vthr ? 1.0f : 0.0f;
    const float noise_amp = uv * sigma + ((1.0f - uv) * amp) / 3.0f;
    float acc = 0.0f;
    for (int h = 0; h < H; ++h) {
        const float sine = sinf(phase[(long long)t * H + h]) * amp;
        const float v = sine * uv + noise_amp * table[n * H + h];
        acc += v * w[h];
    }
    s_out[n] = tanhf(acc + b[0]);
}
int launch_hift_source(const float* f0, const float* phase, const float* table, const float* w, const float* b, float* s_out, int T, int H,
                       int up, float amp, float sigma, float vthr, hipStream_t s) {
    if (T <= 0) return 0;
    const long long L = (long long)T * up;
    hipLaunchKernelGGL(hift_source_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, f0, phase, table, w, b, s_out, T, H, up, amp, sigma, vthr);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("hift_source launch failed"), -1);
}

// ---- STFT n_fft=16, hop=4, periodic Hann, center=True / reflect (_stft :491-497) ----------------------------
__global__ void hift_stft_kernel(const float* x, float* spec, int L, int frames, int ld) {
    __shared__ float cs[16], sn[16], wnd[16];
    if (threadIdx.x < 16) {
        const float ang = 6.28318530717958647692f * (float)threadIdx.x / 16.0f;
        cs[threadIdx.x] = cosf(ang);
        sn[threadIdx.x] = sinf(ang);
        wnd[threadIdx.x] = 0.5f - 0.5f * cosf(ang);
    }
    __syncthreads();
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    float xs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        int n = 4 * f + j - 8;
        if (n < 0) n = -n;
        if (n >= L) n = 2 * (L - 1) - n;
        xs[j] = x[n] * wnd[j];
    }
    float* o = spec + (long long)f * ld;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float re = 0.0f, im = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = (j * k) & 15;
            re += xs[j] * cs[m];
            im -= xs[j] * sn[m];
        }
        o[k] = re;
        o[9 + k] = im;
    }
    for (int c = 18; c < ld; ++c) o[c] = 0.0f;
}
int launch_hift_stft(const float* s_in, float* spec, int L, int ld, hipStream_t s) {
    if (L <= 0) return 0;
    const int frames = L / 4 + 1;
    hipLaunchKernelGGL(hift_stft_kernel, dim3((frames + 127) / 128), dim3(128), 0, s, s_in, spec, L, frames, ld);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("hift_stft launch failed"), -1);
}

// ---- magnitude/phase -> iSTFT -> clamp (decode :702-710, _istft :499-505) -----------------------------------
// torch.istft: per frame irfft (imag of DC / Nyquist ignored) * window, overlap-add, divide by the sum of squared
// windows actually covering the sample, trim n_fft/2 at both ends.
__global__ __launch_bounds__(256) void hift_istft_kernel(const float* x, int ld, float* wav, int frames, float limit) {
    constexpr int FR = 64;                       // frames per workgroup -> 256 output samples
    __shared__ float re[FR + 3][9], im[FR + 3][9];
    __shared__ float cs[16], sn[16], wnd[16];
    if (threadIdx.x < 16) {
        const float ang = 6.28318530717958647692f * (float)threadIdx.x / 16.0f;
        cs[threadIdx.x] = cosf(ang);
        sn[threadIdx.x] = sinf(ang);
        wnd[threadIdx.x] = 0.5f - 0.5f * cosf(ang);
    }
    const int fbase = blockIdx.x * FR - 1;       // sample n uses frames floor((n+8)/4)-3 .. floor((n+8)/4)
    for (int i = threadIdx.x; i < (FR + 3) * 9; i += 256) {
        const int fl = i / 9, k = i - fl * 9;
        const int f = fbase + fl;
        float r = 0.0f, m = 0.0f;
        if (f >= 0 && f < frames) {
            const float mag = fminf(expf(x[(long long)f * ld + k]), 100.0f);
            const float ph = sinf(x[(long long)f * ld + 9 + k]);
            r = mag * cosf(ph);
            m = mag * sinf(ph);
        }
        re[fl][k] = r;
        im[fl][k] = m;
    }
    __syncthreads();
    const long long L = 4LL * (frames - 1);
    const long long n = (long long)blockIdx.x * (FR * 4) + threadIdx.x;
    if (n >= L) return;
    const int fhi = (int)((n + 8) >> 2);
    float num = 0.0f, den = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = fhi - 3 + i;
        if (f < 0 || f >= frames) continue;
        const int j = (int)(n + 8 - 4LL * f);
        const int fl = f - fbase;
        float v = re[fl][0] + ((j & 1) ? -re[fl][8] : re[fl][8]);
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const int m = (j * k) & 15;
            v += 2.0f * (re[fl][k] * cs[m] - im[fl][k] * sn[m]);
        }
        v *= (1.0f / 16.0f);
        num += v * wnd[j];
        den += wnd[j] * wnd[j];
    }
    float y = num / den;
    y = fminf(fmaxf(y, -limit), limit);
    wav[n] = y;
}
int launch_hift_istft(const float* x, int ld, float* wav, int frames, float limit, hipStream_t s) {
    if (frames <= 1) return 0;
    const long long L = 4LL * (frames - 1);
    hipLaunchKernelGGL(hift_istft_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, x, ld, wav, frames, limit);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("hift_istft launch failed"), -1);
}

__global__ void copy_row_kernel(float* buf, int ld, int dst_row, int src_row, int cols) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < cols) buf[(long long)dst_row * ld + c] = buf[(long long)src_row * ld + c];
}
int launch_copy_row(float* buf, int ld, int dst_row, int src_row, int cols, hipStream_t s) {
    hipLaunchKernelGGL(copy_row_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, buf, ld, dst_row, src_row, cols);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("copy_row launch failed"), -1);
}

}  // namespace hvx
