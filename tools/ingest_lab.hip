// ingest_lab.hip — what a compute unit takes in by LDS-DMA per second, by the SHAPE of a wave instruction's 1 KiB: 16 rows x 64 B, 8 rows x 128 B,
// 4 rows x 256 B or one contiguous KiB, rows `stride` bytes apart (the activation planes of the vocoder's convolutions are [rows][C] with 64-byte
// K-steps; gemm_big's tiles are 128-byte rows; the decode GEMMs' fragment order is the contiguous case).  Loop shape of gemm_x3p_kernel: 4 waves, 32 KB
// per step, double buffer, one barrier per step, nothing else (no LDS reads, no MFMAs).  The source is `mb` MB (L2 / infinity-cache resident when small).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ingest_lab.hip -o tools/bin/ingest_lab ; tools/bin/ingest_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

// PIECE = bytes per row piece (64 / 128 / 256 / 1024); a wave instruction covers 1024 / PIECE rows
// PLAIN: global_load_dwordx4 into registers and ds_write_b128 (the path of attn_dit_kernel's K / V^T tiles) instead of LDS-DMA
template <int PIECE, int PLAIN = 0>
__global__ __launch_bounds__(256) void ingest_kernel(const char* src, long long bytes, int stride, int steps, int* sink) {
    __shared__ __attribute__((aligned(1024))) char buf[2][32768];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = PIECE / 16;                         // lanes per row piece
    const int row = lane / LPR, col = (lane % LPR) * 16;
    constexpr int RPI = 1024 / PIECE;                       // rows per instruction
    // a workgroup walks its own region: step s, instruction q (8 per wave and step), rows (q * 4 + wave) * RPI + row of a 32 KB / PIECE-row tile
    const long long tile_rows = 32768 / PIECE;
    const long long W = bytes / 2, ext = tile_rows * stride;                     // window of starts; a tile reaches `ext` bytes beyond its start (<= 1 MB)
    const long long start = ((long long)blockIdx.x * 2654435761LL) % W;
    for (int s = 0; s < steps; ++s) {
        __syncthreads();
        const char* tb = src + (((start + (long long)s * ext) % W) & ~1023LL);
        if constexpr (PLAIN) {
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            i32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const i32x4*>(tb + (long long)((q * 4 + wave) * RPI + row) * stride + col);
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<i32x4*>(buf[s & 1] + (q * 4 + wave) * 1024 + lane * 16) = v[q];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = (q * 4 + wave) * RPI + row;
                __builtin_amdgcn_global_load_lds((glb_ptr)(tb + (long long)r * stride + col), (lds_ptr)(buf[s & 1] + (q * 4 + wave) * 1024), 16, 0, 0);
            }
        }
    }
    __syncthreads();
    if (buf[0][threadIdx.x] == 77 && buf[1][threadIdx.x] == 78) sink[0] = 1;
}

// The DMA stream of gemm_x3p_kernel<128,128> on a k-tap convolution over C = 128 channels and nothing else: per K-step (32 channels of one tap) a
// workgroup brings 128 activation rows x 64 B x 2 planes and 128 weight rows x 64 B x 2 planes into a double buffer, one barrier per step; consecutive
// steps walk along the rows (4 per tap), the next tap is the tile shifted by `dil` rows.  `pitch` = bytes between activation rows (256 = dense C = 128).
__global__ __launch_bounds__(256) void conv_walk_kernel(const char* act, long long plane, int pitch, const char* w, int taps, int dil, int rows, int* sink) {
    __shared__ __attribute__((aligned(1024))) char buf[2][32768];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane >> 2, lcol = (lane & 3) * 16;
    const long long m0 = (long long)blockIdx.x * 128;
    const int kbytes = taps * 256;                          // a weight row: taps x 128 channels x 2 B
    int s = 0;
    for (int tap = 0; tap < taps; ++tap)
        for (int ci = 0; ci < 4; ++ci, ++s) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r16 = (wave * 2 + q) * 16;
                long long m = m0 + r16 + lrow + (long long)tap * dil;
                m = m < rows ? m : rows - 1;
                const char* gp = act + m * pitch + ci * 64 + lcol;
                __builtin_amdgcn_global_load_lds((glb_ptr)gp, (lds_ptr)(buf[s & 1] + r16 * 64), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_ptr)(gp + plane), (lds_ptr)(buf[s & 1] + 8192 + r16 * 64), 16, 0, 0);
                const char* wp = w + (long long)(r16 + lrow) * kbytes + tap * 256 + ci * 64 + lcol;
                __builtin_amdgcn_global_load_lds((glb_ptr)wp, (lds_ptr)(buf[s & 1] + 16384 + r16 * 64), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_ptr)(wp + 128 * kbytes), (lds_ptr)(buf[s & 1] + 24576 + r16 * 64), 16, 0, 0);
            }
        }
    __syncthreads();
    if (buf[0][threadIdx.x] == 77 && buf[1][threadIdx.x] == 78) sink[0] = 1;
}

void run_conv(int pitch, int taps, int dil, int* sink) {
    const int rows = 225280;                                // the C = 128 stage of a 5632-frame utterance
    const long long plane = (long long)rows * pitch;
    char *act, *w;
    CK(hipMalloc(&act, 2 * plane)); CK(hipMemset(act, 1, 2 * plane));
    CK(hipMalloc(&w, 2LL * 128 * taps * 256)); CK(hipMemset(w, 1, 2LL * 128 * taps * 256));
    const int wgs = rows / 128;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(conv_walk_kernel, dim3(wgs), dim3(256), 0, 0, act, plane, pitch, w, taps, dil, rows, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv_walk_kernel, dim3(wgs), dim3(256), 0, 0, act, plane, pitch, w, taps, dil, rows, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, total = (double)wgs * taps * 4 * 32768;
    printf("conv walk C=128 taps %2d dil %d  act pitch %4d B  %4d wgs  %8.1f us  %6.2f TB/s into LDS  %6.1f GB/s per CU\n", taps, dil, pitch, wgs, us, total / us / 1e6, total / us / 1e3 / 256);
    CK(hipFree(act)); CK(hipFree(w));
}

template <int PIECE, int PLAIN = 0>
void run(const char* src, long long bytes, int stride, int wgs, int* sink, const char* what) {
    const int steps = 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((ingest_kernel<PIECE, PLAIN>), dim3(wgs), dim3(256), 0, 0, src, bytes, stride, steps, sink);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ingest_kernel<PIECE, PLAIN>), dim3(wgs), dim3(256), 0, 0, src, bytes, stride, steps, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, total = (double)wgs * steps * 32768;
    printf("%s %-18s piece %4d B  stride %5d  %4d wgs  %8.1f us  %6.2f TB/s  %6.1f GB/s per CU\n", PLAIN ? "plain" : "dma  ", what, PIECE, stride, wgs, us, total / us / 1e6, total / us / 1e3 / 256);
}

int main() {
    int* sink; CK(hipMalloc(&sink, 4));
    for (int taps : {3, 7, 11})
        for (int pitch : {256, 320, 512, 1024}) run_conv(pitch, taps, taps == 3 ? 1 : 3, sink);
    for (long long mb : {24LL, 512LL}) {
        char* src; CK(hipMalloc(&src, mb << 20)); CK(hipMemset(src, 1, mb << 20));
        char what[64]; snprintf(what, sizeof what, "source %lld MB", mb);
        for (int wgs : {256, 512}) {
            run<64>(src, mb << 20, 256, wgs, sink, what);     // C = 128 planes: 64-byte K-steps of 256-byte rows
            run<64>(src, mb << 20, 1024, wgs, sink, what);
            run<128>(src, mb << 20, 256, wgs, sink, what);
            run<128>(src, mb << 20, 2048, wgs, sink, what);   // gemm_big: 128-byte K-tiles of 2 KB rows
            run<256>(src, mb << 20, 256, wgs, sink, what);
            run<1024>(src, mb << 20, 1024, wgs, sink, what);  // contiguous (fragment order)
            run<128, 1>(src, mb << 20, 128, wgs, sink, what);  // attention's K tile: 128-byte rows back to back (contiguous)
            run<128, 1>(src, mb << 20, 256, wgs, sink, what);
            run<128, 1>(src, mb << 20, 2048, wgs, sink, what);
            run<64, 1>(src, mb << 20, 256, wgs, sink, what);
            run<1024, 1>(src, mb << 20, 1024, wgs, sink, what);
        }
        CK(hipFree(src));
    }
    return 0;
}
