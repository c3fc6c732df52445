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
