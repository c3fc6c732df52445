// store_lab.hip — how fast can the epilogue of a 256 x 256 GEMM tile grid write (or read-modify-write) its fp32 / bf16 output, by store pattern?
// Every workgroup (8 waves as 2 x 4, a wave owns 128 rows x 64 columns of the tile, as in csrc/gemm_big.hip) stores its tile of an [M][N]
// row-major matrix; no GEMM in front.  Development aid:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/store_lab tools/store_lab.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// PAT 0: instruction = 16 rows x 64 B   (4 lanes per row)            — the product's fp32 pattern (PERM 1)
// PAT 1: instruction =  8 rows x 128 B  (8 lanes per row)
// PAT 2: instruction =  4 rows x 256 B  (16 lanes per row: the wave's whole 64-column slice of a row)
// PAT 3: tile-major output: the wave's 128 x 64 slice is 32 KiB contiguous (instruction = 1 KiB contiguous)
template <int PAT, bool RMW, bool NT>
__global__ __launch_bounds__(512) void store_kernel(float* out, int M, int N, int tiles_n, int active) {
    if ((int)blockIdx.x >= active) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qn = nwg >> 3, rn = nwg & 7;
    const int wgid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const int tn = wgid % tiles_n, tm = wgid / tiles_n;
    const int m0 = tm * 256 + (wave / 4) * 128, n0 = tn * 256 + (wave % 4) * 64;
    const f32x4 v = {(float)tid, 1.0f, 2.0f, 3.0f};
    constexpr int LPR = PAT == 0 ? 4 : PAT == 1 ? 8 : 16, RPI = 64 / LPR, IPR = 16 / LPR;   // lanes per row, rows per instruction, instructions per row slice
    if (PAT == 3) {
        float* base = out + ((long long)wgid * 8 + wave) * 128 * 64;
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            f32x4* p = reinterpret_cast<f32x4*>(base + i * 256 + lane * 4);
            f32x4 x = v;
            if (RMW) x += *p;
            if (NT) __builtin_nontemporal_store(x, p); else *p = x;
        }
        return;
    }
#pragma unroll 2
    for (int rb = 0; rb < 128; rb += RPI) {
#pragma unroll
        for (int q = 0; q < IPR; ++q) {
            const int row = m0 + rb + lane / LPR, col = n0 + q * LPR * 4 + (lane % LPR) * 4;
            if (row < M) {
                f32x4* p = reinterpret_cast<f32x4*>(out + (long long)row * N + col);
                f32x4 x = v;
                if (RMW) x += *p;
                if (NT) __builtin_nontemporal_store(x, p); else *p = x;
            }
        }
    }
}

// the reference point: a plain contiguous streaming kernel over the same bytes
template <bool RMW>
__global__ __launch_bounds__(256) void stream_kernel(float* out, long long n4) {
    const f32x4 v = {1.0f, 1.0f, 2.0f, 3.0f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4* p = reinterpret_cast<f32x4*>(out) + i;
        f32x4 x = v;
        if (RMW) x += *p;
        *p = x;
    }
}

template <class F>
static float time_us(F launch, int reps = 20) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const int M = 45056, N = 1024, tiles_n = N / 256, tiles = M / 256 * tiles_n;
    float* out;
    (void)hipMalloc(&out, (size_t)M * N * 4);
    (void)hipMemset(out, 0, (size_t)M * N * 4);
    const double mb = (double)M * N * 4 / 1e6;
#define RUN(name, kern, act, bytes_mb)                                                                              \
    {                                                                                                               \
        const float us = time_us([&] { hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), 0, 0, out, M, N, tiles_n, act); }); \
        printf("  %-66s %8.1f us  %6.2f TB/s\n", name, us, (bytes_mb) / us);                                        \
    }
    printf("[%d][%d] fp32 = %.0f MB, %d tiles\n", M, N, mb, tiles);
    RUN("write  16 rows x 64 B per instruction (product pattern)", (store_kernel<0, false, false>), tiles, mb);
    RUN("write   8 rows x 128 B", (store_kernel<1, false, false>), tiles, mb);
    RUN("write   4 rows x 256 B", (store_kernel<2, false, false>), tiles, mb);
    RUN("write   tile-major (1 KiB contiguous per instruction)", (store_kernel<3, false, false>), tiles, mb);
    RUN("write   8 rows x 128 B, nontemporal", (store_kernel<1, false, true>), tiles, mb);
    RUN("write   tile-major, nontemporal", (store_kernel<3, false, true>), tiles, mb);
    RUN("r-m-w  16 rows x 64 B per instruction (product pattern)", (store_kernel<0, true, false>), tiles, 2 * mb);
    RUN("r-m-w   4 rows x 256 B", (store_kernel<2, true, false>), tiles, 2 * mb);
    RUN("r-m-w   tile-major", (store_kernel<3, true, false>), tiles, 2 * mb);
    RUN("write  16 rows x 64 B, only the first 256 tiles (one round)", (store_kernel<0, false, false>), 256, mb * 256 / tiles);
    RUN("write  16 rows x 64 B, only the first 128 tiles (half the CUs)", (store_kernel<0, false, false>), 128, mb * 128 / tiles);
    RUN("r-m-w  16 rows x 64 B, only the first 256 tiles", (store_kernel<0, true, false>), 256, 2 * mb * 256 / tiles);
    RUN("r-m-w  16 rows x 64 B, only the first 128 tiles", (store_kernel<0, true, false>), 128, 2 * mb * 128 / tiles);
    {
        const long long n4 = (long long)M * N / 4;
        float us = time_us([&] { hipLaunchKernelGGL(stream_kernel<false>, dim3(4096), dim3(256), 0, 0, out, n4); });
        printf("  %-66s %8.1f us  %6.2f TB/s\n", "contiguous grid-stride write", us, mb / us);
        us = time_us([&] { hipLaunchKernelGGL(stream_kernel<true>, dim3(4096), dim3(256), 0, 0, out, n4); });
        printf("  %-66s %8.1f us  %6.2f TB/s\n", "contiguous grid-stride r-m-w", us, 2 * mb / us);
    }
    return 0;
}
