// stream_lab.hip — what a launch that only READS n bytes once costs on this chip (cold: the buffers cycled exceed the 256 MB infinity cache).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_lab.hip -o tools/bin/stream_lab ; tools/bin/stream_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));

// every workgroup owns a contiguous byte range; a wave walks it 1 KiB per instruction, U instructions in flight
template <int U>
__global__ __launch_bounds__(256) void stream_kernel(const i32x4* p, long long n16, int* sink) {
    const long long per = (n16 + gridDim.x - 1) / gridDim.x;
    const long long b0 = (long long)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
    i32x4 acc = {0, 0, 0, 0};
    for (long long i = b0 + threadIdx.x; i < b1; i += 256 * U) {
        i32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < b1) ? __builtin_nontemporal_load(p + i + u * 256) : i32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) sink[0] = 1;
}

int main() {
    const size_t sizes[] = {16u << 20, 40u << 20, 80u << 20, 160u << 20};
    int* sink;
    CK(hipMalloc(&sink, 4));
    for (size_t sz : sizes) {
        const int NB = (int)((768u << 20) / sz) + 1;
        std::vector<void*> bufs(NB);
        for (auto& b : bufs) { CK(hipMalloc(&b, sz)); CK(hipMemset(b, 1, sz)); }
        for (int wgs : {256, 512, 1024, 2048}) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto run = [&](int reps) { for (int r = 0; r < reps; ++r) for (int b = 0; b < NB; ++b) hipLaunchKernelGGL(stream_kernel<8>, dim3(wgs), dim3(256), 0, 0, (const i32x4*)bufs[b], (long long)(sz / 16), sink); };
            run(1);
            CK(hipEventRecord(e0)); run(3); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / (3.0 * NB);
            printf("%4zu MB  %5d wgs  U=8   %8.2f us  %6.2f TB/s\n", sz >> 20, wgs, us, (double)sz / us / 1e6);
        }
        for (auto b : bufs) CK(hipFree(b));
    }
    return 0;
}
