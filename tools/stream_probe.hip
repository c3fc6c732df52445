// stream_probe.hip — how fast can G workgroups of W waves pull a contiguous slab each out of HBM on gfx950?
// Shapes the decode GEMMs (gemm_skinny.hip): their weight stream is `G` slabs of `bytes_per_wg`, every byte read once.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// each wave reads `trips` x U x 1 KiB, U loads in flight per trip
template <int U, bool NT>
__global__ void probe(const f32x4* __restrict__ src, float* sink, long long wg_stride_v4, int trips) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const f32x4* p = src + (long long)blockIdx.x * wg_stride_v4 + (long long)wave * trips * U * 64 + lane;
    f32x4 acc = {0, 0, 0, 0};
    for (int t = 0; t < trips; ++t) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + (t * U + u) * 64) : p[(t * U + u) * 64];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = 1.0f;
    (void)nw;
}

template <int U, bool NT>
static void run(const f32x4* buf, float* sink, int G, int W, long long total_bytes, size_t buf_bytes) {
    const long long per_wg = total_bytes / G;
    int trips = (int)(per_wg / (W * U * 1024LL));
    if (trips < 1) trips = 1;
    const long long stride_v4 = (long long)W * trips * U * 64;
    if ((size_t)G * stride_v4 * 16 > buf_bytes) return;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 20;
    // rotate over distinct regions so that nothing is served from L2 / MALL
    const long long region_v4 = (long long)G * stride_v4;
    const int n_regions = (int)(buf_bytes / 16 / region_v4);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<U, NT>), dim3(G), dim3(64 * W), 0, 0, buf, sink, stride_v4, trips);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((probe<U, NT>), dim3(G), dim3(64 * W), 0, 0, buf + (long long)(i % n_regions) * region_v4, sink, stride_v4, trips);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double bytes = (double)G * stride_v4 * 16;
    printf("G=%4d W=%2d U=%2d nt=%d trips=%3d  %7.1f KB/wg  %6.2f MB  %6.2f us/launch  %7.1f GB/s  %5.1f GB/s/wg\n", G, W, U, (int)NT, trips,
           stride_v4 * 16 / 1024.0, bytes / 1e6, us, bytes / us / 1e3, bytes / us / 1e3 / G);
}

int main() {
    const size_t buf_bytes = 2ull << 30;
    f32x4* buf;
    float* sink;
    hipMalloc(&buf, buf_bytes);
    hipMalloc(&sink, 4);
    hipMemset(buf, 0, buf_bytes);
    hipDeviceSynchronize();
    const long long sizes[] = {2LL << 20, 9LL << 20, 18LL << 20};
    for (long long total : sizes) {
        printf("---- total %.1f MB per launch\n", total / 1e6);
        for (int G : {56, 112, 224, 256, 512, 1024}) {
            for (int W : {4, 8, 16}) {
                run<8, true>(buf, sink, G, W, total, buf_bytes);
            }
        }
        run<4, true>(buf, sink, 256, 4, total, buf_bytes);
        run<16, true>(buf, sink, 256, 4, total, buf_bytes);
        run<8, false>(buf, sink, 256, 4, total, buf_bytes);
        run<8, false>(buf, sink, 56, 16, total, buf_bytes);
    }
    // empty-kernel floor
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((probe<8, true>), dim3(256), dim3(256), 0, 0, buf, sink, 0, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("empty 256x256 kernel: %.2f us/launch (eager, back to back)\n", ms * 1e3 / 200);
    }
    return 0;
}
