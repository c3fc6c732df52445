// mfma_valu_lab.hip — how many vector instructions hide in the shadow of one MFMA on gfx950, per MFMA shape, per kind of vector instruction, for one and two waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/mfma_valu_lab tools/mfma_valu_lab.hip && tools/bin/mfma_valu_lab
// Every wave runs ITER iterations of a fixed, hand-ordered stream (inline asm, nothing for the compiler to move): 4 MFMAs on 4 INDEPENDENT accumulators, each followed by
// NV vector instructions on registers no MFMA touches.  Reported: nanoseconds per MFMA of one SIMD (wall time x SIMDs' share), relative to the NV = 0 stream of the same
// shape and occupancy — the marginal cost of a vector instruction beside the matrix pipe.  DEP = 1: the 4 MFMAs share ONE accumulator (the accumulate chain).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP1(x) x
#define REP2(x) x x
#define REP3(x) x x x
#define REP4(x) x x x x
#define REP6(x) x x x x x x
#define REP8(x) x x x x x x x x

template <int SHAPE, int NV, int VK, int DEP>
__global__ __launch_bounds__(256, 2) void lab(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x * 3 + e)); }
    float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1.0f, v2 = v0 + 2.0f, v3 = v0 + 3.0f;
    if constexpr (SHAPE == 32) {
        f32x16 c0, c1, c2, c3;
        for (int r = 0; r < 16; ++r) { c0[r] = 0; c1[r] = 0; c2[r] = 0; c3[r] = 0; }
        for (int it = 0; it < iters; ++it) {
#define VOPS32                                                                                                                           \
    if constexpr (NV >= 1) { if constexpr (VK == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v0)); else asm volatile("v_add_f32 %0, %0, %0" : "+v"(v0)); }   \
    if constexpr (NV >= 2) { if constexpr (VK == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v1)); else asm volatile("v_add_f32 %0, %0, %0" : "+v"(v1)); }   \
    if constexpr (NV >= 3) { if constexpr (VK == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v2)); else asm volatile("v_add_f32 %0, %0, %0" : "+v"(v2)); }   \
    if constexpr (NV >= 4) { if constexpr (VK == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v3)); else asm volatile("v_add_f32 %0, %0, %0" : "+v"(v3)); }   \
    if constexpr (NV >= 5) { if constexpr (VK == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v0)); else asm volatile("v_add_f32 %0, %0, %0" : "+v"(v0)); }   \
    if constexpr (NV >= 6) { if constexpr (VK == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v1)); else asm volatile("v_add_f32 %0, %0, %0" : "+v"(v1)); }   \
    if constexpr (NV >= 7) { if constexpr (VK == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v2)); else asm volatile("v_add_f32 %0, %0, %0" : "+v"(v2)); }   \
    if constexpr (NV >= 8) { if constexpr (VK == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v3)); else asm volatile("v_add_f32 %0, %0, %0" : "+v"(v3)); }
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            VOPS32
            if constexpr (DEP) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
            VOPS32
            if constexpr (DEP) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
            VOPS32
            if constexpr (DEP) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
            VOPS32
        }
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        float s = v0 + v1 + v2 + v3;
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
        if (s == 123.456f) out[threadIdx.x] = s;
    } else {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            VOPS32
            if constexpr (DEP) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
            VOPS32
            if constexpr (DEP) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
            VOPS32
            if constexpr (DEP) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
            VOPS32
        }
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        float s = v0 + v1 + v2 + v3;
        for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
        if (s == 123.456f) out[threadIdx.x] = s;
    }
}

template <int SHAPE, int NV, int VK, int DEP>
static double run(int waves_per_simd, float* out) {
    const int iters = 20000;
    const dim3 grid(256 * waves_per_simd), block(256);          // 256 CUs x (waves_per_simd workgroups of 4 waves): one wave of each workgroup per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((lab<SHAPE, NV, VK, DEP>), grid, block, 0, 0, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((lab<SHAPE, NV, VK, DEP>), grid, block, 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)ms * 1e6 / ((double)iters * 4 * waves_per_simd);          // ns per MFMA of one SIMD
}

template <int SHAPE, int VK, int DEP>
static void sweep(const char* name, float* out) {
    for (int w = 1; w <= 2; ++w) {
        const double t[9] = {run<SHAPE, 0, VK, DEP>(w, out), run<SHAPE, 1, VK, DEP>(w, out), run<SHAPE, 2, VK, DEP>(w, out), run<SHAPE, 3, VK, DEP>(w, out), run<SHAPE, 4, VK, DEP>(w, out),
                             run<SHAPE, 5, VK, DEP>(w, out), run<SHAPE, 6, VK, DEP>(w, out), run<SHAPE, 7, VK, DEP>(w, out), run<SHAPE, 8, VK, DEP>(w, out)};
        printf("%-34s %d wave(s)/SIMD: ns per MFMA with 0..8 vector instructions behind each:", name, w);
        for (int n = 0; n < 9; ++n) printf(" %6.2f", t[n]);
        printf("\n");
    }
}

int main() {
    float* out;
    hipMalloc(&out, 1 << 20);
    sweep<32, 0, 0>("32x32x16, v_exp_f32, independent", out);
    sweep<32, 1, 0>("32x32x16, v_add_f32, independent", out);
    sweep<16, 0, 0>("16x16x32, v_exp_f32, independent", out);
    sweep<16, 1, 0>("16x16x32, v_add_f32, independent", out);
    sweep<32, 0, 1>("32x32x16, v_exp_f32, ONE accumulator", out);
    sweep<16, 0, 1>("16x16x32, v_exp_f32, ONE accumulator", out);
    sweep<32, 1, 1>("32x32x16, v_add_f32, ONE accumulator", out);
    sweep<16, 1, 1>("16x16x32, v_add_f32, ONE accumulator", out);
    return 0;
}
