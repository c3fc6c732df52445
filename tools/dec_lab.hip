// dec_lab.hip — lab bench for the decode-step GEMM forms at the wide-grid shape (M = 128 rows = 64 sequences x 2 heads).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops tools/dec_lab.hip -o gpurun_out/dec_lab
//   gpurun_out/dec_lab [shape: swiglu|qkv|oproj|down] [reps]
//
// Weights are in the MFMA fragment order of the product ([N/16][K/32][64 lanes][8], packing.pack_frag); 24 distinct weight buffers are cycled so
// that no launch finds its weights in the 256 MB infinity cache.  Every form is checked against a CPU reference on buffer 0.
//
// Forms:
//   stream   : read the weight bytes once with all CUs and do nothing else (the floor of a launch that streams them)
//   ring     : A-stationary.  The 4 waves of a workgroup split the ROWS (MTW 16-row tiles each) and keep their activation fragments of the whole K
//              in registers; the weights of the workgroup's column tiles go ONCE through an LDS ring by LDS-DMA (1 KiB fragment per wave
//              instruction, D stages in flight), every wave reads every fragment (conflict-free ds_read_b128) and feeds MTW MFMAs with it.
//              No K split, no cross-wave reduction; counted vmcnt waits + bare s_barrier per stage.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <type_traits>
#include <utility>
#include <vector>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

enum { EP_STORE = 0, EP_SWIGLU = 1 };

struct Args {
    const bf16_t* A; int lda; int M;
    const bf16_t* W; int N, K;         // packed
    bf16_t* out; int ldo;              // EP_STORE: [M][N]; EP_SWIGLU: [M][N/2]
    float* part;                       // split-K partials [split][M][N] (EP_STORE with split > 1)
    int n_groups, gpw;
    int afrag;                         // A is stored in fragment order [M/16][K/32][64 lanes][8]
};

template <int OFF> __device__ __forceinline__ i32x4 lds_read16(unsigned addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int CNT> __device__ __forceinline__ void lds_wait(i32x4& v) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(CNT)); }
template <int CNT> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory"); }

__global__ __launch_bounds__(256) void stream_kernel(const bf16_t* W, long long n16, float* sink) {
    // every thread reads 16-byte pieces, grid-strided; 8 in flight
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.0f;
    const i32x4* p = reinterpret_cast<const i32x4*>(W);
    for (; i + 7 * stride < n16; i += 8 * stride) {
        i32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (float)(v[u][0] ^ v[u][3]);
    }
    for (; i < n16; i += stride) s += (float)p[i][1];
    if (s == 1.2345f) sink[0] = s;
}

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// MTW row tiles per wave, NTG column tiles per group (SwiGLU: a (gate, up) pair), KT k-steps (the whole K of the workgroup),
// SF k-steps per stage, D ring stages
template <int MTW, int NTG, int EPI, int KT, int SF, int D, int NT_AUX>
__global__ __launch_bounds__(256) void ring_kernel(Args a) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass has no __amdgpu_buffer_rsrc_t and drops the stub of a kernel whose body it cannot build)
    constexpr int NST = (KT + SF - 1) / SF;
    constexpr int SFN = SF * NTG;
    constexpr int LPS = (SFN + 3) / 4;
    __shared__ __attribute__((aligned(1024))) char ring[(D * SFN + 4) * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int g0 = blockIdx.x * a.gpw;
    const int ng = min(a.n_groups - g0, a.gpw);
    const int KTT = a.K >> 5;
    const int ks0 = blockIdx.y * KT;
    const int m0 = blockIdx.z * (64 * MTW) + wave * (16 * MTW);
    if (ng <= 0) return;

    // activation fragments of this wave's rows, whole K of the workgroup
    bf16x8 af[KT][MTW];
    {
        const bf16_t* ap[MTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            int r = m0 + i * 16 + fr;
            r = r < a.M ? r : a.M - 1;
            ap[i] = a.afrag ? a.A + ((long long)((m0 >> 4) + i) * KTT + ks0) * 512 + lane * 8 : a.A + (long long)r * a.lda + ks0 * 32 + fg * 8;
        }
        const int kstride = a.afrag ? 512 : 32;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks)
#pragma unroll
            for (int i = 0; i < MTW; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(ap[i] + ks * kstride);
    }

    // weight stream of this workgroup: group gi starts at byte wg_cur = ((g0 + gi) * NTG * KTT + ks0) * 1024; within a stage, DMA instruction li of
    // wave w moves fragment q = 4 li + w = (k-step q / NTG, tile q % NTG) of the stage.  buffer_load ... lds (MUBUF), not global_load_lds: the
    // FLAT-encoded form makes hipcc's waitcnt pass treat every later vector-memory wait as vmcnt(0) + lgkmcnt(0) ("pending flat")
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, a.N * a.K * 2, 0x00020000);
    const int GS = NTG * KTT * 1024;
    int wg_cur = (g0 * NTG * KTT + ks0) * 1024;
    int foff[LPS];
#pragma unroll
    for (int li = 0; li < LPS; ++li) {
        const int q = li * 4 + wave;
        foff[li] = ((q % NTG) * KTT + q / NTG) * 1024;
    }
    char* const dummy = ring + (D * SFN + wave) * 1024;
    // issue stage (group gi + carry, st) into ring slot `slot`; past the last group the loads still happen (uniform vmcnt accounting) but
    // re-read the current group into a dummy slot
    auto issue = [&](int gi, auto CARRY, auto ST, int slot) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, carry = decltype(CARRY)::value;
        constexpr int nf = (KT - st * SF < SF ? KT - st * SF : SF) * NTG;
        const bool live = gi + carry < ng;
        const int src = wg_cur + (live ? carry * GS : 0) + st * SF * 1024;
        char* const dst = ring + (slot * SFN + wave) * 1024;
#pragma unroll
        for (int li = 0; li < LPS; ++li) {
            bool ok = live;
            if (li * 4 + 3 >= nf) ok = ok && (li * 4 + wave < nf);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(ok ? dst + li * 4096 : dummy), 16, lane * 16, src + (ok ? foff[li] : 0), 0, NT_AUX);
        }
    };
    // prologue: stages 0 .. D-2
    static_for<0, D - 1>([&](auto P) {
        constexpr int p = decltype(P)::value;
        issue(0, std::integral_constant<int, p / NST>{}, std::integral_constant<int, p % NST>{}, p);
    });
    // (a use of the last activation fragment outside the loop: the compiler then waits for the activation loads with a counted vmcnt here instead of
    // flushing vmcnt(0) — DMA prologue included — in the loop preheader)
    asm volatile("" ::"v"(af[KT - 1][MTW - 1]));

    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    int slot = 0;
    for (int gi = 0; gi < ng; ++gi) {
        f32x4 acc[MTW][NTG];
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int j = 0; j < NTG; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        static_for<0, NST>([&](auto ST) {
            constexpr int st = decltype(ST)::value;
            constexpr int nf = (KT - st * SF < SF ? KT - st * SF : SF) * NTG;
            vm_wait<LPS*(D - 2)>();
            __builtin_amdgcn_s_barrier();
            {
                int sn = slot + (D - 1);
                sn = sn >= D ? sn - D : sn;
                issue(gi, std::integral_constant<int, (st + D - 1) / NST>{}, std::integral_constant<int, (st + D - 1) % NST>{}, sn);
            }
            const unsigned addr = ring_base + slot * (SFN * 1024);
            i32x4 b[nf];
            static_for<0, nf>([&](auto Q) { b[decltype(Q)::value] = lds_read16<decltype(Q)::value * 1024>(addr); });
            static_for<0, nf>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                lds_wait<(nf - 1 - q < 15 ? nf - 1 - q : 15)>(b[q]);
                const bf16x8 bb = __builtin_bit_cast(bf16x8, b[q]);
#pragma unroll
                for (int i = 0; i < MTW; ++i) acc[i][q % NTG] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[st * SF + q / NTG][i], bb, acc[i][q % NTG], 0, 0, 0);
            });
            slot = slot + 1 == D ? 0 : slot + 1;
        });
        wg_cur += GS;
        // ---- epilogue (C layout: col = fr, rows = fg * 4 + r)
        const int tile0 = (g0 + gi) * NTG;
        if constexpr (EPI == EP_SWIGLU) {
            static_assert(EPI != EP_SWIGLU || NTG == 2, "pairs");
            const int col = (tile0 >> 1) * 16 + fr;
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + i * 16 + fg * 4 + r;
                    if (row >= a.M) continue;
                    const float g = acc[i][0][r], u = acc[i][1][r];
                    const float v = g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g)) * u;
                    a.out[(long long)row * a.ldo + col] = (bf16_t)v;
                }
        } else {
#pragma unroll
            for (int j = 0; j < NTG; ++j) {
                const int col = (tile0 + j) * 16 + fr;
#pragma unroll
                for (int i = 0; i < MTW; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + i * 16 + fg * 4 + r;
                        if (row >= a.M) continue;
                        if (a.part) a.part[((long long)blockIdx.y * a.M + row) * a.N + col] = acc[i][j][r];
                        else a.out[(long long)row * a.ldo + col] = (bf16_t)acc[i][j][r];
                    }
            }
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
static float bf2f(bf16_t v) { return (float)v; }

struct Shape { const char* name; int N, K, epi, split; };

template <int MTW, int NTG, int EPI, int KT, int SF, int D, int NT_AUX>
static void run_ring(const char* label, const Shape& sh, int M, const bf16_t* dA, std::vector<bf16_t*>& dW, bf16_t* dOut, float* dPart, int gpw, int reps,
                     const std::vector<float>& ref, int ldo, int afrag = 0) {
    const int n_groups = sh.N / 16 / NTG;
    const int mch = (M + 64 * MTW - 1) / (64 * MTW);
    const int split = sh.K / 32 / KT;
    dim3 grid((n_groups + gpw - 1) / gpw, split, mch);
    Args a{dA, sh.K, M, dW[0], sh.N, sh.K, dOut, ldo, split > 1 ? dPart : nullptr, n_groups, gpw, afrag};
    CK(hipMemset(dOut, 0, (size_t)M * ldo * 2));
    hipLaunchKernelGGL((ring_kernel<MTW, NTG, EPI, KT, SF, D, NT_AUX>), grid, dim3(256), 0, 0, a);
    CK(hipDeviceSynchronize());
    // check
    double maxerr = 0.0;
    if (split == 1) {
        std::vector<bf16_t> h((size_t)M * ldo);
        CK(hipMemcpy(h.data(), dOut, h.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); ++i) {
            const double e = fabs(bf2f(h[i]) - ref[i]) / (fabs(ref[i]) + 1.0);
            if (e > maxerr) maxerr = e;
        }
    } else {
        std::vector<float> h((size_t)split * M * sh.N);
        CK(hipMemcpy(h.data(), dPart, h.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < (size_t)M * sh.N; ++i) {
            double s = 0;
            for (int k = 0; k < split; ++k) s += h[(size_t)k * M * sh.N + i];
            const double e = fabs(s - ref[i]) / (fabs(ref[i]) + 1.0);
            if (e > maxerr) maxerr = e;
        }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w)
        for (size_t b = 0; b < dW.size(); ++b) {
            a.W = dW[b];
            hipLaunchKernelGGL((ring_kernel<MTW, NTG, EPI, KT, SF, D, NT_AUX>), grid, dim3(256), 0, 0, a);
        }
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r)
        for (size_t b = 0; b < dW.size(); ++b) {
            a.W = dW[b];
            hipLaunchKernelGGL((ring_kernel<MTW, NTG, EPI, KT, SF, D, NT_AUX>), grid, dim3(256), 0, 0, a);
        }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * dW.size());
    printf("%-8s %-30s%s grid %4d x %d x %d  gpw %3d  %7.2f us  %6.2f TB/s(W)  maxrel %.2e\n", sh.name, label, afrag ? " AF" : "   ", grid.x, grid.y, grid.z, gpw, us,
           (double)sh.N * sh.K * 2 / us / 1e6, maxerr);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const std::string which = argc > 1 ? argv[1] : "swiglu";
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int M = argc > 3 ? atoi(argv[3]) : 128;
    const Shape shapes[] = {{"swiglu", 9728, 896, EP_SWIGLU, 1}, {"qkv", 1152, 896, EP_STORE, 1}, {"oproj", 896, 896, EP_STORE, 1}, {"down", 896, 4864, EP_STORE, 8},
                            {"hmlp", 44032, 896, EP_SWIGLU, 1}};
    Shape sh = shapes[0];
    for (const Shape& s : shapes)
        if (which == s.name) sh = s;
    const int NB = which == "hmlp" ? 8 : 24;
    const int KT = sh.K / 32;
    srand(7);
    std::vector<bf16_t> hA((size_t)M * sh.K), hW((size_t)sh.N * sh.K);
    for (auto& v : hA) v = (bf16_t)((rand() % 2001 - 1000) / 1000.0f);
    for (auto& v : hW) v = (bf16_t)((rand() % 2001 - 1000) / 1000.0f / sqrtf((float)sh.K));
    // reference on buffer 0 (packed layout: element (n, k) at ((n/16) * KT + k/32) * 512 + ((n%16) + 16 * ((k%32)/8)) * 8 + k%8)
    const int ldo = sh.epi == EP_SWIGLU ? sh.N / 2 : sh.N;
    std::vector<float> ref((size_t)M * ldo);
    {
        std::vector<float> wf((size_t)sh.N * sh.K), af((size_t)M * sh.K);
        for (int n = 0; n < sh.N; ++n)
            for (int k = 0; k < sh.K; ++k)
                wf[(size_t)n * sh.K + k] = bf2f(hW[((size_t)(n / 16) * KT + k / 32) * 512 + ((n % 16) + 16 * ((k % 32) / 8)) * 8 + k % 8]);
        for (size_t i = 0; i < af.size(); ++i) af[i] = bf2f(hA[i]);
        std::vector<float> full((size_t)M * sh.N);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < sh.N; ++n) {
                double s = 0;
                for (int k = 0; k < sh.K; ++k) s += (double)af[(size_t)m * sh.K + k] * wf[(size_t)n * sh.K + k];
                full[(size_t)m * sh.N + n] = (float)s;
            }
        if (sh.epi == EP_SWIGLU) {
            for (int m = 0; m < M; ++m)
                for (int p = 0; p < sh.N / 32; ++p)
                    for (int c = 0; c < 16; ++c) {
                        const float g = full[(size_t)m * sh.N + (2 * p) * 16 + c], u = full[(size_t)m * sh.N + (2 * p + 1) * 16 + c];
                        ref[(size_t)m * ldo + p * 16 + c] = g / (1.0f + expf(-g)) * u;
                    }
        } else {
            ref = full;
        }
    }
    bf16_t* dA;
    CK(hipMalloc(&dA, hA.size() * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    bf16_t* dAf;
    {
        const int MP = (M + 15) / 16 * 16;
        std::vector<bf16_t> hAf((size_t)MP * sh.K, (bf16_t)0.0f);
        for (int m = 0; m < M; ++m)
            for (int k = 0; k < sh.K; ++k) hAf[((size_t)(m / 16) * KT + k / 32) * 512 + ((m % 16) + 16 * ((k % 32) / 8)) * 8 + k % 8] = hA[(size_t)m * sh.K + k];
        CK(hipMalloc(&dAf, hAf.size() * 2));
        CK(hipMemcpy(dAf, hAf.data(), hAf.size() * 2, hipMemcpyHostToDevice));
    }
    std::vector<bf16_t*> dW(NB);
    for (int b = 0; b < NB; ++b) {
        CK(hipMalloc(&dW[b], hW.size() * 2));
        CK(hipMemcpy(dW[b], hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    }
    bf16_t* dOut;
    CK(hipMalloc(&dOut, (size_t)M * sh.N * 2));
    float* dPart;
    CK(hipMalloc(&dPart, (size_t)16 * M * sh.N * 4));
    float* dSink;
    CK(hipMalloc(&dSink, 64));

    // ---- floor: stream the weights
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int wgs : {256, 512, 1024}) {
            for (int b = 0; b < NB; ++b) hipLaunchKernelGGL(stream_kernel, dim3(wgs), dim3(256), 0, 0, dW[b], (long long)hW.size() / 8, dSink);
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r)
                for (int b = 0; b < NB; ++b) hipLaunchKernelGGL(stream_kernel, dim3(wgs), dim3(256), 0, 0, dW[b], (long long)hW.size() / 8, dSink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / (reps * NB);
            printf("%-8s stream %4d wgs                                                  %7.2f us  %6.2f TB/s(W)\n", sh.name, wgs, us, (double)hW.size() * 2 / us / 1e6);
        }
    }
#define RING(MTW, NTG, EPI, KT_, SF, D, AUX, GPW) run_ring<MTW, NTG, EPI, KT_, SF, D, AUX>("ring<" #MTW "," #NTG "," #KT_ "," #SF "," #D "," #AUX ">", sh, M, dA, dW, dOut, dPart, GPW, reps, ref, ldo)
#define RINGF(MTW, NTG, EPI, KT_, SF, D, AUX, GPW) run_ring<MTW, NTG, EPI, KT_, SF, D, AUX>("ring<" #MTW "," #NTG "," #KT_ "," #SF "," #D "," #AUX ">", sh, M, dAf, dW, dOut, dPart, GPW, reps, ref, ldo, 1)
    if (which == "hmlp") {                  // one MTP head's gate / up projection (79 MB of weights), M = 64 rows
        for (int gpw : {6, 11, 22}) RINGF(1, 2, EP_SWIGLU, 28, 7, 4, 0, gpw);
        for (int gpw : {6, 11}) RINGF(1, 2, EP_SWIGLU, 28, 7, 8, 0, gpw);
        for (int gpw : {6, 11}) RINGF(1, 2, EP_SWIGLU, 28, 4, 10, 0, gpw);
        for (int gpw : {11}) RINGF(1, 2, EP_SWIGLU, 28, 14, 3, 0, gpw);
    } else if (which == "swiglu") {
        for (int gpw : {2, 3}) RING(1, 2, EP_SWIGLU, 28, 7, 4, 0, gpw);
        for (int gpw : {2, 3}) RINGF(1, 2, EP_SWIGLU, 28, 7, 4, 0, gpw);
        for (int gpw : {2, 3}) RINGF(1, 2, EP_SWIGLU, 28, 7, 6, 0, gpw);
        for (int gpw : {2, 3}) RINGF(1, 2, EP_SWIGLU, 28, 14, 3, 0, gpw);
        for (int gpw : {2}) RINGF(2, 2, EP_SWIGLU, 28, 7, 4, 0, gpw);
        for (int gpw : {5, 10}) RINGF(2, 2, EP_SWIGLU, 28, 7, 4, 0, gpw);
        for (int gpw : {5, 10}) RINGF(2, 2, EP_SWIGLU, 28, 4, 8, 0, gpw);
    } else if (which == "qkv" || which == "oproj") {
        for (int gpw : {1}) RING(1, 1, EP_STORE, 28, 7, 4, 0, gpw);
        for (int gpw : {1, 2}) RINGF(1, 1, EP_STORE, 28, 7, 4, 0, gpw);
        for (int gpw : {1}) RINGF(1, 1, EP_STORE, 28, 14, 2, 0, gpw);
        for (int gpw : {1}) RINGF(1, 1, EP_STORE, 28, 4, 7, 0, gpw);
        for (int gpw : {1}) RINGF(2, 1, EP_STORE, 28, 7, 4, 0, gpw);
    } else {
        for (int gpw : {2}) RING(1, 2, EP_STORE, 19, 10, 3, 0, gpw);
        for (int gpw : {1, 2, 3}) RINGF(1, 2, EP_STORE, 19, 10, 3, 0, gpw);
        for (int gpw : {2}) RINGF(1, 2, EP_STORE, 19, 19, 2, 0, gpw);
        for (int gpw : {1, 2}) RINGF(1, 2, EP_STORE, 38, 19, 3, 0, gpw);
        for (int gpw : {2}) RINGF(2, 1, EP_STORE, 19, 19, 2, 0, gpw);
    }
    return 0;
}
