// gemm_dec.hip — the four backbone GEMMs of the AR decode step on a WIDE grid (33..256 rows = 17..128 sequences x K heads), bf16, and the two
// largest GEMMs of its MTP heads: gate / up (K heads stacked over blockIdx.y, each with its own weights and its own fragment-order rows) and the
// shared output projection (the K heads' rows as one stacked matrix).
//
// What bounds these launches (tools/dec_lab.hip, MI355X): a compute unit ingests ~45 GB/s whatever the source (HBM or L2) and however
// many waves ask, so a launch takes  floor + max over CUs of (weight bytes + activation bytes that CU reads) / 45 GB/s.  At 128 rows the
// activation matrix (128 x 896 bf16 = 229 KB) is as large as a CU's share of the weights, and the 16-rows-x-64-bytes pieces of a
// row-major fragment load are served at half the rate of a contiguous 1 KiB load.  Hence:
//
//   * activations arrive in FRAGMENT ORDER (hvx_device.h: frag_index), written that way by the producing epilogue: one contiguous 1 KiB
//     load per 16 x 32 fragment;
//   * A-stationary: a workgroup owns 64 rows (blockIdx -> row chunk) and a run of column tiles; its four waves split the ROWS (one 16-row
//     tile each) and keep their activation fragments of the WHOLE K in registers (28 x 4 VGPRs at K = 896), loaded once;
//   * the weights of the workgroup's column tiles — a contiguous byte range of the fragment-packed matrix — go ONCE through an LDS ring by
//     LDS-DMA (buffer_load ... lds, one 1 KiB fragment per wave instruction, D stages in flight), every wave reads every fragment
//     (lane-linear ds_read_b128, conflict-free) and feeds one MFMA with it: no K split inside the workgroup, no cross-wave reduction,
//     every wave runs its own epilogue on its own 16 rows;
//   * counted waits: `s_waitcnt vmcnt(LPS * (D - 2))` + bare `s_barrier` per stage.  The ring reads are inline asm (a compiler-visible LDS
//     read after an LDS-DMA makes hipcc drain vmcnt(0)), and the DMA is the MUBUF form: after a FLAT-encoded global_load_lds hipcc's
//     waitcnt pass treats EVERY later vector-memory wait as vmcnt(0) + lgkmcnt(0) ("pending flat").  Every wave issues the same number of
//     DMA instructions per stage (past the end of its stream they re-read the current tile into a dummy slot), so the count is exact;
//     epilogue loads / stores issued in between only make the wait stronger (memory operations complete in order).
//
// Measured at 128 rows (us per launch, eager back-to-back; the 64-row-chunk skinny / mid forms of gemm_skinny.hip on row-major rows in
// brackets): QKV + RoPE 3.9 [11.1], o_proj + residual 3.9 [10.0], gate/up + SwiGLU 9.1 [15.2], down_proj partials 5.7 [12.8]; as kernel durations
// inside the decode step's graph (rocprofv3): 6.3 [11.1], 5.2 [10.0], 8.4 [15.2], 5.9 [12.8]; heads at 64 sequences x 2: gate / up 33 [48],
// output projection 9.5 [~25] — docs/history/DESIGN_rounds1-4.md §4.1.
#include <stdlib.h>

#include <type_traits>

#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int OFF> __device__ __forceinline__ i32x4 lds_read16(unsigned addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int CNT> __device__ __forceinline__ void lds_wait(i32x4& v) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(CNT)); }
template <int CNT> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory"); }

// 16 e4m3 codes (OCP) -> two bf16 fragments: codes 0..7 = the lane's 8 k of the even k-step, 8..15 = of the odd one.  gfx950's
// v_cvt_scalef32_pk_bf16_fp8 turns two codes into two bf16 in one instruction (scale 1.0: the conversion is exact) — 8 per fragment pair
// instead of 16 with the f32 detour, which would make the vector ALU (not the matrix pipe or the ring) the pace of the stream.
__device__ __forceinline__ void fp8x16_to_bf16(const i32x4& v, bf16x8& lo, bf16x8& hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const bf16x2_t p01 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v[w], 1.0f, false), p23 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v[w], 1.0f, true);
        bf16x8& o = w < 2 ? lo : hi;
        const int e = (w & 1) * 4;
        o[e + 0] = p01[0];
        o[e + 1] = p01[1];
        o[e + 2] = p23[0];
        o[e + 3] = p23[1];
    }
}

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// NTG column tiles per group (SwiGLU: a (gate, up) pair), KT k-steps (the workgroup's whole K range), SF k-steps per ring stage, D stages.
// Grid (1-D, XCD-aware): id -> xcd = id & 7, j = id >> 3, row chunk = j % mch, K split = (j / mch) % split_k, column group =
// (j / mch / split_k) * 8 + xcd — the row chunks of one column group land on the same XCD (same L2: the second chunk finds the weights there);
// with exactly 8 K splits: K split = xcd, column group = j / mch.
// W8: the weights are e4m3 codes (SkinnyArgs.w_fp8): a 1 KiB ring fragment then holds TWO k-steps of a tile (lane: 8 codes of k-step 2j, 8 of 2j + 1;
// packing.pack_frag_fp8) — the ring walks KT / 2 double steps (SF counts those), every fragment is converted to two bf16 operands (e4m3 -> bf16 is
// exact) and feeds two MFMAs in k order, and the per-column power-of-two scale multiplies the accumulator in the epilogue: the products and their
// order are those of the bf16 stream of the dequantised weights, bit for bit.
template <int NTG, int EPI, int ANORM, int KT, int SF, int D, int W8 = 0>
__global__ __launch_bounds__(256, 2) void gemm_dec_kernel(SkinnyArgs a, int n_groups, int gpw, int n_cg, int mch) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass has no __amdgpu_buffer_rsrc_t and drops the stub of a kernel whose body it cannot build)
    static_assert(!W8 || (EPI == SK_SWIGLU && KT % 2 == 0), "fp8 weights: the SwiGLU form, whole double steps");
    constexpr int KTR = W8 ? KT / 2 : KT;                     // ring steps of the workgroup's K range
    constexpr int NST = (KTR + SF - 1) / SF;
    constexpr int SFN = SF * NTG;
    constexpr int LPS = (SFN + 3) / 4;
    static_assert(LPS * (D - 2) <= 63 && D >= 2, "vmcnt range");
    __shared__ __attribute__((aligned(1024))) char ring[(D * SFN + 4) * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int chunk = jj % mch, j2 = jj / mch;
    // eight K slices (down_proj): slice = XCD, so every slice of the activation matrix is fetched into ONE L2 instead of all eight (FETCH_SIZE of the
    // launch 18.7 -> 10.0 MB: the 1.2 MB of MLP rows used to arrive 8 x)
    const int ksplit = a.split_k == 8 ? xcd : j2 % a.split_k, cg = a.split_k == 8 ? j2 : (j2 / a.split_k) * 8 + xcd;
    if (cg >= n_cg) return;
    const int g0 = cg * gpw;
    const int ng = min(n_groups - g0, gpw);
    if (ng <= 0) return;
    const int KTT = a.K >> 5;
    const int KTTR = W8 ? KTT >> 1 : KTT;                     // ring steps of a whole weight row
    const int ks0 = ksplit * KT;
    const int zz = blockIdx.y;                                // stacked launch (the MTP heads' MLP): head z has its own weights, activation rows and output rows
    const int m0 = chunk * 64 + wave * 16;                    // this wave's 16 rows

    // ---- activation fragments of this wave's rows over the workgroup's K range (fragment order: 1 KiB per load) -----------------------------------
    bf16x8 af[KT];
    {
        const int mt = min(m0 >> 4, (a.M - 1) >> 4);          // (a wave past the last row tile re-reads it; none of its rows is ever stored)
        const bf16_t* ap = reinterpret_cast<const bf16_t*>(a.A) + (long long)zz * a.a_zs + ((long long)mt * KTT + ks0) * 512 + lane * 8;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) af[ks] = load8(ap + ks * 512);
    }

    // ---- weight stream: group gi starts at byte ((g0 + gi) * NTG * KTT + ks0) * 1024; within a stage, DMA instruction li of wave w moves
    // fragment q = 4 li + w = (k-step q / NTG, tile q % NTG)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.W) + (long long)zz * a.w_zs * (W8 ? 1 : 2)), 0, a.N * a.K * (W8 ? 1 : 2), 0x00020000);
    const int GS = NTG * KTTR * 1024;
    int wg_cur = (g0 * NTG * KTTR + (W8 ? 0 : ks0)) * 1024;
    int foff[LPS];
#pragma unroll
    for (int li = 0; li < LPS; ++li) {
        const int q = li * 4 + wave;
        foff[li] = ((q % NTG) * KTTR + q / NTG) * 1024;
    }
    char* const dummy = ring + (D * SFN + wave) * 1024;
    // stage (group gi + carry, st) -> ring slot `slot`; past the last group the loads still happen (uniform vmcnt accounting) but re-read
    // the current group into the dummy slot
    auto issue = [&](int gi, auto CARRY, auto ST, int slot) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, carry = decltype(CARRY)::value;
        constexpr int nf = (KTR - st * SF < SF ? KTR - st * SF : SF) * NTG;
        const bool live = gi + carry < ng;
        const int src = wg_cur + (live ? carry * GS : 0) + st * SF * 1024;
        char* const dst = ring + (slot * SFN + wave) * 1024;
#pragma unroll
        for (int li = 0; li < LPS; ++li) {
            bool ok = live;
            if (li * 4 + 3 >= nf) ok = ok && (li * 4 + wave < nf);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(ok ? dst + li * 4096 : dummy), 16, lane * 16, src + (ok ? foff[li] : 0), 0, 0);
        }
    };
    static_for<0, D - 1>([&](auto P) {
        constexpr int p = decltype(P)::value;
        issue(0, std::integral_constant<int, p / NST>{}, std::integral_constant<int, p % NST>{}, p);
    });

    // ---- per-row controls of the QKV epilogue (position, cache slot, active or not).  Unconditional loads on clamped indices, requested BEHIND the
    // operand streams: a load under a branch on another load's value kept the wave from issuing anything else for a round trip at the kernel's start
    int rc_ok[4] = {0, 0, 0, 0}, rc_pos[4] = {0, 0, 0, 0}, rc_slot[4] = {0, 0, 0, 0};
    if constexpr (EPI == SK_QKV_ROPE) {
        int nn[4], lt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = min(m0 + fg * 4 + r, a.M - 1);
            const int si = row / a.kn;
            lt[r] = row - si * a.kn;
            nn[r] = a.n_new[si];
            rc_pos[r] = a.pos0[si] + lt[r];
            rc_slot[r] = a.slot[si];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) rc_ok[r] = (m0 + fg * 4 + r < a.M) && (lt[r] < nn[r]);
    }

    // ---- fused RMSNorm: 1 / rms of this wave's rows from the very fragments the MFMAs consume (the gain is folded into W, llm.py) --------------
    // (also the first use of the activation fragments outside the loop: hipcc then waits for them with a counted vmcnt here instead of
    // flushing vmcnt(0) — DMA prologue included — in the loop preheader)
    float inv[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if constexpr (ANORM) {
        // sum of squares of the 16 rows ON THE MATRIX CORES: frag x frag^T accumulates sum_k x[i][k] x[j][k]; its diagonal (tile row i == tile column
        // i: lane 20 g + r holds row 4 g + r in accumulator element r) is what is wanted.  28 MFMAs on a pipe that has nothing else to do yet,
        // against ~450 VALU instructions per wave for the convert-and-fma form (gate/up 9.5 -> 8.4 us per launch at 128 rows).
        f32x4 ssq = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) mma32(ssq, af[ks], af[ks]);
#pragma unroll
        for (int r = 0; r < 4; ++r) inv[r] = rsqrtf(__shfl(ssq[r], fg * 20 + r, 64) / (float)a.K + a.norm_eps);
    } else {
        asm volatile("" ::"v"(af[KT - 1]));
    }

    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    int slot = 0;
    for (int gi = 0; gi < ng; ++gi) {
        const int tile0 = (g0 + gi) * NTG;
        // ---- epilogue operands of this group, requested before its stages so that waiting for them later does not drain the ring ----------
        float e_bias[NTG], e_res[NTG][4], e_cs[NTG][4], e_sn[NTG][4];
#pragma unroll
        for (int j = 0; j < NTG; ++j) {
            e_bias[j] = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e_res[j][r] = 0.0f;
                e_cs[j][r] = 1.0f;
                e_sn[j][r] = 0.0f;
            }
        }
        if constexpr (W8) {                                  // (the per-column scale rides in the bias slot of the epilogue operands)
            const float* sc = a.w_scale + (long long)zz * a.w_scale_zs;
#pragma unroll
            for (int j = 0; j < NTG; ++j) e_bias[j] = sc[(tile0 + j) * 16 + fr];
        }
        if constexpr (EPI == SK_RESID || EPI == SK_QKV_ROPE || EPI == SK_STORE) {
            if (a.bias) {
#pragma unroll
                for (int j = 0; j < NTG; ++j) e_bias[j] = a.bias[(tile0 + j) * 16 + fr];
            }
        }
        if constexpr (EPI == SK_RESID) {
            const float* xo = reinterpret_cast<const float*>(a.out);
#pragma unroll
            for (int j = 0; j < NTG; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + fg * 4 + r;
                    if (row < a.M) e_res[j][r] = xo[(long long)row * a.ldo + (tile0 + j) * 16 + fr];
                }
        }
        if constexpr (EPI == SK_QKV_ROPE) {
#pragma unroll
            for (int j = 0; j < NTG; ++j) {
                const int head = (tile0 + j) >> 2, t = (tile0 + j) & 3;
                if (head < a.q_heads + a.kv_heads) {
                    const int f = 8 * t + (fr & 7);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const long long pr = rc_ok[r] ? rc_pos[r] : 0;               // (an inactive row's position may lie past the tables)
                        e_cs[j][r] = a.rope_cos[pr * 32 + f];
                        e_sn[j][r] = a.rope_sin[pr * 32 + f];
                    }
                }
            }
        }

        f32x4 acc[NTG];
#pragma unroll
        for (int j = 0; j < NTG; ++j) acc[j] = f32x4{0, 0, 0, 0};
        static_for<0, NST>([&](auto ST) {
            constexpr int st = decltype(ST)::value;
            constexpr int nf = (KTR - st * SF < SF ? KTR - st * SF : SF) * NTG;
            vm_wait<LPS*(D - 2)>();                             // this wave's DMA of the stage has landed ...
            __builtin_amdgcn_s_barrier();                      // ... and so has everybody's; everybody has read the previous stage
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
                if constexpr (W8) {
                    bf16x8 w0, w1;
                    fp8x16_to_bf16(b[q], w0, w1);
                    mma32(acc[q % NTG], af[2 * (st * SF + q / NTG)], w0);
                    mma32(acc[q % NTG], af[2 * (st * SF + q / NTG) + 1], w1);
                } else {
                    mma32(acc[q % NTG], af[st * SF + q / NTG], __builtin_bit_cast(bf16x8, b[q]));
                }
            });
            slot = slot + 1 == D ? 0 : slot + 1;
        });
        wg_cur += GS;

        // ---- epilogues (C layout: column = fr, rows = fg * 4 + r) ---------------------------------------------------------------------
        if constexpr (EPI == SK_PARTIAL) {
            float* part = a.part + (long long)ksplit * a.M * a.N;
#pragma unroll
            for (int j = 0; j < NTG; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + fg * 4 + r;
                    if (row < a.M) part[(long long)row * a.N + (tile0 + j) * 16 + fr] = acc[j][r];
                }
        } else if constexpr (EPI == SK_STORE) {
            // fp32 rows of the stacked (head z, sequence s) grid, row = z * kn + s, stored where the per-head launches put them: out[s][z][col]
            float* out = reinterpret_cast<float*>(a.out);
            const int nv = a.n_valid > 0 ? a.n_valid : a.N;
#pragma unroll
            for (int j = 0; j < NTG; ++j) {
                const int col = (tile0 + j) * 16 + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + fg * 4 + r;
                    if (row >= a.M || col >= nv) continue;
                    const int z = row / a.kn, si = row - z * a.kn;
                    out[(long long)si * a.ldo + (long long)z * a.out_zs + col] = acc[j][r] + e_bias[j];
                }
            }
        } else if constexpr (EPI == SK_RESID) {
            // residual stream update in place: x[row][col] += acc + bias, one writer per element; the 16-bit copy feeds the next fused-norm GEMM
            float* xo = reinterpret_cast<float*>(a.out);
            bf16_t* xc = reinterpret_cast<bf16_t*>(a.out2);
            const int KTo = a.N >> 5;
#pragma unroll
            for (int j = 0; j < NTG; ++j) {
                const int col = (tile0 + j) * 16 + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + fg * 4 + r;
                    if (row >= a.M) continue;
                    const float x1 = e_res[j][r] + (acc[j][r] + e_bias[j]);
                    xo[(long long)row * a.ldo + col] = x1;
                    if (xc) xc[a.out_frag ? frag_index(row, col, KTo) : (long long)row * a.ldo2 + col] = f32_to_bf16(x1);
                }
            }
        } else if constexpr (EPI == SK_SWIGLU) {
            static_assert(EPI != SK_SWIGLU || NTG == 2, "SwiGLU works on (gate, up) tile pairs");
            bf16_t* out = reinterpret_cast<bf16_t*>(a.out) + (long long)zz * a.out_zs;
            const int col = (tile0 >> 1) * 16 + fr;
            const int KTo = a.N >> 6;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + fg * 4 + r;
                if (row >= a.M) continue;
                float gte = acc[0][r] * inv[r], up = acc[NTG - 1][r] * inv[r];
                if constexpr (W8) {
                    gte *= e_bias[0];
                    up *= e_bias[NTG - 1];
                }
                const float v = gte * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gte)) * up;   // silu(gate) * up
                out[a.out_frag ? frag_index(row, col, KTo) : (long long)row * a.ldo + col] = f32_to_bf16(v);
            }
        } else {   // SK_QKV_ROPE: tile t of a head holds d = 8t..8t+7 (lanes fr < 8) and their rotate-half partners d + 32 (packing.qkv_row_perm)
#pragma unroll
            for (int j = 0; j < NTG; ++j) {
                const int head = (tile0 + j) >> 2, t = (tile0 + j) & 3;
                const int which = head < a.q_heads ? 0 : (head < a.q_heads + a.kv_heads ? 1 : 2);
                const int hh = which == 0 ? head : (which == 1 ? head - a.q_heads : head - a.q_heads - a.kv_heads);
                const int d = (fr < 8) ? (8 * t + fr) : (32 + 8 * t + (fr - 8));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + fg * 4 + r;
                    float x = acc[j][r] * inv[r] + e_bias[j];
                    const float partner = __shfl_xor(x, 8, 64);          // same row, the other half of the head
                    if (which < 2) x = (fr < 8) ? (x * e_cs[j][r] - partner * e_sn[j][r]) : (x * e_cs[j][r] + partner * e_sn[j][r]);
                    if (!rc_ok[r]) continue;
                    if (which == 0) {
                        reinterpret_cast<bf16_t*>(a.qbuf)[((long long)row * a.q_heads + hh) * 64 + d] = f32_to_bf16(x);
                    } else if (which == 1) {
                        const long long blk = ((long long)rc_slot[r] * a.kv_heads + hh) * a.max_ctx * 64;
                        reinterpret_cast<bf16_t*>(a.kcache)[blk + (a.kv_frag ? frag_index(rc_pos[r], d, 2) : (long long)rc_pos[r] * 64 + d)] = f32_to_bf16(x);
                    } else {
                        const long long blk = ((long long)rc_slot[r] * a.kv_heads + hh) * a.max_ctx * 64;
                        reinterpret_cast<bf16_t*>(a.vTcache)[blk + (a.kv_frag ? vfrag_index(rc_pos[r], d) : (long long)d * a.max_ctx + rc_pos[r])] = f32_to_bf16(x);
                    }
                }
            }
        }
    }
#endif
}

template <int NTG, int EPI, int ANORM, int KT, int SF, int D, int W8 = 0>
int launch_form(const SkinnyArgs& a, int gpw, hipStream_t s) {
    const int n_groups = a.N / 16 / NTG;
    const int n_cg = (n_groups + gpw - 1) / gpw;
    const int mch = (a.M + 63) / 64;
    const int grid = a.split_k == 8 ? 8 * mch * n_cg : 8 * mch * a.split_k * ((n_cg + 7) / 8);
    const double bytes = (double)a.N * a.K * (W8 ? 1 : 2) + (double)a.M * a.K * 2 + (double)a.M * a.N * (EPI == SK_PARTIAL ? 4.0 * a.split_k : 2.0);
    const int slot = prof_begin(PK_SKINNY, bytes, s);
    hipLaunchKernelGGL((gemm_dec_kernel<NTG, EPI, ANORM, KT, SF, D, W8>), dim3(grid, a.nz > 1 ? a.nz : 1), dim3(256), 0, s, a, n_groups, gpw, n_cg, mch);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 1 : (set_error("decode gemm launch failed"), -1);
}

}  // namespace

// The instantiations cover the Qwen2-0.5B backbone of HydraVox-CV3 (hidden 896 = 28 k-steps, intermediate 4864 = 8 x 19 k-steps); any other
// shape stays on the generic kernels of gemm_skinny.hip.
bool dec_gemm_shape_ok(int M, int N, int K, int epi, int split_k) {
    const bool off = opt(OPT_DEC_GEMM) == 0;          // (option dec_gemm: A / B switch for tools/bench_decode.py and the parity tests)
    // (rows come in chunks of 64 per workgroup: 33..128 rows are the two-chunk geometry the lab tuned; up to 256 rows — 128 sequences x 2 heads — the
    // weights are read once per chunk from the XCD's L2)
    if (off || M <= 32 || M > 256 || (N & 15) || (K & 31)) return false;
    const int kt = K / 32;
    switch (epi) {
        case SK_QKV_ROPE: return kt == 28 && split_k == 1;
        case SK_RESID: return kt == 28 && split_k == 1;
        case SK_STORE: return kt == 28 && split_k == 1;
        case SK_SWIGLU: return kt == 28 && split_k == 1 && (N & 31) == 0;
        case SK_PARTIAL: return split_k > 1 && kt == split_k * 19 && (N & 31) == 0;
    }
    return false;
}

int launch_dec_gemm(const SkinnyArgs& a_in, hipStream_t s) {
    SkinnyArgs a = a_in;
    if (a.epi == SK_STORE && a.nz >= 1 && a.w_zs == 0 && a.out_f32 && (!a.bias || a.bias_zs == 0)) {
        // the shared output projection over the K heads' last rows (hvx_llm.hip): the per-head row blocks [z][s] are ONE stacked activation
        // matrix in fragment order; rows go back to out[s][z][:] in the epilogue
        a.kn = a.M;
        a.M = a.M * a.nz;
        a.nz = 1;
    }
    // (nz > 1: only the SwiGLU form carries the per-head strides — the MTP heads' MLP, every head its own fragment-order matrix of a.M rows)
    if (a.dtype != DT_BF16 || !a.a_frag || (a.nz > 1 && a.epi != SK_SWIGLU) || a.w_narrow || !dec_gemm_shape_ok(a.M, a.N, a.K, a.epi, a.split_k)) return 0;
    if (a.epi != SK_PARTIAL && a.split_k != 1) return 0;
    if ((long long)a.N * a.K * 2 >= (1LL << 31)) return 0;            // (32-bit buffer offsets)
    if (a.w_fp8 && a.epi != SK_SWIGLU) return set_error("launch_dec_gemm: fp8 weights on epilogue %d", a.epi), -1;
    // column groups per workgroup (tools/dec_lab.hip, 128 rows): one tile for the two narrow projections (72 / 56 tiles x 2 row chunks), three
    // (gate, up) pairs for the MLP (102 x 2 workgroups: the activation re-reads of more, smaller workgroups cost more than the idle CUs), two
    // pairs of tiles per K slice for the down projection (14 x 8 x 2)
    const int gpw_qkv = (int)opt(OPT_DEC_GPW_QKV), gpw_res = (int)opt(OPT_DEC_GPW_RES), gpw_mlp = (int)opt(OPT_DEC_GPW_MLP),
              gpw_down = (int)opt(OPT_DEC_GPW_DOWN), gpw_out = (int)opt(OPT_DEC_GPW_OUT), gpw_hmlp = (int)opt(OPT_DEC_GPW_HMLP);       // (lab options: the defaults are the tuned values)
    switch (a.epi) {
        case SK_QKV_ROPE:
            if (a.N != (a.q_heads + 2 * a.kv_heads) * 64) return set_error("launch_dec_gemm: QKV width %d != (q+2kv)*64", a.N), -1;
            return a.a_norm ? launch_form<1, SK_QKV_ROPE, 1, 28, 4, 7>(a, gpw_qkv, s) : launch_form<1, SK_QKV_ROPE, 0, 28, 4, 7>(a, gpw_qkv, s);
        case SK_RESID: return launch_form<1, SK_RESID, 0, 28, 4, 7>(a, gpw_res, s);
        case SK_STORE: return launch_form<1, SK_STORE, 0, 28, 4, 7>(a, gpw_out, s);      // (423 vocabulary tiles: 141 x 2 workgroups)
        case SK_SWIGLU:
            // (the heads' MLP is 1376 (gate, up) pairs per head: 11 per workgroup = 126 workgroups x K heads, 616 KB of weights behind one 115 KB activation chunk)
            // (deeper rings — 6, 8, 10 stages — and 3 .. 22 pairs per workgroup measured within 1 % of this: the launch runs at ~4.8 TB/s, 33 us for 158 MB)
            if (a.w_fp8) {
                if (a.a_norm || !a.w_scale) return set_error("launch_dec_gemm: fp8 weights serve the un-normalised SwiGLU form with per-column scales"), -1;
                return launch_form<2, SK_SWIGLU, 0, 28, 7, 4, 1>(a, gpw_hmlp, s);          // (7 double steps x 2 tiles per stage: the bf16 form's 14 KB)
            }
            if (a.nz > 1 || a.N > 16384) return a.a_norm ? launch_form<2, SK_SWIGLU, 1, 28, 7, 4>(a, gpw_hmlp, s) : launch_form<2, SK_SWIGLU, 0, 28, 7, 4>(a, gpw_hmlp, s);
            return a.a_norm ? launch_form<2, SK_SWIGLU, 1, 28, 7, 4>(a, gpw_mlp, s) : launch_form<2, SK_SWIGLU, 0, 28, 7, 4>(a, gpw_mlp, s);
        case SK_PARTIAL: return launch_form<2, SK_PARTIAL, 0, 19, 10, 3>(a, gpw_down, s);
    }
    return 0;
}

}  // namespace hvx
