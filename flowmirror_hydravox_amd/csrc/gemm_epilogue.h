// gemm_epilogue.h — the fused epilogues of the implicit-GEMM kernels (gemm_tiled.hip, gemm_big.hip): see hvx_kernels.h: GemmArgs.
#pragma once
#include <type_traits>

#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

// a 64-byte row of zeros: where the LDS-DMA form needs a padding / out-of-range row it points the lane's source address here
static __device__ __attribute__((aligned(64))) const float g_zero_row[16] = {0};

// ---- epilogue (shared by the tile forms) -------------------------------------------------------------------------------------------
// Ordering of a wave's OWN LDS traffic (the staging tile is private to the wave): DS instructions of one wave execute in issue order, so a
// ds_write followed by a ds_read of the same wave needs no hardware fence — only the compiler must keep the program order.  A
// __builtin_amdgcn_fence(release, "wavefront") here lowers to s_waitcnt vmcnt(0): every staging pass then waited for the previous pass's
// global STORES to be acknowledged (~2 us under load, 8 passes per tile).
__device__ __forceinline__ void wave_lds_order() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// fp32 values leaving as a (hi, lo) bf16 plane pair (GemmArgs.out_planes / out2_planes): four consecutive columns = two 8-byte stores
__device__ __forceinline__ void store_planes4(bf16_t* hi, long long plane, const f32x4& v) {
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = f32_to_bf16(v[e]);
        l[e] = f32_to_bf16(v[e] - bf16_to_f32(h[e]));
    }
    *reinterpret_cast<bf16x4*>(hi) = h;
    *reinterpret_cast<bf16x4*>(hi + plane) = l;
}
__device__ __forceinline__ void store_planes1(bf16_t* hi, long long plane, float v) {
    const bf16_t h = f32_to_bf16(v);
    hi[0] = h;
    hi[plane] = f32_to_bf16(v - bf16_to_f32(h));
}

// acc: the wave's MT x NT accumulator tiles whose first row / column are mw0 / nw0; scr: this wave's private fp32 LDS staging tile of
// ((64 / WN) * 16) rows x (WN + 4) floats (the caller has passed the barrier that frees it).
template <bool RT, int ACT, int RES, int OF32, int O2, int ACT2 = 0> struct LeanMode {
    static constexpr bool rt = RT;
    static constexpr int act = ACT, res = RES, of32 = OF32, o2 = O2, act2 = ACT2;
};

// LEAN: compile the streamlined pass for whole, aligned column tiles (costs ~50 VGPRs: only the one-workgroup-per-CU 256-tile form takes it)
// QT: the storage type of q / k / V^T in EPI_QKV_DIT (the attention's operand type; differs from T only for fp16-operand Linears, whose attention stays bf16)
// AT: the accumulator array — f32x4 [MT][NT] (16 x 16 x 32 MFMA tiles: lane (fr, fg) holds rows 4 fg .. 4 fg + 3 of column fr) or f32x16 [MT / 2][NT / 2]
// (32 x 32 x 16 tiles: lane (c = lane & 31, hi = lane >> 5), register r holds row (r & 3) + 8 (r >> 2) + 4 hi of column c).  The layout matters in ONE place, the
// staging of 16-row passes into the wave's LDS tile (acc_stage below); everything behind it reads whole rows from there.
typedef float f32x16_epi __attribute__((ext_vector_type(16)));
template <int MT2, int NT2, int NII, int ip>
__device__ __forceinline__ void acc_stage(const f32x16_epi (&acc)[MT2][NT2], float* scr, int sl, int lane) {
    const int c = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ii = 0; ii < NII; ++ii) {
        const int t = ip + ii;                                   // 16-row unit of the wave tile: the (t & 1)-th half of 32-row tile t / 2
#pragma unroll
        for (int j = 0; j < NT2; ++j)
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) scr[(ii * 16 + (r8 & 3) + 8 * (r8 >> 2) + 4 * hi) * sl + j * 32 + c] = acc[t >> 1][j][8 * (t & 1) + r8];
    }
}

template <class T, int MT, int NT, int WN, int EPI, int LEAN = 0, class QT = T, class AT = f32x4[MT][NT]>       // LEAN 1: DiT Linears (no per-column activation tables); 2: + Snake tables (vocoder)
__device__ __forceinline__ void gemm_epilogue(const GemmArgs a, AT& acc, float* scr, int lane, int mw0, int nw0, int bz, int g) {
    constexpr int SLD = WN + 4;                      // staging row stride (floats)
    constexpr int ROWS_PASS = (64 / WN) * 16;        // rows a wave stages per pass: 64 lanes x 16 columns each
    const int fr = lane & 15, fg = lane >> 4;
    // ---- epilogue ----------------------------------------------------------------------------------
    // The MFMA C layout gives a lane 4 rows x 1 column per tile: stored directly that is 2-4 byte scatters.  Each wave
    // therefore transposes its accumulators through a private fp32 LDS tile and every lane finishes 16 CONSECUTIVE columns
    // of one row: bias / gate / residual come in as 16-byte loads and the result leaves as 32-64 contiguous bytes per lane
    // (a full 128-256 B row segment per 4 lanes).
    constexpr int MT_PASS = ROWS_PASS / 16;
    static_assert(MT % MT_PASS == 0, "wave tile must be a whole number of staging passes");
    auto stage = [&](auto IP) __attribute__((always_inline)) {
        constexpr int ip = decltype(IP)::value;                // compile-time: a runtime index would push acc[][] to scratch
        if constexpr (std::is_same<AT, f32x4[MT][NT]>::value) {
#pragma unroll
            for (int ii = 0; ii < MT_PASS; ++ii)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) scr[(ii * 16 + fg * 4 + r) * SLD + j * 16 + fr] = acc[ip + ii][j][r];
        } else acc_stage<MT / 2, NT / 2, MT_PASS, ip>(acc, scr, SLD, lane);
        wave_lds_order();
    };
    auto unstage = [&]() __attribute__((always_inline)) {
        wave_lds_order();
    };

    if constexpr (EPI == EPI_GENERIC) {
        constexpr int LPR = WN / 16;                     // lanes per staged row
        const int prow = lane / LPR, pcs = (lane % LPR) * 16;
        const long long ob = (long long)bz * a.out_bs;
        const int col0 = nw0 + pcs;
        const int gc0 = g * a.N + col0;
        const bool full = (col0 + 16) <= a.N;
        // 16-byte vector access is legal when every leading dimension / base keeps 4-float (8-bf16) alignment
        auto al = [](const void* p, long long ld, int esz) { return p == nullptr || ((((unsigned long long)p) & 15) == 0 && ((ld * esz) & 15) == 0); };
        const bool vec = al(a.bias, 0, 4) && al(a.act_alpha, 0, 4) && al(a.act2_alpha, 0, 4) && al(a.gate, a.gate_bs, 4) &&
                         al(a.res, a.ldres, a.res_f16 ? 2 : 4) && ((a.res_bs * (a.res_f16 ? 2 : 4)) & 15) == 0 && al(a.res2, a.ldres2, 4) && ((a.res2_bs * 4) & 15) == 0 &&
                         al(a.out, a.ldo, a.out_f16 ? 2 : a.out_f32 ? 4 : (int)sizeof(T)) && ((a.out_bs * (a.out_f16 ? 2 : a.out_f32 ? 4 : (int)sizeof(T))) & 15) == 0 &&
                         al(a.out2, a.ldo2, (int)sizeof(T)) && ((a.out2_bs * (int)sizeof(T)) & 15) == 0 && ((g * a.N) & 7) == 0;
        // ---- streamlined form: whole column tiles, 16-byte alignment everywhere, no per-column activation tables, one residual --------
        // (every DiT Linear).  The column operands (bias, gate) do not depend on the pass and are loaded once; the residual rows of pass
        // p + 1 are requested before pass p is finished, so no pass waits on a cold global load; bf16 results leave as 16-byte stores.
        constexpr bool ALPHA = LEAN == 2;
        const bool half_io = a.res_f16 || a.out_f16;
        const bool lean = LEAN && (!half_io || (LEAN == 1 && WN == 64 && MT_PASS == 1 && sizeof(T) == 2 && a.res && a.res_f16 && a.out_f16 && a.out && !a.out2 && a.scale == 1.0f && a.act == ACT_NONE)) && vec && (nw0 + WN) <= a.N && (ALPHA || (!a.act_alpha && !a.act2_alpha)) && a.act != ACT_SNAKEBETA && !a.res2 &&
                          a.div == 0.0f && a.out_row_off >= 0 && a.out2_row_off >= 0 && a.res_row_off >= 0;
        if constexpr (LEAN != 0) if (lean) {
            // Lane -> (row, column) maps of a pass (PERM), chosen per mode so that every global instruction of the wave covers WHOLE contiguous
            // row segments: with "16 consecutive columns per lane" (PERM 0) a 16-byte access of the wave is a comb — 16 B used of every 64 B
            // (fp32) or 32 B (bf16) — and every 128-byte line of the tile is touched by 2-4 instructions.
            //   PERM 1 (fp32 in / out, 4 lanes per row): instruction q covers columns [16 q, 16 q + 16) of 16 rows = 64 contiguous bytes per row
            //   PERM 2 (bf16 out, 8 lanes per row):      store u covers all 64 columns of rows [8 u, 8 u + 8) = 128 contiguous bytes per row
            //   PERM 3 (fp32 in / out, 8 lanes per row): instruction q covers columns [32 (q & 1), + 32) of rows [8 (q >> 1), + 8) = 128 contiguous
            //           bytes per row; a second bf16 output leaves as 8-byte stores (64 contiguous bytes per row and instruction)
            const int gcw = g * a.N + nw0;
            const float* const resw = (a.res && !a.res_f16) ? a.res + (long long)bz * a.res_bs + gcw : nullptr;
            const f16_t* const resh = (a.res && a.res_f16) ? reinterpret_cast<const f16_t*>(a.res) + (long long)bz * a.res_bs + gcw : nullptr;
            auto lean_pass = [&](auto IP, auto MODE, auto PERMC, f32x4 (&bi)[4], f32x4 (&gt)[4], f32x4 (&al)[4], f32x4 (&al2)[4], f32x4 (&rs)[4])
                                 __attribute__((always_inline)) {
                constexpr int ip = decltype(IP)::value;
                constexpr int PERM = decltype(PERMC)::value;
                typedef decltype(MODE) MD;
                auto rsel = [&](int q) { return PERM >= 2 ? (lane >> 3) + 8 * (q >> 1) : prow; };
                auto cof = [&](int q) {
                    return PERM == 3 ? (q & 1) * 32 + (lane & 7) * 4 : PERM == 2 ? (lane & 7) * 8 + (q & 1) * 4 : PERM == 1 ? q * 16 + (lane & 3) * 4 : pcs + q * 4;
                };
                // bf16 results: 16-byte stores of two adjacent column groups (PERM 0 / 2) or one 8-byte store per group (PERM 1 / 3)
                auto store_bf16 = [&](auto* base, long long ld, int row_off, const f32x4 (&v)[4]) __attribute__((always_inline)) {
                    if constexpr (PERM == 0 || PERM == 2) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            typename Vec8<T>::type w8;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { w8[e] = from_f32<T>(v[2 * u][e]); w8[4 + e] = from_f32<T>(v[2 * u + 1][e]); }
                            if (mw0 + ip * 16 + rsel(2 * u) < a.M) store8(base + (long long)(mw0 + ip * 16 + rsel(2 * u) + row_off) * ld + cof(2 * u), w8);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const typename Vec4<T>::type w4 = {from_f32<T>(v[q][0]), from_f32<T>(v[q][1]), from_f32<T>(v[q][2]), from_f32<T>(v[q][3])};
                            if (mw0 + ip * 16 + rsel(q) < a.M) *reinterpret_cast<typename Vec4<T>::type*>(base + (long long)(mw0 + ip * 16 + rsel(q) + row_off) * ld + cof(q)) = w4;
                        }
                    }
                };
                const int act = MD::rt ? a.act : MD::act;
                const bool has_res = MD::rt ? (resw != nullptr) : (MD::res != 0);
                const bool out_f32 = MD::rt ? (a.out_f32 != 0) : (bool)MD::of32;
                const bool has_out = MD::rt ? (a.out != nullptr) : true;
                const bool has_out2 = MD::rt ? (a.out2 != nullptr) : (bool)MD::o2;
                auto res_load = [&](int row0, f32x4 (&r4)[4]) __attribute__((always_inline)) {
                    if constexpr (MD::res == 2) {              // fp16 stream (PERM 2: 8 consecutive columns of rows rsel(0) and rsel(2)): raw bits until use
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int row = row0 + rsel(2 * u);
                            const int rr = (row < a.M ? row : a.M - 1) + a.res_row_off;
                            r4[u] = *reinterpret_cast<const f32x4*>(resh + (long long)rr * a.ldres + cof(2 * u));
                        }
                        return;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = row0 + rsel(q);
                        const int rr = (row < a.M ? row : a.M - 1) + a.res_row_off;       // rows past the end are never stored: any valid row serves
                        r4[q] = *reinterpret_cast<const f32x4*>(resw + (long long)rr * a.ldres + cof(q));
                    }
                };
                if constexpr (ip == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if constexpr (ALPHA) {
                            al[q] = a.act_alpha ? *reinterpret_cast<const f32x4*>(a.act_alpha + gcw + cof(q)) : f32x4{1, 1, 1, 1};
                            al2[q] = a.act2_alpha ? *reinterpret_cast<const f32x4*>(a.act2_alpha + gcw + cof(q)) : f32x4{1, 1, 1, 1};
                        }
                        bi[q] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + gcw + cof(q)) : f32x4{0, 0, 0, 0};
                        gt[q] = a.gate ? *reinterpret_cast<const f32x4*>(a.gate + (long long)bz * a.gate_bs + gcw + cof(q)) : f32x4{1, 1, 1, 1};
                    }
                    if (has_res) res_load(mw0, rs);
                }
                f32x4 rn[4];
                if (has_res && ip + MT_PASS < MT) res_load(mw0 + (ip + MT_PASS) * 16, rn);
                stage(IP);
                f32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(&scr[rsel(q) * SLD + cof(q)]);
                unstage();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] += bi[q];
                    if (act != ACT_NONE) {
                        const f32x4 alq = ALPHA ? al[q] : f32x4{1, 1, 1, 1};
                        if constexpr (!MD::rt && (MD::act == ACT_GELU_TANH || MD::act == ACT_SNAKE)) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[q][e] = act_apply(MD::act, v[q][e], 0.0f, alq[e], 1.0f);
                        } else {
                            v[q] = act_apply4(act, v[q], a.act_param, alq, f32x4{1, 1, 1, 1});
                        }
                    }
                    v[q] *= gt[q];
                    if constexpr (MD::res == 2) {
                        const f16x8 h8 = __builtin_bit_cast(f16x8, rs[q >> 1]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[q][e] += (float)h8[(q & 1) * 4 + e];
                    } else {
                        if (has_res) v[q] += rs[q];
                    }
                    v[q] *= a.scale;
                }
                const int row0 = mw0 + ip * 16;
                if constexpr (MD::of32 == 2) {               // fp16 stream out (PERM 2): 16-byte stores, 128 contiguous bytes per row and instruction
                    f16_t* const obh = reinterpret_cast<f16_t*>(a.out) + ob + gcw;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        f16x8 w8;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { w8[e] = f32_to_f16_sat(v[2 * u][e]); w8[4 + e] = f32_to_f16_sat(v[2 * u + 1][e]); }
                        if (row0 + rsel(2 * u) < a.M) *reinterpret_cast<f16x8*>(obh + (long long)(row0 + rsel(2 * u) + a.out_row_off) * a.ldo + cof(2 * u)) = w8;
                    }
                } else if (has_out) {
                    if (sizeof(T) != 2 && a.out_planes) {
                        bf16_t* const obp = reinterpret_cast<bf16_t*>(a.out) + ob + gcw;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (row0 + rsel(q) < a.M) store_planes4(obp + (long long)(row0 + rsel(q) + a.out_row_off) * a.ldo + cof(q), a.out_plane, v[q]);
                    } else if (out_f32 || sizeof(T) != 2) {
                        float* const ob32 = reinterpret_cast<float*>(a.out) + ob + gcw;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (row0 + rsel(q) < a.M) *reinterpret_cast<f32x4*>(ob32 + (long long)(row0 + rsel(q) + a.out_row_off) * a.ldo + cof(q)) = v[q];
                    } else if constexpr (sizeof(T) == 2) {
                        store_bf16(reinterpret_cast<T*>(a.out) + ob + gcw, a.ldo, a.out_row_off, v);
                    }
                }
                if (has_out2) {
                    T* const ob2 = reinterpret_cast<T*>(a.out2) + (long long)bz * a.out2_bs + gcw;
                    const int act2 = MD::rt ? a.act2 : MD::act2;
                    if (act2 != ACT_NONE) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 alq = ALPHA ? al2[q] : f32x4{1, 1, 1, 1};
                            if constexpr (!MD::rt && MD::act2 == ACT_SNAKE) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[q][e] = act_apply(ACT_SNAKE, v[q][e], 0.0f, alq[e], 1.0f);
                            } else {
                                v[q] = act_apply4(act2, v[q], a.act2_param, alq, f32x4{1, 1, 1, 1});
                            }
                        }
                    }
                    if constexpr (sizeof(T) == 2) {
                        store_bf16(ob2, a.ldo2, a.out2_row_off, v);
                    } else if (a.out2_planes) {
                        bf16_t* const obp = reinterpret_cast<bf16_t*>(a.out2) + (long long)bz * a.out2_bs + gcw;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (row0 + rsel(q) < a.M) store_planes4(obp + (long long)(row0 + rsel(q) + a.out2_row_off) * a.ldo2 + cof(q), a.out2_plane, v[q]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (row0 + rsel(q) < a.M)
                                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ob2) + (long long)(row0 + rsel(q) + a.out2_row_off) * a.ldo2 + cof(q)) = v[q];
                    }
                }
                if (has_res && ip + MT_PASS < MT) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) rs[q] = rn[q];
                }
            };
            auto lean_all = [&](auto MODE) __attribute__((always_inline)) {
                typedef decltype(MODE) MD;
                // the permuted maps assume 4 lanes per 64-column row and one 16-row tile per pass (the 256-tile form and the vocoder form)
                constexpr int PERM = (MD::rt || WN != 64 || MT_PASS != 1) ? 0 : MD::of32 == 2 ? 2 : MD::o2 ? 3 : (MD::of32 || sizeof(T) != 2) ? 1 : 2;   // (PERM 3 for the plain fp32 modes measured 2-3 % slower than PERM 1)
                std::integral_constant<int, PERM> pc;
                f32x4 bi[4], gt[4], al[4], al2[4], rs[4];
                lean_pass(std::integral_constant<int, 0>{}, MODE, pc, bi, gt, al, al2, rs);
                if constexpr (MT > 1 * MT_PASS) lean_pass(std::integral_constant<int, 1 * MT_PASS>{}, MODE, pc, bi, gt, al, al2, rs);
                if constexpr (MT > 2 * MT_PASS) lean_pass(std::integral_constant<int, 2 * MT_PASS>{}, MODE, pc, bi, gt, al, al2, rs);
                if constexpr (MT > 3 * MT_PASS) lean_pass(std::integral_constant<int, 3 * MT_PASS>{}, MODE, pc, bi, gt, al, al2, rs);
                if constexpr (MT > 4 * MT_PASS) lean_pass(std::integral_constant<int, 4 * MT_PASS>{}, MODE, pc, bi, gt, al, al2, rs);
                if constexpr (MT > 5 * MT_PASS) lean_pass(std::integral_constant<int, 5 * MT_PASS>{}, MODE, pc, bi, gt, al, al2, rs);
                if constexpr (MT > 6 * MT_PASS) lean_pass(std::integral_constant<int, 6 * MT_PASS>{}, MODE, pc, bi, gt, al, al2, rs);
                if constexpr (MT > 7 * MT_PASS) lean_pass(std::integral_constant<int, 7 * MT_PASS>{}, MODE, pc, bi, gt, al, al2, rs);
            };
            const bool simple = a.out && !a.out2 && a.scale == 1.0f;
            constexpr bool HALF_OK = LEAN == 1 && WN == 64 && MT_PASS == 1 && sizeof(T) == 2;     // (the fp16 stream exists in the DiT's bf16 Linears only)
            if constexpr (HALF_OK) {
                if (simple && a.act == ACT_NONE && resh && a.out_f16) { lean_all(LeanMode<false, ACT_NONE, 2, 2, 0>{}); return; }   // out_proj, FF2 on the half stream
            }
            if (simple && a.act == ACT_NONE && resw && a.out_f32) lean_all(LeanMode<false, ACT_NONE, 1, 1, 0>{});              // out_proj, FF2: x += gate * (.)
            else if (simple && a.act == ACT_GELU_TANH && !resw && !a.out_f32) lean_all(LeanMode<false, ACT_GELU_TANH, 0, 0, 0>{});   // FF1
            else if (simple && a.act == ACT_NONE && !resw && !a.out_f32) lean_all(LeanMode<false, ACT_NONE, 0, 0, 0>{});           // plain store
            else if (ALPHA && a.out && !a.out2 && a.scale == 1.0f && a.act == ACT_SNAKE && !resw && a.out_f32 && !a.gate) lean_all(LeanMode<false, ACT_SNAKE, 0, 1, 0>{});   // ResBlock convs1
            else if (ALPHA && a.out && a.out2 && a.scale == 1.0f && a.act == ACT_NONE && resw && a.out_f32 && a.act2 == ACT_SNAKE && !a.gate)
                lean_all(LeanMode<false, ACT_NONE, 1, 1, 1, ACT_SNAKE>{});                                                                                                 // ResBlock convs2 + next Snake
            else lean_all(LeanMode<true, 0, 0, 0, 0>{});
            return;
        }
        auto do_pass = [&](auto IP) __attribute__((always_inline)) {
            constexpr int ip = decltype(IP)::value;
            stage(IP);
            float x[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&scr[prow * SLD + pcs + q * 4]);
                x[q * 4 + 0] = t[0]; x[q * 4 + 1] = t[1]; x[q * 4 + 2] = t[2]; x[q * 4 + 3] = t[3];
            }
            unstage();
            const int row = mw0 + ip * 16 + prow;
            if (row >= a.M || (row + a.out_row_off) < 0) return;
            if (full) {
                // four self-contained 4-column chunks (keeps the live register set small: the accumulators own most of the file)
                const float* resp = (a.res && !a.res_f16) ? a.res + (long long)bz * a.res_bs + (long long)(row + a.res_row_off) * a.ldres + gc0 : nullptr;
                const f16_t* resph = (a.res && a.res_f16) ? reinterpret_cast<const f16_t*>(a.res) + (long long)bz * a.res_bs + (long long)(row + a.res_row_off) * a.ldres + gc0 : nullptr;
                const float* res2p = a.res2 ? a.res2 + (long long)bz * a.res2_bs + (long long)row * a.ldres2 + gc0 : nullptr;
                const long long o1 = ob + (long long)(row + a.out_row_off) * a.ldo + gc0;
                const long long o2 = (long long)bz * a.out2_bs + (long long)(row + a.out2_row_off) * a.ldo2 + gc0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c4 = q * 4;
                    auto ld4 = [&](const float* p) -> f32x4 {
                        if (vec) return *reinterpret_cast<const f32x4*>(p);
                        return f32x4{p[0], p[1], p[2], p[3]};
                    };
                    f32x4 v = {x[c4], x[c4 + 1], x[c4 + 2], x[c4 + 3]};
                    if (a.bias) v += ld4(a.bias + gc0 + c4);
                    if (a.act != ACT_NONE) {
                        f32x4 al = {1.0f, 1.0f, 1.0f, 1.0f}, be = {1.0f, 1.0f, 1.0f, 1.0f};
                        if (a.act_alpha) al = ld4(a.act_alpha + gc0 + c4);
                        if (a.act == ACT_SNAKEBETA) be = ld4(a.act_alpha + a.groups * a.N + gc0 + c4);   // second half of the table
                        v = act_apply4(a.act, v, a.act_param, al, be);
                    }
                    if (a.gate) v *= ld4(a.gate + (long long)bz * a.gate_bs + gc0 + c4);
                    if (resp) v += ld4(resp + c4);
                    if (resph) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)resph[c4 + e];
                    }
                    if (res2p) v += ld4(res2p + c4);
                    v *= a.scale;
                    if (a.div != 0.0f) v = v / a.div;
                    if (a.out && a.out_f16) {
                        f16_t* op = reinterpret_cast<f16_t*>(a.out) + o1 + c4;
                        const f16x4 w4 = {f32_to_f16_sat(v[0]), f32_to_f16_sat(v[1]), f32_to_f16_sat(v[2]), f32_to_f16_sat(v[3])};
                        if (vec) *reinterpret_cast<f16x4*>(op) = w4;
                        else { op[0] = w4[0]; op[1] = w4[1]; op[2] = w4[2]; op[3] = w4[3]; }
                    } else if (a.out) {
                        if (sizeof(T) != 2 && a.out_planes) {
                            bf16_t* op = reinterpret_cast<bf16_t*>(a.out) + o1 + c4;
                            if (vec) store_planes4(op, a.out_plane, v);
                            else { for (int e = 0; e < 4; ++e) store_planes1(op + e, a.out_plane, v[e]); }
                        } else if (a.out_f32) {
                            float* op = reinterpret_cast<float*>(a.out) + o1 + c4;
                            if (vec) *reinterpret_cast<f32x4*>(op) = v;
                            else { op[0] = v[0]; op[1] = v[1]; op[2] = v[2]; op[3] = v[3]; }
                        } else {
                            T* op = reinterpret_cast<T*>(a.out) + o1 + c4;
                            if constexpr (sizeof(T) == 2) {
                                typename Vec4<T>::type w4 = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
                                if (vec) *reinterpret_cast<typename Vec4<T>::type*>(op) = w4;
                                else { op[0] = w4[0]; op[1] = w4[1]; op[2] = w4[2]; op[3] = w4[3]; }
                            } else {
                                if (vec) *reinterpret_cast<f32x4*>(op) = v;
                                else { op[0] = v[0]; op[1] = v[1]; op[2] = v[2]; op[3] = v[3]; }
                            }
                        }
                    }
                    if (a.out2) {
                        f32x4 al2 = {1.0f, 1.0f, 1.0f, 1.0f};
                        if (a.act2_alpha) al2 = ld4(a.act2_alpha + gc0 + c4);
                        f32x4 u;
                        u = act_apply4(a.act2, v, a.act2_param, al2, f32x4{1, 1, 1, 1});
                        T* op = reinterpret_cast<T*>(a.out2) + o2 + c4;
                        if constexpr (sizeof(T) == 2) {
                            typename Vec4<T>::type w4 = {from_f32<T>(u[0]), from_f32<T>(u[1]), from_f32<T>(u[2]), from_f32<T>(u[3])};
                            if (vec) *reinterpret_cast<typename Vec4<T>::type*>(op) = w4;
                            else { op[0] = w4[0]; op[1] = w4[1]; op[2] = w4[2]; op[3] = w4[3]; }
                        } else if (a.out2_planes) {
                            bf16_t* opp = reinterpret_cast<bf16_t*>(a.out2) + o2 + c4;
                            if (vec) store_planes4(opp, a.out2_plane, u);
                            else { for (int e = 0; e < 4; ++e) store_planes1(opp + e, a.out2_plane, u[e]); }
                        } else {
                            if (vec) *reinterpret_cast<f32x4*>(op) = u;
                            else { op[0] = u[0]; op[1] = u[1]; op[2] = u[2]; op[3] = u[3]; }
                        }
                    }
                }
            } else {
                // partial column tile: element-wise with the padding rules (cols [groups*N, out_cols) are zero-filled)
                for (int c = 0; c < 16; ++c) {
                    const int col = col0 + c, gc = gc0 + c;
                    const bool col_ok = col < a.N;
                    const bool col_pad = (!col_ok) && (g == a.groups - 1) && (gc < a.out_cols);
                    const bool col_pad2 = (!col_ok) && (g == a.groups - 1) && (gc < a.out2_cols);
                    if (!col_ok && !col_pad && !col_pad2) continue;
                    float v = 0.0f;
                    float al2 = 1.0f;
                    if (col_ok) {
                        const float bi = a.bias ? a.bias[gc] : 0.0f;
                        const float al1 = a.act_alpha ? a.act_alpha[gc] : 1.0f;
                        al2 = a.act2_alpha ? a.act2_alpha[gc] : 1.0f;
                        const float gt = a.gate ? a.gate[(long long)bz * a.gate_bs + gc] : 1.0f;
                        const float be1 = a.act == ACT_SNAKEBETA ? a.act_alpha[a.groups * a.N + gc] : 1.0f;
                        v = act_apply(a.act, x[c] + bi, a.act_param, al1, be1) * gt;
                        if (a.res && a.res_f16) v += (float)reinterpret_cast<const f16_t*>(a.res)[(long long)bz * a.res_bs + (long long)(row + a.res_row_off) * a.ldres + gc];
                        else if (a.res) v += a.res[(long long)bz * a.res_bs + (long long)(row + a.res_row_off) * a.ldres + gc];
                        if (a.res2) v += a.res2[(long long)bz * a.res2_bs + (long long)row * a.ldres2 + gc];
                        v *= a.scale;
                        if (a.div != 0.0f) v = v / a.div;
                    }
                    if (a.out && (col_ok || col_pad)) {
                        const long long o = ob + (long long)(row + a.out_row_off) * a.ldo + gc;
                        if (a.out_f16) reinterpret_cast<f16_t*>(a.out)[o] = f32_to_f16_sat(v);
                        else if (sizeof(T) != 2 && a.out_planes) store_planes1(reinterpret_cast<bf16_t*>(a.out) + o, a.out_plane, v);
                        else if (a.out_f32) reinterpret_cast<float*>(a.out)[o] = v;
                        else reinterpret_cast<T*>(a.out)[o] = from_f32<T>(v);
                    }
                    if (a.out2 && (col_ok || col_pad2)) {
                        const long long o2 = (long long)bz * a.out2_bs + (long long)(row + a.out2_row_off) * a.ldo2 + gc;
                        const float u = col_ok ? act_apply(a.act2, v, a.act2_param, al2) : 0.0f;
                        if (sizeof(T) != 2 && a.out2_planes) store_planes1(reinterpret_cast<bf16_t*>(a.out2) + o2, a.out2_plane, u);
                        else reinterpret_cast<T*>(a.out2)[o2] = from_f32<T>(u);
                    }
                }
            }
        };
        do_pass(std::integral_constant<int, 0>{});
        if constexpr (MT > 1 * MT_PASS) do_pass(std::integral_constant<int, 1 * MT_PASS>{});
        if constexpr (MT > 2 * MT_PASS) do_pass(std::integral_constant<int, 2 * MT_PASS>{});
        if constexpr (MT > 3 * MT_PASS) do_pass(std::integral_constant<int, 3 * MT_PASS>{});
        if constexpr (MT > 4 * MT_PASS) do_pass(std::integral_constant<int, 4 * MT_PASS>{});
        if constexpr (MT > 5 * MT_PASS) do_pass(std::integral_constant<int, 5 * MT_PASS>{});
        if constexpr (MT > 6 * MT_PASS) do_pass(std::integral_constant<int, 6 * MT_PASS>{});
        if constexpr (MT > 7 * MT_PASS) do_pass(std::integral_constant<int, 7 * MT_PASS>{});
        static_assert(MT <= 8 * MT_PASS, "epilogue passes are unrolled by hand up to 8");
    } else {   // EPI_QKV_DIT: a wave's WN columns lie inside one of q / k / v and one head
        const int D = a.heads * 64;
        const int cw = nw0;                          // first column of this wave
        const bool wave_ok = cw < a.N;
        const int which = wave_ok ? cw / D : 0;
        const int cb = cw - which * D;                    // channel inside q / k / v
        const int h = cb >> 6;
        constexpr int LPR = WN / 16;
        if constexpr (LEAN && WN == 64) {
            // streamlined form (the caller's staging tile holds 32 rows): the bias is loaded once, q / k leave as before (a 128-byte head row
            // per 4 lanes), V^T is transposed 32 time steps at a time so that a lane writes 64 contiguous bytes of its channel's row
            if (wave_ok && which < 2) {
                // 8 lanes per row: store u of a pass writes rows [8 u, 8 u + 8) of the head's [t][64] block = 1 KB of contiguous memory
                const int row8 = lane >> 3, c8 = (lane & 7) * 8;
                const bool rope_rt = cb < 64 && a.rope_cos;            // wave-uniform (cb is a multiple of 64): channels [0, 64) of q and k = head 0
                const float qs = (which == 0 && a.q_scale != 0.0f) ? a.q_scale : 1.0f;
                f32x4 bi[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) bi[hh] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + cw + c8 + hh * 4) : f32x4{0, 0, 0, 0};
                QT* const dbase = reinterpret_cast<QT*>(which == 0 ? a.q : a.k) + (((long long)bz * a.heads + h) * a.t_pad) * 64 + c8;
                auto qk_pass = [&](auto IP, auto ROPE) __attribute__((always_inline)) {
                    constexpr int ip = decltype(IP)::value;
                    constexpr bool rope = decltype(ROPE)::value;       // compile-time: the pass stays one basic block (see LeanMode)
                    const int row0 = mw0 + ip * 16 + row8;
                    f32x4 cs[2], sn[2];
                    if (rope) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int row = row0 + 8 * u;
                            const long long ro = (long long)(row < a.M ? row : a.M - 1) * 32 + (c8 >> 1);
                            cs[u] = *reinterpret_cast<const f32x4*>(a.rope_cos + ro);
                            sn[u] = *reinterpret_cast<const f32x4*>(a.rope_sin + ro);
                        }
                    }
                    stage(IP);
                    f32x4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(&scr[(row8 + 8 * (q >> 1)) * SLD + c8 + (q & 1) * 4]) + bi[q & 1];
                    unstage();
                    if (rope) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; e += 2) {
                                const float c_ = cs[q >> 1][(q & 1) * 2 + (e >> 1)], s_ = sn[q >> 1][(q & 1) * 2 + (e >> 1)];
                                const float ev = v[q][e], od = v[q][e + 1];
                                v[q][e] = ev * c_ - od * s_;
                                v[q][e + 1] = od * c_ + ev * s_;
                            }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        typename Vec8<QT>::type w8;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { w8[e] = from_f32<QT>(v[2 * u][e] * qs); w8[4 + e] = from_f32<QT>(v[2 * u + 1][e] * qs); }
                        if (row0 + 8 * u < a.M) store8(dbase + (long long)(row0 + 8 * u) * 64, w8);
                    }
                };
                auto qk_all = [&](auto ROPE) __attribute__((always_inline)) {
                    qk_pass(std::integral_constant<int, 0>{}, ROPE);
                    if constexpr (MT > 1) qk_pass(std::integral_constant<int, 1>{}, ROPE);
                    if constexpr (MT > 2) qk_pass(std::integral_constant<int, 2>{}, ROPE);
                    if constexpr (MT > 3) qk_pass(std::integral_constant<int, 3>{}, ROPE);
                    if constexpr (MT > 4) qk_pass(std::integral_constant<int, 4>{}, ROPE);
                    if constexpr (MT > 5) qk_pass(std::integral_constant<int, 5>{}, ROPE);
                    if constexpr (MT > 6) qk_pass(std::integral_constant<int, 6>{}, ROPE);
                    if constexpr (MT > 7) qk_pass(std::integral_constant<int, 7>{}, ROPE);
                };
                if (rope_rt) qk_all(std::true_type{});
                else qk_all(std::false_type{});
                static_assert(MT <= 8 && MT % 2 == 0, "lean QKV epilogue: up to 8 row tiles per wave, in pairs");
                return;
            }
            if (wave_ok) {
                // V^T [d][t]: 4 lanes per channel, each 8 of the 32 staged time steps: a store covers 16 channels x 64 contiguous bytes (one
                // channel per lane would scatter 64 separate 16-byte pieces per instruction).  The staging rows are 65 floats apart here so
                // that the transposed ds_read_b32 of the four time groups fall into different banks.
                constexpr int SLV = WN + 1;
                const int ch4 = lane >> 2, tq = lane & 3;
                float bi[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) bi[i] = a.bias ? a.bias[cw + i * 16 + ch4] : 0.0f;
                QT* const dbase = reinterpret_cast<QT*>(a.vT) + (((long long)bz * a.heads + h) * 64 + ((cb + ch4) & 63)) * a.t_pad + tq * 8;
                auto v_pass = [&](auto IP) __attribute__((always_inline)) {
                    constexpr int ip = decltype(IP)::value;
                    if constexpr (std::is_same<AT, f32x4[MT][NT]>::value) {
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                            for (int j = 0; j < NT; ++j)
#pragma unroll
                                for (int r = 0; r < 4; ++r) scr[(ii * 16 + fg * 4 + r) * SLV + j * 16 + fr] = acc[ip + ii][j][r];
                    } else acc_stage<MT / 2, NT / 2, 2, ip>(acc, scr, SLV, lane);
                    wave_lds_order();
                    float x[4][8];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[i][e] = scr[(tq * 8 + e) * SLV + i * 16 + ch4] + bi[i];
                    unstage();
                    const int row0 = mw0 + ip * 16;
                    if (row0 >= a.M) return;
                    QT* dst = dbase + row0;
                    if (row0 + 32 <= a.M) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            typename Vec8<QT>::type w8;
#pragma unroll
                            for (int e = 0; e < 8; ++e) w8[e] = from_f32<QT>(x[i][e]);
                            store8(dst + (long long)i * 16 * a.t_pad, w8);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (row0 + tq * 8 + e < a.M) dst[(long long)i * 16 * a.t_pad + e] = from_f32<QT>(x[i][e]);
                    }
                };
                v_pass(std::integral_constant<int, 0>{});
                if constexpr (MT > 2) v_pass(std::integral_constant<int, 2>{});
                if constexpr (MT > 4) v_pass(std::integral_constant<int, 4>{});
                if constexpr (MT > 6) v_pass(std::integral_constant<int, 6>{});
            }
            return;
        }
        auto do_pass = [&](auto IP) __attribute__((always_inline)) {
            constexpr int ip = decltype(IP)::value;
            stage(IP);
            if (which < 2) {
                // q, k: row-major, 16 consecutive channels per lane; interleaved-pair RoPE on channels [0, 64) of the row
                const int prow = lane / LPR, pcs = (lane % LPR) * 16;
                float x[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(&scr[prow * SLD + pcs + q * 4]);
                    x[q * 4 + 0] = t[0]; x[q * 4 + 1] = t[1]; x[q * 4 + 2] = t[2]; x[q * 4 + 3] = t[3];
                }
                unstage();
                const int row = mw0 + ip * 16 + prow;
                const int c0 = cb + pcs, d0 = c0 & 63;
                if (!wave_ok || row >= a.M) return;
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] += a.bias ? a.bias[cw + pcs + c] : 0.0f;
                if (c0 < 64 && a.rope_cos) {                 // (no table: plain multi-head QKV, e.g. the Matcha transformer blocks)
#pragma unroll
                    for (int c = 0; c < 16; c += 2) {
                        const float cs = a.rope_cos[(long long)row * 32 + ((d0 + c) >> 1)], sn = a.rope_sin[(long long)row * 32 + ((d0 + c) >> 1)];
                        const float e = x[c], o = x[c + 1];
                        x[c] = e * cs - o * sn;
                        x[c + 1] = o * cs + e * sn;
                    }
                }
                if (which == 0 && a.q_scale != 0.0f) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) x[c] *= a.q_scale;
                }
                QT* dst = reinterpret_cast<QT*>(which == 0 ? a.q : a.k) + ((((long long)bz * a.heads + h) * a.t_pad) + row) * 64 + d0;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    typename Vec8<QT>::type w8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) w8[e] = from_f32<QT>(x[q * 8 + e]);
                    store8(dst + q * 8, w8);
                }
            } else {
                // v: written transposed (V^T [d][t]): a lane takes one channel and 16 consecutive time steps
                const int pc = lane % WN, rblk = lane / WN;
                float x[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = scr[(rblk * 16 + r) * SLD + pc];
                unstage();
                const int row0 = mw0 + ip * 16 + rblk * 16;
                if (!wave_ok || row0 >= a.M) return;
                const float bi = a.bias ? a.bias[cw + pc] : 0.0f;
                const int d = (cb + pc) & 63;
                QT* dst = reinterpret_cast<QT*>(a.vT) + (((long long)bz * a.heads + h) * 64 + d) * a.t_pad + row0;
                if (row0 + 16 <= a.M) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        typename Vec8<QT>::type w8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) w8[e] = from_f32<QT>(x[q * 8 + e] + bi);
                        store8(dst + q * 8, w8);
                    }
                } else {
                    for (int r = 0; r < 16 && row0 + r < a.M; ++r) dst[r] = from_f32<QT>(x[r] + bi);
                }
            }
        };
        do_pass(std::integral_constant<int, 0>{});
        if constexpr (MT > 1 * MT_PASS) do_pass(std::integral_constant<int, 1 * MT_PASS>{});
        if constexpr (MT > 2 * MT_PASS) do_pass(std::integral_constant<int, 2 * MT_PASS>{});
        if constexpr (MT > 3 * MT_PASS) do_pass(std::integral_constant<int, 3 * MT_PASS>{});
        if constexpr (MT > 4 * MT_PASS) do_pass(std::integral_constant<int, 4 * MT_PASS>{});
        if constexpr (MT > 5 * MT_PASS) do_pass(std::integral_constant<int, 5 * MT_PASS>{});
        if constexpr (MT > 6 * MT_PASS) do_pass(std::integral_constant<int, 6 * MT_PASS>{});
        if constexpr (MT > 7 * MT_PASS) do_pass(std::integral_constant<int, 7 * MT_PASS>{});
        static_assert(MT <= 8 * MT_PASS, "epilogue passes are unrolled by hand up to 8");
    }
}


}  // namespace hvx
