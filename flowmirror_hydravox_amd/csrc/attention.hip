// attention.hip — flash-style attention, head_dim 64, for gfx950 (see hvx_kernels.h: AttnArgs).
//
// Generic form (attn_fwd_kernel): one wave64 owns QT 16-row query tiles and walks its key range 32 keys at a
// time with K / V^T fragments straight from L2 (both are produced fragment-friendly by the fused QKV
// epilogues — K as [key][64], V already transposed as V^T [64][key]); LLM prefill / decode and fp32 parity.
// Long bf16 sequences (DiT) take attn_dit_kernel below: K / V^T tiles staged in LDS and shared by 4 waves.
//
// Orientation (everything stays in registers):
//   S^T[key, q] = K . Q^T          A = K rows (8 consecutive d per lane), B = Q^T (8 consecutive d per lane)
//   C-layout of S^T: lane holds column q = lane & 15 and keys 4g..4g+3 (g = lane >> 4) of each 16-key tile
//   -> a query's scores live in the 4 lanes {q, q+16, q+32, q+48}: row max / sum = 2 xor-shuffles.
//   O^T[d, q] += V^T . P^T         the MFMA k-slot (g, j) is *defined* as key 4g+j (j<4) / 16+4g+(j-4) (j>=4)
//   of the 32-key step, which is exactly what the lane already holds after the S^T MFMAs (no permute,
//   no LDS round trip); V^T is read with the same slot->key map (two 4-key vectors per lane).
// Used by: DiT self-attention (non-causal, key padding via kv_len), LLM prefill and decode (causal,
// GQA-packed: the 7 query heads of a KV head are stacked as rows so each K/V byte is read once,
// optional key splits with (m, l, o) partials combined by attn_combine_kernel).
#include <type_traits>

#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

__device__ __forceinline__ bf16x4 load4(const bf16_t* p) { return *reinterpret_cast<const bf16x4*>(p); }
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
// maxima of values that are never NaN here (scores and -inf masks): this file is built with -fno-honor-nans (build.py) so that
// they become plain v_max / v_max3 without a canonicalising self-max in front of every operand
__device__ __forceinline__ float fmax2(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// KVF: K / V^T are the LLM's fragment-order caches — every operand fragment is ONE contiguous load (1 KiB per wave instruction at bf16) instead of
// 16 key rows x 64 B (K) or 64 channel rows x 8 B twice (V^T)
// NWV: waves per workgroup of the decode (MERGE) form — 4, or 8 on wide grids (one 64-key trip per wave over a 512-key split: every load of the
// split is in flight at once, and the eight partial states still merge in LDS into ONE partial per workgroup)
template <class T, int QT, bool MERGE, bool KVF, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void attn_fwd_kernel(AttnArgs a) {
    typedef typename Vec8<T>::type V8;
    typedef typename Vec4<T>::type V4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    constexpr bool merge = MERGE;                             // decode form: workgroup = one split, waves = its four quarters
    int b = blockIdx.z, h = blockIdx.y, sp_m = blockIdx.x;
    if constexpr (merge) {
        // Workgroups go to the 8 XCDs round-robin in LAUNCH order (x fastest).  With (split, head, sequence) = (x, y, z) and 8 splits — capacity 4096,
        // 512-key splits — split s would run on XCD s only, and a grid whose sequences are 1536 keys long would use 3 of the 8 XCDs (measured: the
        // 64-sequence decode step 1.29 -> 1.46 ms).  The launch order is re-read as split-major: consecutive workgroups are the (head, sequence) pairs of one
        // split, so every split is spread over all XCDs, and the live (low) splits are dispatched first.
        const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), HB = gridDim.y * gridDim.z;
        sp_m = L / HB;
        const int rem = L - sp_m * HB;
        b = rem / (int)gridDim.y;
        h = rem - b * (int)gridDim.y;
    }
    const int n_qt = (a.n_rows + 16 * QT - 1) / (16 * QT);
    const int w = merge ? sp_m : blockIdx.x * 4 + wave;
    if (!merge && w >= n_qt * a.n_splits) return;
    const int qt = merge ? 0 : w / a.n_splits, sp = merge ? sp_m : w - qt * a.n_splits;

    const int kv_len = a.kv_len ? a.kv_len[b] : a.kv_len_const;
    const int pos0 = (a.causal && a.pos0) ? a.pos0[b] : 0;
    const int n_valid_lo = a.n_valid_lo ? a.n_valid_lo[b] : a.kn;
    int key_begin = 0, key_end = kv_len;
    if (a.n_splits > 1) {
        key_begin = sp * a.split_chunk;
        // The grid is sized by the cache CAPACITY (a captured launch cannot know the live lengths): a split that starts beyond every row's visible keys
        // leaves at once — before the query loads, the LDS merge and the partial stores.  attn_combine_kernel reads only the first
        // ceil(visible / split_chunk) splits of a row, so nothing of a dead split is ever looked at.  (At capacity 4096 and context 1536 five of
        // eight workgroups are dead; running their epilogues cost the 64-sequence decode step 190 us.)
        if (merge && key_begin >= (a.causal ? min(kv_len, pos0 + a.kn) : kv_len)) return;
        key_end = min(kv_len, key_begin + a.split_chunk);
        if (merge) {
            key_begin += wave * a.sub_chunk;
            key_end = min(key_end, key_begin + a.sub_chunk);
        }
    }

    // ---- this lane's query rows (one per q-tile) -----------------------------------------------------
    int r_lo[QT];
    bool r_ok[QT];
    V8 qf[QT][2];
    int max_pos = -1, max_lim = 0;
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const int r = (qt * QT + i) * 16 + fr;
        const int rh = r / a.kn;
        r_lo[i] = r - rh * a.kn;
        r_ok[i] = (r < a.n_rows) && (r_lo[i] < n_valid_lo);
        if (r_ok[i]) {
            const T* qp = reinterpret_cast<const T*>(a.q) + (long long)b * a.q_bs + (long long)h * a.q_hs + (long long)rh * a.q_hi +
                          (long long)r_lo[i] * a.q_lo + fg * 8;
            qf[i][0] = load8(qp);
            qf[i][1] = load8(qp + 32);
            max_pos = max(max_pos, pos0 + r_lo[i]);
            if (a.chunk > 0) max_lim = max(max_lim, (r_lo[i] / a.chunk + 1) * a.chunk);
        } else {
            qf[i][0] = zero8<T>();
            qf[i][1] = zero8<T>();
        }
    }
    if (a.causal) {
        // no key beyond the largest visible position of this wave's rows needs to be touched
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) max_pos = max(max_pos, __shfl_xor(max_pos, o, 64));
        key_end = min(key_end, max_pos + 1);
    }
    if (a.chunk > 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) max_lim = max(max_lim, __shfl_xor(max_lim, o, 64));
        key_end = min(key_end, max_lim);
    }

    const int slot = a.kv_slot ? a.kv_slot[b] : b;
    const T* __restrict__ kb = reinterpret_cast<const T*>(a.k) + (long long)slot * a.k_bs + (long long)h * a.k_hs;
    const T* __restrict__ vb = reinterpret_cast<const T*>(a.vT) + (long long)slot * a.v_bs + (long long)h * a.v_hs;

    float m_run[QT], l_run[QT];
    f32x4 o_acc[QT][4];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        m_run[i] = -INFINITY;
        l_run[i] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o_acc[i][dt] = f32x4{0, 0, 0, 0};
    }

    // KS 32-key steps per trip.  The decode form (one q-tile) requests both steps of a wave's 64 keys before the first MFMA: the walk
    // is a chain of dependent round trips to L2 / HBM otherwise, and latency is all a decode step has to lose.
    constexpr int KS = (QT == 1 || MERGE) ? 2 : 1;
    for (int key0 = key_begin; key0 < key_end; key0 += 32 * KS) {
        // K fragments: 2 key tiles x 2 d-halves.  Rows past the end are clamped (their scores are masked by select).
        V8 kf[KS][2][2];
        // V^T fragments: 4 d-tiles, keys {4g..4g+3} and {16+4g..16+4g+3} of each 32-key step
        V8 vf[KS][4];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // a step that starts past the end (only the second of a trip can) re-reads the first one and is skipped below
            const int kbase = (key0 + 32 * ks < key_end) ? key0 + 32 * ks : key0;
            if constexpr (KVF) {
                // (a step may reach up to 31 keys past kv_len: the cache holds them — capacity is a multiple of 32 — and their scores are masked)
                const T* kp = kb + (long long)(kbase >> 4) * 1024 + lane * 8;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    kf[ks][kt][0] = load8(kp + kt * 1024);
                    kf[ks][kt][1] = load8(kp + kt * 1024 + 512);
                }
                const T* vp = vb + (long long)(kbase >> 5) * 2048 + lane * 8;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) vf[ks][dt] = load8(vp + dt * 512);
                continue;
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                int key = kbase + kt * 16 + fr;
                key = key < kv_len ? key : (kv_len - 1);
                const T* kp = kb + (long long)key * 64 + fg * 8;
                kf[ks][kt][0] = load8(kp);
                kf[ks][kt][1] = load8(kp + 32);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const T* vp = vb + (long long)(dt * 16 + fr) * a.v_ld + kbase + fg * 4;
                const V4 lo = load4(vp), hi = load4(vp + 16);
                V8 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = lo[j];
                    v[4 + j] = hi[j];
                }
                vf[ks][dt] = v;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kstep = key0 + 32 * ks;
            if (kstep >= key_end) break;
#pragma unroll
            for (int i = 0; i < QT; ++i) {
                f32x4 s[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    s[kt] = f32x4{0, 0, 0, 0};
                    mma32(s[kt], kf[ks][kt][0], qf[i][0]);
                    mma32(s[kt], kf[ks][kt][1], qf[i][1]);
                }
                // scale, mask, running max
                int lim = a.causal ? min(key_end, pos0 + r_lo[i] + 1) : key_end;
                if (a.chunk > 0) lim = min(lim, (r_lo[i] / a.chunk + 1) * a.chunk);
                float sv[8];
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kstep + kt * 16 + fg * 4 + r;
                        const float x = (key < lim) ? s[kt][r] * a.scale : -INFINITY;
                        sv[kt * 4 + r] = x;
                        mx = fmaxf(mx, x);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[i], mx);
                const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
                const float alpha = (m_run[i] == -INFINITY) ? 0.0f : expf(m_run[i] - m_safe);
                float psum = 0.0f;
                V8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float p = (sv[e] == -INFINITY) ? 0.0f : expf(sv[e] - m_safe);
                    psum += p;
                    pf[e] = from_f32<T>(p);
                }
                l_run[i] = l_run[i] * alpha + psum;
                m_run[i] = m_new;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    o_acc[i][dt] *= alpha;
                    mma32(o_acc[i][dt], vf[ks][dt], pf);
                }
            }
        }
    }

    // ---- finish -------------------------------------------------------------------------------------
    if constexpr (MERGE) {
        // merge the waves' partial softmax states in LDS (fixed wave order) and write ONE partial per workgroup
        constexpr int RW = 16 * QT;
        __shared__ float mo[NWV][RW][64];
        __shared__ float mml[NWV][RW][2];
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            float l = l_run[i];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) mo[wave][i * 16 + fr][dt * 16 + fg * 4 + e] = o_acc[i][dt][e];
            if (fg == 0) {
                mml[wave][i * 16 + fr][0] = m_run[i];
                mml[wave][i * 16 + fr][1] = l;
            }
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < QT; ++pass) {
            const int row = pass * 16 + ((threadIdx.x & 255) >> 4), d0 = (threadIdx.x & 15) * 4;
            const int rh = row / a.kn, rl = row - rh * a.kn;
            if (threadIdx.x < 256 && row < a.n_rows && rl < n_valid_lo) {
                float m = -INFINITY;
#pragma unroll
                for (int ww = 0; ww < NWV; ++ww) m = fmaxf(m, mml[ww][row][0]);
                f32x4 acc = {0, 0, 0, 0};
                float ls = 0.0f;
#pragma unroll
                for (int ww = 0; ww < NWV; ++ww) {
                    const float mw = mml[ww][row][0];
                    const float wgt = (mw == -INFINITY) ? 0.0f : expf(mw - m);
                    acc += wgt * *reinterpret_cast<const f32x4*>(&mo[ww][row][d0]);
                    ls += wgt * mml[ww][row][1];
                }
                const long long base = (((long long)b * a.heads + h) * a.n_splits + sp) * a.n_rows_pad + row;
                *reinterpret_cast<f32x4*>(a.part_o + base * 64 + d0) = acc;
                if (d0 == 0) {
                    a.part_ml[base * 2 + 0] = m;
                    a.part_ml[base * 2 + 1] = ls;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        float l = l_run[i];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int r = (qt * QT + i) * 16 + fr;
        if (!r_ok[i]) continue;
        const int rh = r / a.kn;
        if (a.n_splits == 1) {
            const float inv = l > 0.0f ? 1.0f / l : 0.0f;
            const long long off = (long long)b * a.o_bs + (long long)h * a.o_hs + (long long)rh * a.o_hi + (long long)r_lo[i] * a.o_lo;
            T* op = reinterpret_cast<T*>(a.out) + off;
            if (a.o_frag_kt > 0) {                            // fragment-order activation matrix (the o_proj of a wide decode grid reads it)
                const int ncol = a.o_frag_kt * 32, orow = (int)(off / ncol), ocol = (int)(off - (long long)orow * ncol);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        reinterpret_cast<T*>(a.out)[frag_index(orow, ocol + dt * 16 + fg * 4 + e, a.o_frag_kt)] = from_f32<T>(o_acc[i][dt][e] * inv);
                continue;
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) op[dt * 16 + fg * 4 + e] = from_f32<T>(o_acc[i][dt][e] * inv);
        } else {
            const long long base = (((long long)b * a.heads + h) * a.n_splits + sp) * a.n_rows_pad + r;
            float* po = a.part_o + base * 64;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) po[dt * 16 + fg * 4 + e] = o_acc[i][dt][e];
            if (fg == 0) {
                a.part_ml[base * 2 + 0] = m_run[i];
                a.part_ml[base * 2 + 1] = l;
            }
        }
    }
}

// out[row, :] = sum_s exp(m_s - m) o_s / sum_s exp(m_s - m) l_s      (fixed split order: deterministic)
template <class T>
__global__ void attn_combine_kernel(AttnArgs a) {
    const int b = blockIdx.z, h = blockIdx.y;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int d = threadIdx.x & 63;
    if (r >= a.n_rows) return;
    const int rh = r / a.kn, rl = r - rh * a.kn;
    const int n_valid_lo = a.n_valid_lo ? a.n_valid_lo[b] : a.kn;
    if (rl >= n_valid_lo) return;
    // splits whose key range lies beyond this row's visible keys wrote nothing: only the first n_live splits are read
    const int kv_len = a.kv_len ? a.kv_len[b] : a.kv_len_const;
    int vis = kv_len;
    if (a.causal) vis = min(vis, (a.pos0 ? a.pos0[b] : 0) + rl + 1);
    const int n_live = min(a.n_splits, (vis + a.split_chunk - 1) / a.split_chunk);
    const long long base0 = (((long long)b * a.heads + h) * a.n_splits) * a.n_rows_pad + r;
    const long long sstride = a.n_rows_pad;
    // one round trip per 8 splits: their (m, l, o) are all requested before the first is used; batches are merged online
    constexpr int SB = 8;
    float m = -INFINITY, acc = 0.0f, l = 0.0f;
    for (int s0 = 0; s0 < n_live; s0 += SB) {
        float ms[SB], ls[SB], os[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const long long base = base0 + (long long)min(s0 + u, n_live - 1) * sstride;
            const float2 ml = *reinterpret_cast<const float2*>(a.part_ml + base * 2);
            ms[u] = (s0 + u < n_live) ? ml.x : -INFINITY;
            ls[u] = ml.y;
            os[u] = a.part_o[base * 64 + d];
        }
        float mb = m;
#pragma unroll
        for (int u = 0; u < SB; ++u) mb = fmaxf(mb, ms[u]);
        const float resc = (m == -INFINITY) ? 0.0f : expf(m - mb);
        acc *= resc;
        l *= resc;
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const float wgt = (ms[u] == -INFINITY) ? 0.0f : expf(ms[u] - mb);
            acc += wgt * os[u];
            l += wgt * ls[u];
        }
        m = mb;
    }
    const long long off = (long long)b * a.o_bs + (long long)h * a.o_hs + (long long)rh * a.o_hi + (long long)rl * a.o_lo;
    T* op = reinterpret_cast<T*>(a.out) + off;
    if (a.o_frag_kt > 0) {                                    // fragment-order activation matrix (the o_proj of a wide decode grid reads it)
        const int ncol = a.o_frag_kt * 32, orow = (int)(off / ncol), ocol = (int)(off - (long long)orow * ncol);
        op = reinterpret_cast<T*>(a.out) + frag_index(orow, ocol + d, a.o_frag_kt) - d;
    }
    op[d] = from_f32<T>(l > 0.0f ? acc / l : 0.0f);
}

// ---------------------------------------------------------------------------------------------------------------------
// Long-sequence bf16 form (DiT: non-causal, thousands of keys): a workgroup of 4 waves owns 128 query rows (32 per wave) and
// walks the keys 64 at a time; the K tile [64 keys][64] and the V^T tile [64 d][64 keys] are staged once per workgroup in LDS
// (double-buffered, global -> registers -> LDS with the next tile's loads in flight during the MFMAs) instead of once per
// wave from L2.  Same register-resident S^T / P^T orientation as above.  Softmax runs on exp2 with the scale folded into one
// fma per score, and the key-padding mask is only evaluated on the last tile.
// ---------------------------------------------------------------------------------------------------------------------
//
// FAST form (the default): the softmax of a 64-key tile is phase-pure vector work between two phase-pure MFMA blocks, and the two waves
// that share a SIMD (one from each resident workgroup) run the same phases — the matrix cores idle while both exponentiate.  The fast
// tile therefore (1) drops the running maximum: every row keeps ONE reference m_ref (its maximum over the first key tile, found in a
// prologue) and p = exp2(s c - m_ref) may exceed 1 — softmax is shift-invariant, bf16 keeps its relative precision at any magnitude
// and the sums are fp32 — which removes the max chain, the two cross-lane exchanges and the rescale branch, leaving one basic block of
// 72 MFMAs and ~130 VALU per wave and tile; (2) software-pipelines that block by 16-row query tile inside the wave:
//     QK(0) | QK(1) + SM(0) | QK(2) + SM(1) + PV(0) | QK(3) + SM(2) + PV(1) | SM(3) + PV(2) | PV(3)
// with the MFMA : VALU interleave pinned by sched_group_barrier.  A score that outgrows its reference by more than 2^127 overflows to
// inf; that is detected at the end (l or O not finite, workgroup-wide vote) and the workgroup then simply runs again with the classical
// online-softmax loop below (FAST = false path; exercised by tests/test_gpu_ops.py::test_attention_dit_fallback_on_score_spike).
// PRE: the query operand already carries scale * log2(e) (a.q_log2: written that way by the fused QKV epilogue, one rounding) — the MFMA
// accumulator then starts at -m_ref and exp2 is the only arithmetic left per score.  Otherwise the scale is applied to the fp32 score.
// LAB (tools/attn_probe.py with option attn_lab = n, library built with -DHVX_LAB): timing-only variants that REMOVE one ingredient of the fast loop
// each (results are garbage): 1 no global loads / LDS stash of the next tile, 2 no barrier, 4 K / V^T fragments from registers instead of LDS,
// 8 no exp2 / bf16 conversion, 16 no MFMA.  0 = the product.
// NW: waves per workgroup.  4: 64 QR rows per workgroup, two workgroups per CU.  8: 128 QR rows per workgroup, ONE per CU — the same two waves per SIMD,
// but a K / V^T tile is staged once for twice the rows, so every thread moves half the bytes per tile (one 16-byte piece of K and one of V^T).
// ROT 1 (round 6; the default where it exists: 64-row waves, pre-scaled queries, no chunk mask): the in-wave pipeline ROTATED ACROSS key tiles.  Every step of the loop is
// QK(i + 1) + SM(i) + PV(i - 1) — 18 MFMAs against 16 exponentials + 8 conversions — including the tile boundary (QK(0) of tile t + 1 beside SM(3) and PV(2) of tile t; PV(3)
// of tile t beside QK(1) / SM(0) of tile t + 1), where the in-tile pipeline above runs 8 bare QK MFMAs at the head and 10 bare PV MFMAs at the tail of every tile and packs the
// vector work behind the 54 in between.  tools/mfma_valu_lab.hip prices the difference: ONE exponential behind a 16x16x32 MFMA is free (8.3 ns with or without), a second one
// costs 3.8 ns — so the step is written slot by slot (MFMA, exponential, MFMA, exponential, packed conversion; sched_barrier(0) between slots).  The K / V^T fragment registers
// are re-filled in place one MFMA pair behind their last use (K(t + 1) during QK(3, t), V(t) during PV(3, t - 1)), so the working set is the in-tile form's — minus the 16
// registers of the per-row reference, which this form drops (scores start from 0; see the step's comment and the vote after the loop): with them the loop spills.
// Measured at B = 8, H = 16, T = 5632: 849 against 967 us (profiles/r06_attn_tile_ab.md), flow solve 398.7 against 419.6 ms.
template <int QR, bool FAST, bool PRE, int LAB = 0, int NW = 4, int ROT = 0>   // 16-row query tiles per wave: every K / V^T fragment read from LDS feeds QR MFMAs
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_dit_kernel(AttnArgs a) {
    static_assert(ROT == 0 || (QR == 4 && FAST && PRE && LAB == 0 && NW == 4), "the rotated pipeline exists for the 64-row, pre-scaled fast form");
    typedef bf16_t T;
    constexpr int KT = 64;                 // keys per tile
    constexpr int LD = 72;                 // LDS row stride (elements): 144 B keeps the 16 rows of a lane group on distinct 16-B slots
    __shared__ __attribute__((aligned(16))) T Ks[2][KT * LD];
    __shared__ __attribute__((aligned(16))) T Vs[2][64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    // XCD-aware workgroup order: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2), so with the plain (row block,
    // head, batch) order the row blocks of ONE head land on all 8 XCDs and every L2 fetches that head's K / V^T (rocprofv3 FETCH_SIZE: 4.8x
    // the algorithmic bytes, profiles/r03_pmc_traffic.json).  Remapped, XCD e walks the row blocks of heads e, e + 8, ... one head after the
    // other: a head's K / V^T are fetched into one L2 and re-used there by all its row blocks (bijective when heads x batch is a multiple of 8).
    int b = blockIdx.z, h = blockIdx.y, bx0 = blockIdx.x;
    {
        const int nx = gridDim.x, nh = gridDim.y * gridDim.z;
        if ((nh & 7) == 0) {
            const int L = blockIdx.x + nx * (blockIdx.y + gridDim.y * blockIdx.z);
            const int j = L >> 3, hb = (j / nx) * 8 + (L & 7);
            bx0 = j % nx;
            h = hb % gridDim.y;
            b = hb / gridDim.y;
        }
    }
    // with a chunk mask the work per workgroup grows with its row index: dispatch the long ones first
    const int bx = a.chunk > 0 ? gridDim.x - 1 - bx0 : bx0;
    constexpr int WG_ROWS = 16 * QR * NW;
    constexpr int NLD = 8 / NW;            // 16-byte pieces of K (and of V^T) a thread moves per tile
    const int row0 = bx * WG_ROWS + wave * (16 * QR);
    const int kv_len = a.kv_len ? a.kv_len[b] : a.kv_len_const;
    const T* __restrict__ kb = reinterpret_cast<const T*>(a.k) + (long long)b * a.k_bs + (long long)h * a.k_hs;
    const T* __restrict__ vb = reinterpret_cast<const T*>(a.vT) + (long long)b * a.v_bs + (long long)h * a.v_hs;
    // Static chunk mask (streaming synthesis): row r sees the keys below the end of its chunk.  The workgroup walks the keys that
    // its last row sees; tiles below the limit of its first row need no mask, the rest take the masked form with per-row limits.
    const int wg_row0 = bx * WG_ROWS, wg_row1 = min(wg_row0 + WG_ROWS, a.n_rows) - 1;
    const int lim_hi = a.chunk > 0 ? min(kv_len, (wg_row1 / a.chunk + 1) * a.chunk) : kv_len;
    const int lim_lo = a.chunk > 0 ? min(kv_len, (wg_row0 / a.chunk + 1) * a.chunk) : kv_len;
    int lim[QR];
#pragma unroll
    for (int i = 0; i < QR; ++i) lim[i] = a.chunk > 0 ? min(kv_len, ((row0 + i * 16 + fr) / a.chunk + 1) * a.chunk) : kv_len;

    bf16x8 qf[QR][2];
#pragma unroll
    for (int i = 0; i < QR; ++i) {
        const int r = row0 + i * 16 + fr;
        if (r < a.n_rows) {
            const T* qp = reinterpret_cast<const T*>(a.q) + (long long)b * a.q_bs + (long long)h * a.q_hs + (long long)r * a.q_lo + fg * 8;
            qf[i][0] = load8(qp);
            qf[i][1] = load8(qp + 32);
        } else {
            qf[i][0] = zero8<T>();
            qf[i][1] = zero8<T>();
        }
    }
    // tile loader: thread t moves NLD x 16 B of K (rows t/8 [and t/8+32], chunk t%8) and NLD x 16 B of V^T
    const int lrow = tid >> 3, lchunk = (tid & 7) * 8;
    bf16x8 rk[NLD], rv[NLD];
    auto gload = [&](int key0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int key = key0 + lrow + i * 32;
            key = key < kv_len ? key : kv_len - 1;                       // clamped rows are masked below
            rk[i] = load8(kb + (long long)key * 64 + lchunk);
            rv[i] = load8(vb + (long long)(lrow + i * 32) * a.v_ld + key0 + lchunk);
        }
    };
    // V^T goes into LDS with the keys of every 32-key MFMA step permuted so that the 8 keys a lane needs for its P^T k-slots
    // (4g..4g+3 of the first 16-key score tile and of the second) are adjacent: position = 32m + 8g + 4h + r for key = 32m + 16h +
    // 4g + r.  The PV operand is then one conflict-free 16-byte LDS read instead of two 8-byte reads with 2-way bank conflicts.
    const int vc = tid & 7;                                              // 8-key chunk: m = vc / 4, h = (vc % 4) / 2, g0 = 2 * (vc % 2)
    const int vpos = (vc >> 2) * 32 + ((vc & 1) * 2) * 8 + ((vc >> 1) & 1) * 4;
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            store8(&Ks[buf][(lrow + i * 32) * LD + lchunk], rk[i]);
            bf16x4 lo, hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = rv[i][j];
                hi[j] = rv[i][4 + j];
            }
            *reinterpret_cast<bf16x4*>(&Vs[buf][(lrow + i * 32) * LD + vpos]) = lo;
            *reinterpret_cast<bf16x4*>(&Vs[buf][(lrow + i * 32) * LD + vpos + 8]) = hi;
        }
    };

    float m_run[QR];
    f32x4 o_acc[QR][4];
    float l_acc[QR];
#pragma unroll
    for (int i = 0; i < QR; ++i) {
        m_run[i] = -INFINITY;
        l_acc[i] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o_acc[i][dt] = f32x4{0, 0, 0, 0};
    }
    const float c = PRE ? 1.0f : a.scale * 1.4426950408889634f;         // softmax(s * scale) == exp2(s * c - m * c); PRE: c sits in q
    const f32x2 c2 = {c, c};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = f32_to_bf16(1.0f);

    // One 64-key tile.  The softmax is the bottleneck of this kernel, not the matrix cores (head_dim 64: 32 MFMAs against ~32
    // exp2 + ~100 other VALU instructions per wave and tile), so everything that can leave the vector ALU does: the row sums are
    // one more MFMA against a fragment of ones (which also sums exactly the bf16 probabilities that multiply V), the key-padding
    // mask exists only in the peeled last tile, scale and max subtraction are packed fp32 FMAs.
    auto tile = [&](int buf, int key0, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        // S^T = K Q^T for all query tiles of the wave; each K fragment is read from LDS once and used QR times
        f32x4 s[QR][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const bf16x8 k0 = load8(&Ks[buf][(kt * 16 + fr) * LD + fg * 8]);
            const bf16x8 k1 = load8(&Ks[buf][(kt * 16 + fr) * LD + 32 + fg * 8]);
#pragma unroll
            for (int i = 0; i < QR; ++i) {
                s[i][kt] = f32x4{0, 0, 0, 0};
                mma32(s[i][kt], k0, qf[i][0]);
                mma32(s[i][kt], k1, qf[i][1]);
            }
        }
        bf16x8 pf[QR][2];
#pragma unroll
        for (int i = 0; i < QR; ++i) {
            if constexpr (TAIL) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key0 + kt * 16 + fg * 4 + r >= lim[i]) s[i][kt][r] = -INFINITY;
            }
            float mx = fmax3(fmax3(s[i][0][0], s[i][0][1], s[i][0][2]), fmax3(s[i][0][3], s[i][1][0], s[i][1][1]),
                             fmax3(s[i][1][2], s[i][1][3], s[i][2][0]));
            mx = fmax3(mx, fmax3(s[i][2][1], s[i][2][2], s[i][2][3]), fmax3(s[i][3][0], s[i][3][1], s[i][3][2]));
            mx = fmax2(mx, s[i][3][3]);
            mx = fmax2(mx, __shfl_xor(mx, 16, 64));                   // the four lanes of a row must end up with the same maximum
            mx = fmax2(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmax2(m_run[i], mx * c);               // running max in scaled (log2) units; c > 0
            const f32x2 nm = {-m_new, -m_new};
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float p[8];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x4 sv = s[i][2 * m + hh];
                    const f32x2 lo = f32x2{sv[0], sv[1]} * c2 + nm, hi = f32x2{sv[2], sv[3]} * c2 + nm;     // s == -inf -> exp2 -> 0
                    p[4 * hh + 0] = __builtin_amdgcn_exp2f(lo[0]);
                    p[4 * hh + 1] = __builtin_amdgcn_exp2f(lo[1]);
                    p[4 * hh + 2] = __builtin_amdgcn_exp2f(hi[0]);
                    p[4 * hh + 3] = __builtin_amdgcn_exp2f(hi[1]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[i][m][e] = f32_to_bf16(p[e]);
            }
            if (__any(m_new != m_run[i])) {                              // wave-uniform: most tiles do not raise any row's maximum
                const float alpha = __builtin_amdgcn_exp2f(m_run[i] - m_new);     // exp2(-inf) == 0 on the first tile
                l_acc[i] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o_acc[i][dt] *= alpha;
                m_run[i] = m_new;
            }
        }
        // O^T += V^T P^T; each V^T fragment is read from LDS once and used for all query tiles.  l += 1^T P^T on the matrix cores.
#pragma unroll
        for (int i = 0; i < QR; ++i) {
            f32x4 lt = {0, 0, 0, 0};                                     // every row of the ones-MFMA holds the column (query) sum
            mma32(lt, ones, pf[i][0]);
            mma32(lt, ones, pf[i][1]);
            l_acc[i] += lt[0];
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const bf16x8 v = load8(&Vs[buf][(dt * 16 + fr) * LD + m * 32 + fg * 8]);
#pragma unroll
                for (int i = 0; i < QR; ++i) mma32(o_acc[i][dt], v, pf[i][m]);
            }
    };

    // ---- the fast tile (see the header of this kernel) --------------------------------------------------------------------
    f32x4 nm_ref[QR];
    auto mask_tail = [&](f32x4 (&sv)[4], int key0, int i) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (key0 + kt * 16 + fg * 4 + r >= lim[i]) sv[kt][r] = -INFINITY;
    };
    auto tile_fast = [&](int buf, int key0, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        bf16x8 kf[4][2], vf[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if constexpr (LAB & 4) {
                kf[kt][0] = qf[kt % QR][0];
                kf[kt][1] = qf[kt % QR][1];
            } else {
                kf[kt][0] = load8(&Ks[buf][(kt * 16 + fr) * LD + fg * 8]);
                kf[kt][1] = load8(&Ks[buf][(kt * 16 + fr) * LD + 32 + fg * 8]);
            }
        }
        auto qk = [&](int i, f32x4 (&sv)[4]) {
            if constexpr ((LAB & 128) != 0) {
                // de-paired order: the four first halves, then the four second halves — no two neighbouring MFMAs share an accumulator (same sums, same order per accumulator)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    if constexpr (PRE) sv[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], qf[i][0], nm_ref[i], 0, 0, 0);
                    else sv[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], qf[i][0], f32x4{0, 0, 0, 0}, 0, 0, 0);
                }
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) mma32(sv[kt], kf[kt][1], qf[i][1]);
                return;
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                // the accumulator starts at -m_ref (loop-invariant registers): the scores arrive shifted, exp2 is all that is left
                if constexpr (LAB & 16) {
                    sv[kt] = nm_ref[i] + f32x4{(float)key0, 1.0f, 2.0f, (float)kt};
                } else {
                    if constexpr (PRE) sv[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], qf[i][0], nm_ref[i], 0, 0, 0);
                    else sv[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], qf[i][0], f32x4{0, 0, 0, 0}, 0, 0, 0);
                    mma32(sv[kt], kf[kt][1], qf[i][1]);
                }
            }
        };
        auto sm = [&](int i, f32x4 (&sv)[4], bf16x8 (&pf)[2]) {
            if constexpr (TAIL) mask_tail(sv, key0, i);
            if constexpr (LAB & 8) {                                      // (lab: the scores' raw bits as probabilities: no exp2, no conversion)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    union { f32x4 f; bf16x8 b; } u;
                    u.f = sv[2 * m] + sv[2 * m + 1];
                    pf[m] = u.b;
                }
                return;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        pf[m][4 * hh + r] = f32_to_bf16(__builtin_amdgcn_exp2f(PRE ? sv[2 * m + hh][r] : __builtin_fmaf(sv[2 * m + hh][r], c, nm_ref[i][0])));   // -inf -> 0
        };
        auto pv = [&](int i, const bf16x8 (&pf)[2]) {
            if constexpr ((LAB & 64) != 0) __builtin_amdgcn_s_setprio(1);
            if constexpr (LAB & 16) {                                     // (lab: keep the probabilities alive without the matrix cores)
                union { bf16x8 b; f32x4 f; } u0, u1;
                u0.b = pf[0];
                u1.b = pf[1];
                o_acc[i][0] += u0.f;
                o_acc[i][1] += u1.f;
                return;
            }
            f32x4 lt = {0, 0, 0, 0};
            mma32(lt, ones, pf[0]);
            mma32(lt, ones, pf[1]);
            l_acc[i] += lt[0];
            if constexpr ((LAB & 128) != 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) mma32(o_acc[i][dt], vf[dt][m], pf[m]);
            } else {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int m = 0; m < 2; ++m) mma32(o_acc[i][dt], vf[dt][m], pf[m]);
            }
            if constexpr ((LAB & 64) != 0) __builtin_amdgcn_s_setprio(0);
        };
        // MFMA : VALU interleave of one pipeline step (masks: 0x8 MFMA, 0x2 VALU; the transcendental ops count as VALU)
        auto pin = [&](auto NM, auto NV) {
            constexpr int nm = decltype(NM)::value, nv = decltype(NV)::value;
#pragma unroll
            for (int u = 0; u < nm; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, nv, 0);
            }
        };
        f32x4 sa[4], sb[4];
        bf16x8 pa[2], pb[2];
        static_assert(QR == 4 || QR == 2, "pipeline written for 2 or 4 query tiles per wave");
        qk(0, sa);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if constexpr (LAB & 4) vf[dt][m] = qf[dt % QR][m];
                else vf[dt][m] = load8(&Vs[buf][(dt * 16 + fr) * LD + m * 32 + fg * 8]);
            }
        // (Measured and dropped: __builtin_amdgcn_sched_barrier(0) between the pipeline steps.  The disassembly then shows the pinned
        // "MFMA, exp x 3" interleave with hardly any s_nop left, but 30 more scratch instructions (spills inside the loop) and the kernel is
        // SLOWER: T = 5632 generic form 369 vs 337 us, flow solve (pre-scaled form) 564 vs 431 ms.)
        if constexpr (QR == 4) {
            qk(1, sb); sm(0, sa, pa);
            pin(std::integral_constant<int, 8>{}, std::integral_constant<int, PRE ? 3 : 5>{});
            qk(2, sa); sm(1, sb, pb); pv(0, pa);
            pin(std::integral_constant<int, PRE ? 12 : 18>{}, std::integral_constant<int, 2>{});
            qk(3, sb); sm(2, sa, pa); pv(1, pb);
            pin(std::integral_constant<int, PRE ? 12 : 18>{}, std::integral_constant<int, 2>{});
            sm(3, sb, pb); pv(2, pa);
            pin(std::integral_constant<int, 10>{}, std::integral_constant<int, PRE ? 3 : 4>{});
            pv(3, pb);
        } else {
            qk(1, sb); sm(0, sa, pa);
            pin(std::integral_constant<int, 8>{}, std::integral_constant<int, PRE ? 3 : 5>{});
            sm(1, sb, pb); pv(0, pa);
            pin(std::integral_constant<int, 10>{}, std::integral_constant<int, PRE ? 3 : 4>{});
            pv(1, pb);
        }
    };

    const int n_tiles = (lim_hi + KT - 1) / KT;
    const int n_full = lim_lo / KT;                                    // the first loop holds unmasked tiles only
    auto run = [&](auto&& tf) {
        for (int it = 0; it < n_full; ++it) {
            const int buf = (LAB & 1) ? 0 : (it & 1), key0 = it * KT;
            const bool more = (it + 1) < n_tiles && !(LAB & 1);
            if (more) gload(key0 + KT);
            tf(buf, key0, std::false_type{});
            if (more) stash(buf ^ 1);
            if constexpr (!(LAB & 2)) __syncthreads();
        }
        // masked tiles: one (the ragged end) without a chunk mask, the tiles between the first and the last row's chunk end with one
        for (int it = n_full; it < n_tiles; ++it) {
            const int buf = it & 1, key0 = it * KT;
            const bool more = (it + 1) < n_tiles;
            if (more) gload(key0 + KT);
            tf(buf, key0, std::true_type{});
            if (more) stash(buf ^ 1);
            __syncthreads();
        }
    };
    if (n_tiles > 0) {
        gload(0);
        stash(0);
    }
    __syncthreads();
    if constexpr ((LAB & 32) != 0) {                              // (lab: every other workgroup starts a fraction of a tile later)
        if ((blockIdx.x + blockIdx.y + blockIdx.z) & 1) __builtin_amdgcn_s_sleep(24);
    }
    bool classical = !FAST;
    if constexpr (FAST) {
        if (ROT == 0 && n_tiles > 0) {
            // reference maximum of every row: its (masked) scores against the first key tile
#pragma unroll
            for (int i = 0; i < QR; ++i) {
                f32x4 sv[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    sv[kt] = f32x4{0, 0, 0, 0};
                    mma32(sv[kt], load8(&Ks[0][(kt * 16 + fr) * LD + fg * 8]), qf[i][0]);
                    mma32(sv[kt], load8(&Ks[0][(kt * 16 + fr) * LD + 32 + fg * 8]), qf[i][1]);
                }
                mask_tail(sv, 0, i);
                float mx = fmax3(fmax3(sv[0][0], sv[0][1], sv[0][2]), fmax3(sv[0][3], sv[1][0], sv[1][1]), fmax3(sv[1][2], sv[1][3], sv[2][0]));
                mx = fmax3(mx, fmax3(sv[2][1], sv[2][2], sv[2][3]), fmax3(sv[3][0], sv[3][1], sv[3][2]));
                mx = fmax2(mx, sv[3][3]);
                mx = fmax2(mx, __shfl_xor(mx, 16, 64));
                mx = fmax2(mx, __shfl_xor(mx, 32, 64));
                mx = -fmax2(mx * c, -1e30f);                                  // a row without a visible key in the tile: finite reference, p = 0
                nm_ref[i] = f32x4{mx, mx, mx, mx};
            }
        }
        if constexpr (ROT != 0) {
            // ---- the rotated loop (see the kernel header) ---------------------------------------------------------------------------------------
            bf16x8 kf[4][2], vf[4][2];
            f32x4 sa[4], sb[4];
            bf16x8 pa[2], pb[2];
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            // One pipeline step in a HAND-WRITTEN order (sched_barrier(0) freezes every slot; the packed conversions are pinned to their slot through an empty asm): 18 MFMAs —
            // QK(iq) pairs alternating with PV(ip) pairs, the two row-sum MFMAs last — with ONE exponential of SM(is) behind each of the first 16 and a packed conversion
            // behind every second one.  RF 1: the V^T fragments are re-filled from LDS buffer nbuf (step A), RF 2: the K fragments (step C) — two pairs behind their last use,
            // so that no MFMA is still reading the register the LDS data returns into.  Slot order M E M E C (M M E E C, no vector instruction between the two MFMAs of
            // an accumulator, measured 880 against 849 us).  The scores start from 0 (no per-row reference: softmax is shift-invariant and p = exp2(s) keeps bf16's relative precision at any magnitude;
            // a row whose p overflows fp32 or whose sum underflows to 0 sends the workgroup to the classical loop, see the vote below).
            auto step = [&](auto IQ, f32x4 (&sq)[4], auto IS, f32x4 (&ss)[4], bf16x8 (&po)[2], auto IP, const bf16x8 (&pi)[2], auto RF, int nbuf, int key0, auto tail_tag)
                            __attribute__((always_inline)) {
                constexpr int iq = decltype(IQ)::value, is = decltype(IS)::value, ip = decltype(IP)::value, rf = decltype(RF)::value;
                if constexpr (decltype(tail_tag)::value) mask_tail(ss, key0, is);
                __builtin_amdgcn_sched_barrier(0);
                float x[16];
                u32x4 pw[2];
                auto ex = [&](int k) __attribute__((always_inline)) {           // exponential k of the 16: probabilities (m, e) = (k / 8, k % 8) <- score tile 2 m + e / 4, row e % 4
                    x[k] = __builtin_amdgcn_exp2f(ss[2 * (k >> 3) + ((k & 7) >> 2)][k & 3]);       // -inf -> 0
                };
                auto cv = [&](int k) __attribute__((always_inline)) {           // k odd: the pair (k - 1, k) leaves as one packed conversion, in this slot
                    const bf16x2 h = {f32_to_bf16(x[k - 1]), f32_to_bf16(x[k])};
                    unsigned w = __builtin_bit_cast(unsigned, h);
                    asm volatile("" : "+v"(w));
                    pw[k >> 3][(k & 7) >> 1] = w;
                };
                // pair order: the row-sum pair first (its result is added to l at the END of the step: no MFMA -> VALU wait), then QK / PV pairs alternating — PV first in the
                // step that re-fills V^T, QK first otherwise, so that every re-fill sits one pair behind the last use of its registers and inside the step
                f32x4 lt = {0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    const int k0 = 2 * u, k1 = 2 * u + 1;
                    const int j = (u - 1) >> 1;                                  // fragment index of pair u >= 1
                    const bool is_pv = u >= 1 && (((u - 1) & 1) == (rf == 1 ? 0 : 1));
                    auto first = [&]() __attribute__((always_inline)) {
                        if (u == 0) mma32(lt, ones, pi[0]);
                        else if (is_pv) mma32(o_acc[ip][j], vf[j][0], pi[0]);
                        else sq[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j][0], qf[iq][0], f32x4{0, 0, 0, 0}, 0, 0, 0);
                    };
                    auto second = [&]() __attribute__((always_inline)) {
                        if (u == 0) mma32(lt, ones, pi[1]);
                        else if (is_pv) mma32(o_acc[ip][j], vf[j][1], pi[1]);
                        else mma32(sq[j], kf[j][1], qf[iq][1]);
                    };
                    auto refill = [&]() __attribute__((always_inline)) {        // behind pair u: the fragments pair u - 1 used last
                        if (u < 2 || (u & 1)) return;                            // (u = 2, 4, 6, 8: pair u - 1 is the re-filled kind's pair j' = u / 2 - 1)
                        const int jr = (u >> 1) - 1;
                        if (rf == 1) {
                            vf[jr][0] = load8(&Vs[nbuf][(jr * 16 + fr) * LD + fg * 8]);
                            vf[jr][1] = load8(&Vs[nbuf][(jr * 16 + fr) * LD + 32 + fg * 8]);
                        }
                        if (rf == 2) {
                            kf[jr][0] = load8(&Ks[nbuf][(jr * 16 + fr) * LD + fg * 8]);
                            kf[jr][1] = load8(&Ks[nbuf][(jr * 16 + fr) * LD + 32 + fg * 8]);
                        }
                    };
                    first();
                    if (k0 < 16) ex(k0);
                    __builtin_amdgcn_sched_barrier(0);
                    second();
                    if (k1 < 16) { ex(k1); cv(k1); }
                    refill();
                    __builtin_amdgcn_sched_barrier(0);
                }
                l_acc[ip] += lt[0];
                po[0] = __builtin_bit_cast(bf16x8, pw[0]);
                po[1] = __builtin_bit_cast(bf16x8, pw[1]);
            };
            typedef std::integral_constant<int, 0> I0;
            typedef std::integral_constant<int, 1> I1;
            typedef std::integral_constant<int, 2> I2;
            typedef std::integral_constant<int, 3> I3;
            auto body = [&](int it, auto tail_tag) __attribute__((always_inline)) {
                const int buf = it & 1, key0 = it * KT;
                // A: QK(1) | SM(0) | PV(3) of the tile before — its V^T fragments give way to this tile's
                step(I1{}, sb, I0{}, sa, pa, I3{}, pb, I1{}, buf, key0, tail_tag);
                // B: QK(2) | SM(1) | PV(0)
                step(I2{}, sa, I1{}, sb, pb, I0{}, pa, I0{}, buf, key0, tail_tag);
                // the next tile goes into the other LDS buffer (its last readers finished a tile ago: V^T in step A of tile t - 1, K in step C of tile t - 2)
                // (unconditional, so that the tile stays ONE basic block: past the last tile the registers hold a copy of an earlier tile)
                stash(buf ^ 1);
                __syncthreads();
                gload(min(key0 + 2 * KT, (n_tiles - 1) * KT));
                // C: QK(3) — its K fragments give way to the next tile's — | SM(2) | PV(1)
                step(I3{}, sb, I2{}, sa, pa, I1{}, pb, I2{}, buf ^ 1, key0, tail_tag);
                // D: QK(0) of the NEXT tile | SM(3) | PV(2) (past the last tile: scores of stale keys, never used)
                step(I0{}, sa, I3{}, sb, pb, I2{}, pa, I0{}, buf, key0, tail_tag);
            };
            if (n_tiles > 0) {
                if (n_tiles > 1) gload(KT);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    kf[kt][0] = load8(&Ks[0][(kt * 16 + fr) * LD + fg * 8]);
                    kf[kt][1] = load8(&Ks[0][(kt * 16 + fr) * LD + 32 + fg * 8]);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int m = 0; m < 2; ++m) vf[dt][m] = zero8<T>();
                pb[0] = zero8<T>();
                pb[1] = zero8<T>();
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {                                   // QK(0) of the first tile
                    sa[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], qf[0][0], f32x4{0, 0, 0, 0}, 0, 0, 0);
                    mma32(sa[kt], kf[kt][1], qf[0][1]);
                }
                for (int it = 0; it < n_full; ++it) body(it, std::false_type{});
                for (int it = n_full; it < n_tiles; ++it) body(it, std::true_type{});
                f32x4 lt = {0, 0, 0, 0};                                             // PV(3) of the last tile
                mma32(lt, ones, pb[0]);
                mma32(lt, ones, pb[1]);
                l_acc[3] += lt[0];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int m = 0; m < 2; ++m) mma32(o_acc[3][dt], vf[dt][m], pb[m]);
            }
        } else
        run(tile_fast);
        // overflow vote: any non-finite row sum / output (exponent bits all ones) sends the whole workgroup to the classical loop
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < QR; ++i) {
            bad |= ((__float_as_uint(l_acc[i]) & 0x7f800000u) == 0x7f800000u);
            // no reference: a row sum below 2^-80 means the row's largest p is that small, and probabilities below 2^-126 — within 2^-46 of it — were flushed to 0; such a
            // row (and one whose every p underflowed: l == 0) takes the classical loop too
            if constexpr (ROT != 0) bad |= ((__float_as_uint(l_acc[i]) & 0x7f800000u) < (47u << 23)) && (row0 + i * 16 + fr) < a.n_rows;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) bad |= ((__float_as_uint(o_acc[i][dt][r]) & 0x7f800000u) == 0x7f800000u);
        }
        classical = __syncthreads_or((int)bad) != 0;
        if constexpr (LAB != 0) classical = false;
        if (classical) {
#pragma unroll
            for (int i = 0; i < QR; ++i) {
                l_acc[i] = 0.0f;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o_acc[i][dt] = f32x4{0, 0, 0, 0};
            }
            if (n_tiles > 0) {
                gload(0);
                stash(0);
            }
            __syncthreads();
        }
    }
    if (classical) run(tile);
#pragma unroll
    for (int i = 0; i < QR; ++i) {
        const float l = l_acc[i];
        const int r = row0 + i * 16 + fr;
        if (r >= a.n_rows) continue;
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;
        T* op = reinterpret_cast<T*>(a.out) + (long long)b * a.o_bs + (long long)h * a.o_hs + (long long)r * a.o_lo;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = f32_to_bf16(o_acc[i][dt][e] * inv);
            *reinterpret_cast<bf16x4*>(op + dt * 16 + fg * 4) = o4;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// DiT attention on the 32x32x16 MFMA shape (round 6).  Same contract, same operand layouts and the same fixed-reference softmax as attn_dit_kernel above, another
// tile: v_mfma_f32_32x32x16_bf16 holds the matrix pipe 32 cycles per instruction (8 issue slots) instead of 16, so the exponentials of one score tile fit
// in the shadow of the MFMAs of its neighbours, and half as many MFMA instructions compete for issue slots (MI355X guide: <= 5 fillers per 32x32x16 gap).
//   S^T[key, q] = K Q^T   A = K rows (lane: key = l & 31, 8 consecutive d at 16 kk + 8 (l >> 5)), B = Q^T (lane: q = l & 31, the same d) — 4 MFMAs per 32 x 32 tile;
//                         C/D: lane (q = l & 31, hi = l >> 5), register r: key (r & 3) + 8 (r >> 2) + 4 hi — a lane holds 16 scores of ONE query.
//   O^T[d, q] += V^T P^T  the k-slot (hi, j) of a 16-key step is DEFINED as the key the lane already holds in registers 8 s + j of its score tile
//                         (keys {0-3, 8-11} + 4 hi of the step): P^T needs no cross-lane move at all, and V^T goes into LDS with the two middle 4-key groups of
//                         every 16 swapped, so that its fragment is one 16-byte read.  Row sums: fp32 adds per lane (the two lanes of a query meet once, at the end).
// A wave owns QT query tiles of 32 rows (QT = 4: 128 rows, ONE wave per SIMD and the whole 512-register file, every K / V^T fragment read from LDS feeds 4 MFMAs;
// QT = 2: two waves per SIMD).  Within a 64-key tile the work is cut into units of (32 keys) x (two query tiles) — 8 score MFMAs on two alternating accumulators,
// 32 exponentials + 16 conversions, 8 PV MFMAs on four accumulators — and software-pipelined: scores of unit u+1 | softmax of unit u | PV of unit u-1.
// Consecutive MFMAs never share an accumulator, so vector instructions may sit between any two of them (MI355X guide: an extra issue slot between two MFMAs on the
// SAME accumulator costs 43 cycles; that is what the 16x16x32 tile's dependent pairs ran into).
// ---------------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int QT, bool PRE, int SUMS = 1, int PIN = 0, int WPS = 2, int PIPE = 1, int REF = 1>     // REF 0: no per-row reference — p = exp2(s) as it is; a workgroup whose rows leave [2^-60, 2^127] takes the classical loop; PIPE 0: units one after the other (the other waves of the SIMD cover a wave's vector phases); QT query tiles of 32 rows per wave; SUMS: row sums by fp32 adds (0) or on the matrix cores (1);
                                                                         // PIN: vector instructions pinned behind every MFMA of a unit (0: the compiler's order); WPS: waves per SIMD the register budget is cut for
__global__ __launch_bounds__(256, WPS) void attn_dit32_kernel(AttnArgs a) {
    typedef bf16_t T;
    constexpr int KT = 64, LD = 72;            // keys per tile; LDS row stride (144 B: the 16 rows of a ds_read_b128 lane group fall on 16 distinct 16-byte slots)
    __shared__ __attribute__((aligned(16))) T Ks[2][KT * LD];
    __shared__ __attribute__((aligned(16))) T Vs[2][64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fq = lane & 31, hi = lane >> 5;
    int b = blockIdx.z, h = blockIdx.y, bx = blockIdx.x;
    {   // XCD-aware order (see attn_dit_kernel): XCD e walks the row blocks of heads e, e + 8, ... one head after the other
        const int nx = gridDim.x, nh = gridDim.y * gridDim.z;
        if ((nh & 7) == 0) {
            const int L = blockIdx.x + nx * (blockIdx.y + gridDim.y * blockIdx.z);
            const int j = L >> 3, hb = (j / nx) * 8 + (L & 7);
            bx = j % nx;
            h = hb % gridDim.y;
            b = hb / gridDim.y;
        }
    }
    constexpr int WG_ROWS = 4 * QT * 32;
    const int row0 = bx * WG_ROWS + wave * (QT * 32);
    const int kv_len = a.kv_len ? a.kv_len[b] : a.kv_len_const;
    const T* __restrict__ kb = reinterpret_cast<const T*>(a.k) + (long long)b * a.k_bs + (long long)h * a.k_hs;
    const T* __restrict__ vb = reinterpret_cast<const T*>(a.vT) + (long long)b * a.v_bs + (long long)h * a.v_hs;

    bf16x8 qf[QT][4];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const int r = row0 + i * 32 + fq;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            qf[i][kk] = r < a.n_rows ? load8(reinterpret_cast<const T*>(a.q) + (long long)b * a.q_bs + (long long)h * a.q_hs + (long long)r * a.q_lo + kk * 16 + hi * 8) : zero8<T>();
    }
    // tile loader (the 4 waves together): thread t moves 2 x 16 B of K (rows t / 8 and t / 8 + 32, chunk t % 8) and 2 x 16 B of V^T
    const int lrow = tid >> 3, lchunk = (tid & 7) * 8;
    bf16x8 rk[2], rv[2];
    auto gload = [&](int key0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int key = key0 + lrow + i * 32;
            key = key < kv_len ? key : kv_len - 1;                       // clamped rows are masked in the tail tile
            rk[i] = load8(kb + (long long)key * 64 + lchunk);
            rv[i] = load8(vb + (long long)(lrow + i * 32) * a.v_ld + key0 + lchunk);
        }
    };
    // V^T in LDS: within every 16 keys the 4-key groups go in the order 0, 2, 1, 3 — the k-slots of a PV step as the score tile's registers define them
    const int vc = tid & 7;                                              // the thread's 8-key chunk = groups 2 (vc & 1), 2 (vc & 1) + 1 of 16-key block vc / 2
    const int vpos = (vc >> 1) * 16 + (vc & 1) * 4;
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            store8(&Ks[buf][(lrow + i * 32) * LD + lchunk], rk[i]);
            bf16x4 lo, hv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = rv[i][j];
                hv[j] = rv[i][4 + j];
            }
            *reinterpret_cast<bf16x4*>(&Vs[buf][(lrow + i * 32) * LD + vpos]) = lo;
            *reinterpret_cast<bf16x4*>(&Vs[buf][(lrow + i * 32) * LD + vpos + 8]) = hv;
        }
    };

    f32x16 o_acc[QT][2];
    float l_acc[QT];
    auto zero16f = [] {                                                  // (a literal every time: a zero vector kept in registers across the key loop would cost 16 of them)
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
        return z;
    };
#define zero16 zero16f()
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        l_acc[i] = 0.0f;
        o_acc[i][0] = zero16;
        o_acc[i][1] = zero16;
    }
    const float c = PRE ? 1.0f : a.scale * 1.4426950408889634f;
    f32x16 nm_ref[QT];                                                   // -m_ref of the lane's query in every register: the score accumulators start there

    auto kfrag = [&](int buf, int kh, bf16x8 (&kf)[4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kf[kk] = load8(&Ks[buf][(kh * 32 + fq) * LD + kk * 16 + hi * 8]);
    };
    auto vfrag = [&](int buf, int kh, bf16x8 (&vf)[2][2]) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int s = 0; s < 2; ++s) vf[dt][s] = load8(&Vs[buf][(dt * 32 + fq) * LD + (kh * 2 + s) * 16 + hi * 8]);
    };
    auto qk0 = [&](int i, const bf16x8 (&kf)[4], f32x16& s) {           // scores of one query tile against 32 keys, from zero
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[i][kk], kk == 0 ? zero16 : s, 0, 0, 0);
    };
    auto mask = [&](f32x16& s, int key0) {                               // keys at or beyond the sequence end (tail tile only)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (key0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= kv_len) s[r] = -INFINITY;
    };
    auto pv1 = [&](int i, const bf16x8 (&vf)[2][2], const bf16x8 (&pf)[2]) {
#pragma unroll
        for (int st = 0; st < 4; ++st) o_acc[i][st & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st & 1][st >> 1], pf[st >> 1], o_acc[i][st & 1], 0, 0, 0);
    };

    // ---- the fast tile: units of (32 keys) x (ONE query tile), pipelined  QK(u + 1) | SM(u) | PV(u - 1); the 4 dependent score MFMAs of unit u + 1 alternate with
    //      the 4 PV MFMAs of unit u - 1 (two accumulators, twice each), so no two neighbours share an accumulator ----------------------------------------------
    bf16x8 ones_sel;                                                     // A operand of the row-sum MFMA (16x16x32): row 0 sums the k-groups 0 and 2 (the two lanes of query n), row 1 the groups 1 and 3 (query n + 16)
    {
        const bool on = ((lane & 15) == 0 && ((lane >> 4) & 1) == 0) || ((lane & 15) == 1 && ((lane >> 4) & 1) == 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) ones_sel[e] = f32_to_bf16(on ? 1.0f : 0.0f);
    }
    f32x4 l_mm[QT];
#pragma unroll
    for (int i = 0; i < QT; ++i) l_mm[i] = f32x4{0, 0, 0, 0};
    auto sm1 = [&](int i, f32x16& s, bf16x8 (&pf)[2], int key0, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        if constexpr (TAIL) mask(s, key0);
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(PRE ? s[r] : (REF ? __builtin_fmaf(s[r], c, nm_ref[i][0]) : s[r] * c));      // -inf -> 0
            if constexpr (SUMS == 0) sum += e;
            pf[r >> 3][r & 7] = f32_to_bf16(e);
        }
        if constexpr (SUMS == 0) l_acc[i] += sum;
    };
    auto tile_fast = [&](int buf, int key0, auto tail_tag) {
        constexpr int NU = 2 * QT;                                       // units: u = kh * QT + i
        bf16x8 kf[4], vf[2][2];
        f32x16 sa, sb;
        bf16x8 pa[2], pb[2];
        kfrag(buf, 0, kf);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[0][kk], kk == 0 ? ((PRE && REF) ? nm_ref[0] : zero16) : sa, 0, 0, 0);
        vfrag(buf, 0, vf);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int kh = u / QT, i = u % QT;
            f32x16& s_cur = (u & 1) ? sb : sa;
            f32x16& s_nxt = (u & 1) ? sa : sb;
            bf16x8 (&p_cur)[2] = (u & 1) ? pb : pa;
            bf16x8 (&p_prv)[2] = (u & 1) ? pa : pb;
            const bool has_qk = u + 1 < NU, has_pv = u > 0;
            const int in = (u + 1) % QT, ip = (u + QT - 1) % QT;         // query tiles of units u + 1 / u - 1
            if (has_qk && in == 0) kfrag(buf, (u + 1) / QT, kf);         // the next 32 keys
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                if (has_qk) s_nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[st], qf[in][st], st == 0 ? ((PRE && REF) ? nm_ref[in] : zero16) : s_nxt, 0, 0, 0);
                if (has_pv) o_acc[ip][st & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st & 1][st >> 1], p_prv[st >> 1], o_acc[ip][st & 1], 0, 0, 0);
            }
            if (has_pv && i == 0 && kh == 1) vfrag(buf, 1, vf);          // unit u - 1 was the last of key half 0: its PV (just issued) was the last reader of that half's V^T fragments
            sm1(i, s_cur, p_cur, key0 + kh * 32, tail_tag);
            if constexpr (SUMS == 1) {
                l_mm[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_sel, p_cur[0], l_mm[i], 0, 0, 0);
                l_mm[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_sel, p_cur[1], l_mm[i], 0, 0, 0);
            }
            if constexpr (PIN > 0) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, PIN, 0);
                }
            }
            if constexpr (PIN < 0) __builtin_amdgcn_sched_barrier(0);    // units are scheduling regions: nothing moves across (keeps the fragments of later units out of the registers)
        }
        if (QT == 1) vfrag(buf, 1, vf);                                  // (QT = 1: unit 1 = key half 1 has no later slot that loads its fragments)
        pv1(QT - 1, vf, ((NU - 1) & 1) ? pb : pa);                       // PV of the last unit
    };
    // ---- the same units one after the other (PIPE = 0): scores of all the wave's query tiles against 32 keys (independent accumulators alternate), their
    //      probabilities, their PV — no second score / probability buffer, so three waves fit a SIMD at QT = 1 and cover each other's vector phases ------------------
    auto tile_seq = [&](int buf, int key0, auto tail_tag) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            bf16x8 kf[4], vf[2][2];
            f32x16 s[QT];
            bf16x8 pf[QT][2];
            kfrag(buf, kh, kf);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < QT; ++i) s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[i][kk], kk == 0 ? ((PRE && REF) ? nm_ref[i] : zero16) : s[i], 0, 0, 0);
            vfrag(buf, kh, vf);
#pragma unroll
            for (int i = 0; i < QT; ++i) sm1(i, s[i], pf[i], key0 + kh * 32, tail_tag);
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int i = 0; i < QT; ++i) o_acc[i][st & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st & 1][st >> 1], pf[i][st >> 1], o_acc[i][st & 1], 0, 0, 0);
            if constexpr (SUMS == 1) {
#pragma unroll
                for (int i = 0; i < QT; ++i) {
                    l_mm[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_sel, pf[i][0], l_mm[i], 0, 0, 0);
                    l_mm[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_sel, pf[i][1], l_mm[i], 0, 0, 0);
                }
            }
        }
    };
    // ---- classical online-softmax tile (the fall-back when a score outgrew its reference by 2^127), one query tile at a time ------------------------------
    float m_run[QT];
    auto tile_classical = [&](int buf, int key0, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            bf16x8 kf[4], vf[2][2];
            f32x16 s[2];                                                 // the two key halves
            bf16x8 pf[2][2];
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                kfrag(buf, kh, kf);
                qk0(i, kf, s[kh]);
                if constexpr (TAIL) mask(s[kh], key0 + kh * 32);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmax2(mx, s[kh][r]);
            mx = fmax2(mx, __shfl_xor(mx, 32, 64));                      // the two lanes of a query
            const float m_new = fmax2(m_run[i], mx * c);
            const float alpha = m_new == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f(m_run[i] - m_new);      // exp2(-inf) == 0 on the first tile
            float sum = 0.0f;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = m_new == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(__builtin_fmaf(s[kh][r], c, -m_new));
                    sum += e;
                    pf[kh][r >> 3][r & 7] = f32_to_bf16(e);
                }
            l_acc[i] = l_acc[i] * alpha + sum;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[i][dt][r] *= alpha;
            m_run[i] = m_new;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                vfrag(buf, kh, vf);
                pv1(i, vf, pf[kh]);
            }
        }
    };

    const int n_tiles = (kv_len + KT - 1) / KT, n_full = kv_len / KT;
    auto run = [&](auto&& tf) {
        for (int it = 0; it < n_tiles; ++it) {
            const int buf = it & 1, key0 = it * KT;
            const bool more = (it + 1) < n_tiles;
            if (more) gload(key0 + KT);
            if (it < n_full) tf(buf, key0, std::false_type{});
            else tf(buf, key0, std::true_type{});
            if (more) stash(buf ^ 1);
            __syncthreads();
        }
    };
    if (n_tiles > 0) {
        gload(0);
        stash(0);
    }
    __syncthreads();
    bool classical = false;
    if (n_tiles > 0) {
        // reference maximum of every row: its (masked) scores against the first key tile
#pragma unroll
        for (int i = 0; i < (REF ? QT : 0); ++i) {
            float mx = -INFINITY;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                bf16x8 kf[4];
                f32x16 s;
                kfrag(0, kh, kf);
                qk0(i, kf, s);
                mask(s, kh * 32);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmax2(mx, s[r]);
            }
            float m = fmax2(mx, __shfl_xor(mx, 32, 64));
            m = -fmax2(m * c, -1e30f);                                       // a row without a visible key in the tile: finite reference, p = 0
#pragma unroll
            for (int r = 0; r < 16; ++r) nm_ref[i][r] = m;
            if constexpr (PRE) asm volatile("" : "+v"(nm_ref[i]));          // 16 REGISTERS from here on: the compiler must not rebuild the splat in front of every score tile
        }
        if constexpr (PIPE == 1) run(tile_fast);
        else run(tile_seq);
        if constexpr (SUMS == 1) {
            // row sums off the matrix cores: query n < 16 in lane n, register 0; query n + 16 in lane n, register 1 (both lanes of a query already added)
#pragma unroll
            for (int i = 0; i < QT; ++i) {
                const float v0 = __shfl(l_mm[i][0], lane & 15, 64), v1 = __shfl(l_mm[i][1], lane & 15, 64);
                l_acc[i] = 0.5f * ((fq & 16) ? v1 : v0);                      // (halved: the epilogue adds the lane pair)
            }
        }
        // overflow vote: any non-finite row sum / output sends the whole workgroup to the classical loop
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            bad |= ((__float_as_uint(l_acc[i]) & 0x7f800000u) == 0x7f800000u);
            if constexpr (!REF) {
                const float lq = l_acc[i] + __shfl_xor(l_acc[i], 32, 64);
                bad |= (row0 + i * 32 + fq < a.n_rows) && !(lq >= 8.7e-19f);            // a row whose probabilities sank below 2^-60: not enough range left
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) bad |= ((__float_as_uint(o_acc[i][dt][r]) & 0x7f800000u) == 0x7f800000u);
        }
        classical = __syncthreads_or((int)bad) != 0;
        if (classical) {
#pragma unroll
            for (int i = 0; i < QT; ++i) {
                l_acc[i] = 0.0f;
                o_acc[i][0] = zero16;
                o_acc[i][1] = zero16;
            }
#pragma unroll
            for (int i = 0; i < QT; ++i) m_run[i] = -INFINITY;
            gload(0);
            stash(0);
            __syncthreads();
            run(tile_classical);
        }
    }
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const float l = l_acc[i] + __shfl_xor(l_acc[i], 32, 64);             // the two lanes of a query hold the sums of complementary key sets
        const int r = row0 + i * 32 + fq;
        if (r >= a.n_rows) continue;
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;
        T* op = reinterpret_cast<T*>(a.out) + (long long)b * a.o_bs + (long long)h * a.o_hs + (long long)r * a.o_lo;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = f32_to_bf16(o_acc[i][dt][4 * g + e] * inv);
                *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g + 4 * hi) = o4;
            }
    }
}
#undef zero16

#ifdef HVX_LAB      // (lab builds only: measured slower than the product tile, profiles/r06_attn_tile_ab.md; the hazards of its asm MFMAs are met by construction, not by the compiler)
// ---------------------------------------------------------------------------------------------------------------------
// DiT attention, ONE wave per SIMD with the whole 512-register file (round 6; the structure of the guide's fastest attention, at head_dim 64): a wave owns
// FOUR query tiles of 32 rows (128 rows; 512 per workgroup), so every K / V^T fragment read from LDS feeds 4 MFMAs, and keeps O^T (128 registers) and Q (64)
// in the ACCUMULATOR half of the file, where only MFMAs touch them; the scores start from 0
// in VGPRs and the row's reference is ONE v_add per score on the way into the exponential (a 16-register start-value splat per query tile does not stay resident:
// hipcc copies it into place in front of every use).  With a 512-register budget hipcc selects the AGPR-destination MFMA forms for its builtins and
// copies every score out of the accumulator file (profiles/r06_attn_tile_ab.md), so the MFMAs here are inline asm with the register class of every operand spelled
// out: O^T / row sums in AGPRs, scores and the K / V^T / P / Q fragments in VGPRs.  asm volatile statements keep their order; the compiler places the vector
// instructions and the LDS reads between them and allocates the registers.  What the compiler does NOT do for an asm MFMA (cdna guide §5.7) is done by construction:
//   * an MFMA result is read by a vector instruction only in the NEXT unit's region (sched_barrier between units), behind >= 2 further MFMAs (>= 64 cycles);
//   * the P operand of a PV / row-sum MFMA was written by the vector ALU in the PREVIOUS region, a QK MFMA ahead of its first reader (no s_nop needed);
//   * dependent MFMAs are either adjacent (accumulate chain: forwarded) or one independent MFMA (32 cycles) apart.
// Same tile mathematics as attn_dit32_kernel (units of 32 keys x one query tile, QK(u + 1) | SM(u) | PV(u - 1), fixed-reference softmax, row sums on a selector MFMA).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void a4_qk_first0(f32x16& d, const bf16x8& k, const bf16x8& q) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(k), "v"(q));
}
__device__ __forceinline__ void a4_qk(f32x16& d, const bf16x8& k, const bf16x8& q) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(k), "v"(q));
}
__device__ __forceinline__ void a4_pv(f32x16& o, const bf16x8& v, const bf16x8& p) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(p));
}
__device__ __forceinline__ void a4_sum(f32x4& l, const bf16x8& ones, const bf16x8& p) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(l) : "v"(ones), "v"(p));
}

template <bool PRE>
__global__ __launch_bounds__(256, 1) void attn_dit_a4_kernel(AttnArgs a) {
    typedef bf16_t T;
    constexpr int QT = 4, KT = 64, LD = 72;
    __shared__ __attribute__((aligned(16))) T Ks[2][KT * LD];
    __shared__ __attribute__((aligned(16))) T Vs[2][64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fq = lane & 31, hi = lane >> 5;
    int b = blockIdx.z, h = blockIdx.y, bx = blockIdx.x;
    {   // XCD-aware order (see attn_dit_kernel)
        const int nx = gridDim.x, nh = gridDim.y * gridDim.z;
        if ((nh & 7) == 0) {
            const int L = blockIdx.x + nx * (blockIdx.y + gridDim.y * blockIdx.z);
            const int j = L >> 3, hb = (j / nx) * 8 + (L & 7);
            bx = j % nx;
            h = hb % gridDim.y;
            b = hb / gridDim.y;
        }
    }
    constexpr int WG_ROWS = 4 * QT * 32;
    const int row0 = bx * WG_ROWS + wave * (QT * 32);
    const int kv_len = a.kv_len ? a.kv_len[b] : a.kv_len_const;
    const T* __restrict__ kb = reinterpret_cast<const T*>(a.k) + (long long)b * a.k_bs + (long long)h * a.k_hs;
    const T* __restrict__ vb = reinterpret_cast<const T*>(a.vT) + (long long)b * a.v_bs + (long long)h * a.v_hs;

    bf16x8 qf[QT][4];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const int r = row0 + i * 32 + fq;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            qf[i][kk] = r < a.n_rows ? load8(reinterpret_cast<const T*>(a.q) + (long long)b * a.q_bs + (long long)h * a.q_hs + (long long)r * a.q_lo + kk * 16 + hi * 8) : zero8<T>();
    }
    const int lrow = tid >> 3, lchunk = (tid & 7) * 8;
    bf16x8 rk[2], rv[2];
    auto gload = [&](int key0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int key = key0 + lrow + i * 32;
            key = key < kv_len ? key : kv_len - 1;
            rk[i] = load8(kb + (long long)key * 64 + lchunk);
            rv[i] = load8(vb + (long long)(lrow + i * 32) * a.v_ld + key0 + lchunk);
        }
    };
    const int vc = tid & 7;
    const int vpos = (vc >> 1) * 16 + (vc & 1) * 4;
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            store8(&Ks[buf][(lrow + i * 32) * LD + lchunk], rk[i]);
            bf16x4 lo, hv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = rv[i][j];
                hv[j] = rv[i][4 + j];
            }
            *reinterpret_cast<bf16x4*>(&Vs[buf][(lrow + i * 32) * LD + vpos]) = lo;
            *reinterpret_cast<bf16x4*>(&Vs[buf][(lrow + i * 32) * LD + vpos + 8]) = hv;
        }
    };
    auto zero16f = [] {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
        return z;
    };
    f32x16 o_acc[QT][2];
    f32x4 l_mm[QT];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        o_acc[i][0] = zero16f();
        o_acc[i][1] = zero16f();
        l_mm[i] = f32x4{0, 0, 0, 0};
    }
    const float c = PRE ? 1.0f : a.scale * 1.4426950408889634f;
    float nm_ref[QT];                                                    // -m_ref of the lane's query: one register per query tile, subtracted on the way into the exponential
    bf16x8 ones_sel;
    {
        const bool on = ((lane & 15) == 0 && ((lane >> 4) & 1) == 0) || ((lane & 15) == 1 && ((lane >> 4) & 1) == 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) ones_sel[e] = f32_to_bf16(on ? 1.0f : 0.0f);
    }
    auto kfrag = [&](int buf, int kh, bf16x8 (&kf)[4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kf[kk] = load8(&Ks[buf][(kh * 32 + fq) * LD + kk * 16 + hi * 8]);
    };
    auto vfrag = [&](int buf, int kh, bf16x8 (&vf)[2][2]) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int s = 0; s < 2; ++s) vf[dt][s] = load8(&Vs[buf][(dt * 32 + fq) * LD + (kh * 2 + s) * 16 + hi * 8]);
    };
    auto mask = [&](f32x16& s, int key0) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (key0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= kv_len) s[r] = -INFINITY;
    };
    // four scores of a unit -> probabilities (slice q = registers 4 q .. 4 q + 3 of the score tile): the vector work is cut into slices so that it can be PINNED between
    // the MFMAs.  The compiler sees an asm MFMA as an instruction that takes no time and gathers all vector work in front of the MFMA run (nothing overlaps: 1137 us, and
    // the first slice then reads scores its MFMA has not finished: wrong results); slices pinned behind PAIRS of MFMAs: 1103 us (this form); one 5-instruction slice behind
    // EVERY MFMA — which puts vector instructions between MFMAs on the same accumulator: 2990 us (the guide's 43-cycle cliff, three times per unit).
    auto sm_slice = [&](int i, int q, f32x16& s, bf16x8 (&pf)[2], int key0, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
        for (int r = 4 * q; r < 4 * q + 4; ++r) {
            float x = s[r];
            if constexpr (TAIL)
                if (key0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= kv_len) x = -INFINITY;
            const float e = __builtin_amdgcn_exp2f(PRE ? x + nm_ref[i] : __builtin_fmaf(x, c, nm_ref[i]));
            pf[r >> 3][r & 7] = f32_to_bf16(e);
        }
    };
    // one 64-key tile: 8 units u = kh * 4 + i.  Region of unit u:  Q0 P0 Q1 P1 | slice 0 | Q2 P2 | slice 1 | Q3 P3 | slice 2 | sum sum | slice 3   (Q = score MFMA of unit
    // u + 1, P = PV MFMA of unit u - 1, slices = softmax of unit u; every `|` is a sched_barrier).  The first slice sits behind >= 2 MFMAs: the scores it reads were
    // finished by the previous region's Q3 (>= 12 wait states ago in every case, the tile's first region included).
    auto tile_fast = [&](int buf, int key0, auto tail_tag) {
        constexpr int NU = 2 * QT;
        bf16x8 kf[4], vf[2][2];
        f32x16 sa, sb;
        bf16x8 pa[2], pb[2];
        kfrag(buf, 0, kf);
        vfrag(buf, 0, vf);
        a4_qk_first0(sa, kf[0], qf[0][0]);
#pragma unroll
        for (int kk = 1; kk < 4; ++kk) a4_qk(sa, kf[kk], qf[0][kk]);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            __builtin_amdgcn_sched_barrier(0);
            const int kh = u / QT, i = u % QT;
            f32x16& s_cur = (u & 1) ? sb : sa;
            f32x16& s_nxt = (u & 1) ? sa : sb;
            bf16x8 (&p_cur)[2] = (u & 1) ? pb : pa;
            bf16x8 (&p_prv)[2] = (u & 1) ? pa : pb;
            const bool has_qk = u + 1 < NU, has_pv = u > 0;
            const int in = (u + 1) % QT, ip = (u + QT - 1) % QT;
            if (has_qk && in == 0) kfrag(buf, (u + 1) / QT, kf);
            auto pair = [&](int st) __attribute__((always_inline)) {
                if (has_qk) {
                    if (st == 0) a4_qk_first0(s_nxt, kf[0], qf[in][0]);
                    else a4_qk(s_nxt, kf[st], qf[in][st]);
                }
                if (has_pv) a4_pv(o_acc[ip][st & 1], vf[st & 1][st >> 1], p_prv[st >> 1]);
            };
            pair(0);
            pair(1);
            __builtin_amdgcn_sched_barrier(0);
            sm_slice(i, 0, s_cur, p_cur, key0 + kh * 32, tail_tag);
            __builtin_amdgcn_sched_barrier(0);
            pair(2);
            __builtin_amdgcn_sched_barrier(0);
            sm_slice(i, 1, s_cur, p_cur, key0 + kh * 32, tail_tag);
            __builtin_amdgcn_sched_barrier(0);
            pair(3);
            __builtin_amdgcn_sched_barrier(0);
            sm_slice(i, 2, s_cur, p_cur, key0 + kh * 32, tail_tag);
            __builtin_amdgcn_sched_barrier(0);
            if (has_pv) {
                a4_sum(l_mm[ip], ones_sel, p_prv[0]);
                a4_sum(l_mm[ip], ones_sel, p_prv[1]);
            }
            if (has_pv && i == 0 && kh == 1) vfrag(buf, 1, vf);
            __builtin_amdgcn_sched_barrier(0);
            sm_slice(i, 3, s_cur, p_cur, key0 + kh * 32, tail_tag);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            bf16x8 (&p_lst)[2] = ((NU - 1) & 1) ? pb : pa;
#pragma unroll
            for (int st = 0; st < 4; ++st) a4_pv(o_acc[QT - 1][st & 1], vf[st & 1][st >> 1], p_lst[st >> 1]);
            a4_sum(l_mm[QT - 1], ones_sel, p_lst[0]);
            a4_sum(l_mm[QT - 1], ones_sel, p_lst[1]);
        }
    };
    // classical online-softmax tile (fall-back), one query tile at a time; the compiler moves O^T between the register halves for the rescale
    float m_run[QT], l_acc[QT];
    auto tile_classical = [&](int buf, int key0, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            bf16x8 kf[4], vf[2][2];
            f32x16 s[2];
            bf16x8 pf[2][2];
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                kfrag(buf, kh, kf);
                a4_qk_first0(s[kh], kf[0], qf[i][0]);
#pragma unroll
                for (int kk = 1; kk < 4; ++kk) a4_qk(s[kh], kf[kk], qf[i][kk]);
            }
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");             // the last chain's result: 16 states before the vector ALU reads it
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (TAIL) {
                mask(s[0], key0);
                mask(s[1], key0 + 32);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmax2(mx, s[kh][r]);
            mx = fmax2(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmax2(m_run[i], mx * c);
            const float alpha = m_new == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f(m_run[i] - m_new);
            float sum = 0.0f;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = m_new == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(__builtin_fmaf(s[kh][r], c, -m_new));
                    sum += e;
                    pf[kh][r >> 3][r & 7] = f32_to_bf16(e);
                }
            l_acc[i] = l_acc[i] * alpha + sum;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[i][dt][r] *= alpha;
            m_run[i] = m_new;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                vfrag(buf, kh, vf);
#pragma unroll
                for (int st = 0; st < 4; ++st) a4_pv(o_acc[i][st & 1], vf[st & 1][st >> 1], pf[kh][st >> 1]);
            }
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int n_tiles = (kv_len + KT - 1) / KT, n_full = kv_len / KT;
    auto run = [&](auto&& tf) {
        for (int it = 0; it < n_full; ++it) {                             // (two loops, not one with a branch: the register assignment of O^T then never has to be shuffled between two loop bodies)
            const int buf = it & 1, key0 = it * KT;
            const bool more = (it + 1) < n_tiles;
            if (more) gload(key0 + KT);
            tf(buf, key0, std::false_type{});
            if (more) stash(buf ^ 1);
            __syncthreads();
        }
        for (int it = n_full; it < n_tiles; ++it) {                       // the ragged last tile
            const int buf = it & 1, key0 = it * KT;
            tf(buf, key0, std::true_type{});
            __syncthreads();
        }
    };
    if (n_tiles > 0) {
        gload(0);
        stash(0);
    }
    __syncthreads();
    bool classical = false;
    if (n_tiles > 0) {
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            float mx = -INFINITY;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                bf16x8 kf[4];
                f32x16 s;
                kfrag(0, kh, kf);
                a4_qk_first0(s, kf[0], qf[i][0]);
#pragma unroll
                for (int kk = 1; kk < 4; ++kk) a4_qk(s, kf[kk], qf[i][kk]);
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                mask(s, kh * 32);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmax2(mx, s[r]);
            }
            float m = fmax2(mx, __shfl_xor(mx, 32, 64));
            m = -fmax2(m * c, -1e30f);
            nm_ref[i] = m;
        }
        run(tile_fast);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");                 // the last MFMAs' results before anything reads O^T / the row sums
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            const float v0 = __shfl(l_mm[i][0], lane & 15, 64), v1 = __shfl(l_mm[i][1], lane & 15, 64);
            l_acc[i] = 0.5f * ((fq & 16) ? v1 : v0);
        }
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            bad |= ((__float_as_uint(l_acc[i]) & 0x7f800000u) == 0x7f800000u);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) bad |= ((__float_as_uint(o_acc[i][dt][r]) & 0x7f800000u) == 0x7f800000u);
        }
        classical = __syncthreads_or((int)bad) != 0;
        if (classical) {
#pragma unroll
            for (int i = 0; i < QT; ++i) {
                l_acc[i] = 0.0f;
                m_run[i] = -INFINITY;
                o_acc[i][0] = zero16f();
                o_acc[i][1] = zero16f();
            }
            gload(0);
            stash(0);
            __syncthreads();
            run(tile_classical);
        }
    } else {
#pragma unroll
        for (int i = 0; i < QT; ++i) l_acc[i] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const float l = l_acc[i] + __shfl_xor(l_acc[i], 32, 64);
        const int r = row0 + i * 32 + fq;
        if (r >= a.n_rows) continue;
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;
        T* op = reinterpret_cast<T*>(a.out) + (long long)b * a.o_bs + (long long)h * a.o_hs + (long long)r * a.o_lo;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = f32_to_bf16(o_acc[i][dt][4 * g + e] * inv);
                *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g + 4 * hi) = o4;
            }
    }
}
#endif  // HVX_LAB

template <class T>
static int launch_t(const AttnArgs& a_in, hipStream_t s) {
    AttnArgs a = a_in;
    if (sizeof(T) == 2 && !a.causal && a.n_splits == 1 && a.kn >= a.n_rows && !a.kv_slot && !a.n_valid_lo && a.n_rows >= 256 &&
        (a.v_ld & 63) == 0) {
        const double fl = 4.0 * a.n_rows * (double)a.kv_len_const * 64.0 * a.heads * a.batch;
        const int slot = prof_begin(PK_ATTN, a.kv_len ? 0.0 : fl, s);
        // Query rows per wave: 32 (QR = 2, three waves per SIMD) or 64 (QR = 4, 254 VGPRs, two waves per SIMD, half the K / V^T staging
        // traffic per flop).  Alone the two tie on long sequences (415 vs 419 us at T = 5632) and QR = 2 wins on short ones (45 vs 53 us at
        // T = 1408); beside the decode stream of the pipelined synthesis QR = 4 leaves room on every SIMD and halves the L2 traffic:
        // the decode step next to it takes 2.6 ms instead of 4.1 ms (tools/contention_probe.py).
        // With a chunk mask the last workgroup of a head does twice the average work: 128-row workgroups (twice as many, three per SIMD)
        // balance better than 256-row ones (T = 5632, chunk 50: 244 vs 288 us; the unmasked pass takes 392 us).
        // (Measured and dropped: giving the rows of the last, partial round of 256-row workgroups to a second launch of 128-row workgroups —
        // T = 5632, 704 workgroups: 364 vs 337 us; 2816 workgroups: flow solve 441 vs 431 ms.  The rounds are not uniform enough for the
        // tail to be worth a kernel boundary.)
        const dim3 g4((a.n_rows + 255) / 256, a.heads, a.batch), g2((a.n_rows + 127) / 128, a.heads, a.batch);
#ifdef HVX_LAB
        // (NW = 8, measured on MI355X round 5: 1158 vs 1055 us at B = 8, 349 vs 275 us at B = 2 — the barrier over eight waves costs more than the halved
        // staging saves; two independent 4-wave workgroups per CU drift apart and cover each other.  Lab builds only.)
        const int nw8 = opt(OPT_ATTN_NW) == 8;
        if (nw8 && a.n_rows >= 2048 && a.chunk <= 0 && a.q_log2) {
            const dim3 g8((a.n_rows + 511) / 512, a.heads, a.batch);
            hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 0, 8>), g8, dim3(512), 0, s, a);
            prof_end(slot, s);
            return hipGetLastError() == hipSuccess ? 0 : (set_error("attention launch failed"), -1);
        }
        const int lab = (int)opt(OPT_ATTN_LAB);
        if (lab && a.n_rows >= 2048 && a.chunk <= 0 && a.q_log2) {
            switch (lab) {
                case 1: hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 1>), g4, dim3(256), 0, s, a); break;
                case 2: hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 2>), g4, dim3(256), 0, s, a); break;
                case 3: hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 3>), g4, dim3(256), 0, s, a); break;
                case 7: hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 7>), g4, dim3(256), 0, s, a); break;
                case 11: hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 11>), g4, dim3(256), 0, s, a); break;
                case 15: hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 15>), g4, dim3(256), 0, s, a); break;
                case 19: hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 19>), g4, dim3(256), 0, s, a); break;
                case 23: hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 23>), g4, dim3(256), 0, s, a); break;
                case 32: hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 32>), g4, dim3(256), 0, s, a); break;
                case 64: hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 64>), g4, dim3(256), 0, s, a); break;
                case 96: hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 96>), g4, dim3(256), 0, s, a); break;
                case 128: hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 128>), g4, dim3(256), 0, s, a); break;     // de-paired MFMA order (bit-identical results; round 6: 1382 vs 958 us)
                default: return set_error("option attn_lab=%d is not instantiated", lab), -1;
            }
            prof_end(slot, s);
            return hipGetLastError() == hipSuccess ? 0 : (set_error("attention launch failed"), -1);
        }
#endif
        // attn_dit_form: 0 = per shape (below), 16 = the 16x16x32 tile whatever the shape says, 32 = the 32x32x16 tile (attn_dit32_kernel: measured SLOWER on MI355X,
        // profiles/r06_attn_tile_ab.md — kept selectable, parity-tested, not the default)
        const int form = (int)opt(OPT_ATTN_DIT_FORM);
#ifdef HVX_LAB
        if (form == 48 && a.chunk <= 0) {                               // (lab: the one-wave-per-SIMD, 512-register form with asm MFMAs)
            const dim3 g4q((a.n_rows + 511) / 512, a.heads, a.batch);
            if (a.q_log2) hipLaunchKernelGGL((attn_dit_a4_kernel<true>), g4q, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((attn_dit_a4_kernel<false>), g4q, dim3(256), 0, s, a);
            prof_end(slot, s);
            return hipGetLastError() == hipSuccess ? 0 : (set_error("attention launch failed"), -1);
        }
#endif
        // 0 / 17: the rotated pipeline (ROT 1) where it exists — pre-scaled queries, no chunk mask, >= 2048 rows: the DiT of the batched path —; 16: the in-tile
        // pipeline everywhere (the round-5 product tile)
        if ((form == 0 || form == 17) && a.chunk <= 0 && a.n_rows >= 2048 && a.q_log2) {
            hipLaunchKernelGGL((attn_dit_kernel<4, true, true, 0, 4, 1>), g4, dim3(256), 0, s, a);
        } else if (form == 32 && a.chunk <= 0) {
            const dim3 g1q((a.n_rows + 127) / 128, a.heads, a.batch);
            if (a.q_log2) hipLaunchKernelGGL((attn_dit32_kernel<1, true>), g1q, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((attn_dit32_kernel<1, false>), g1q, dim3(256), 0, s, a);
        } else if (a.n_rows >= 2048 && a.chunk <= 0) {
            if (a.q_log2) hipLaunchKernelGGL((attn_dit_kernel<4, true, true>), g4, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((attn_dit_kernel<4, true, false>), g4, dim3(256), 0, s, a);
        } else {
            if (a.q_log2) hipLaunchKernelGGL((attn_dit_kernel<2, true, true>), g2, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((attn_dit_kernel<2, true, false>), g2, dim3(256), 0, s, a);
        }
        prof_end(slot, s);
        return hipGetLastError() == hipSuccess ? 0 : (set_error("attention launch failed"), -1);
    }
    if (a.q_log2) a.scale = 0.6931471805599453f;                   // generic kernels: softmax(s_log2 * ln 2)
    const bool big = a.n_rows >= 256;
    const int rows_per_wave = big ? 32 : 16;
    const int n_qt = (a.n_rows + rows_per_wave - 1) / rows_per_wave;
    dim3 grid((n_qt * a.n_splits + 3) / 4, a.heads, a.batch);
    const bool merge = !big && a.sub_chunk > 0;
    if (merge) grid.x = a.n_splits;                                // one workgroup per split, 4 waves merge in LDS
    // QK^T + PV flops when the key length is known on the host (DiT); 0 for the device-length LLM calls
    const double flops = a.kv_len ? 0.0 : 4.0 * a.n_rows * (double)a.kv_len_const * 64.0 * a.heads * a.batch * (a.causal ? 0.5 : 1.0);
    const int slot = prof_begin(a.kv_len ? PK_ATTN_LLM : PK_ATTN, flops, s);
    if (a.kv_frag) {
        if (merge && a.n_rows > 16) hipLaunchKernelGGL((attn_fwd_kernel<T, 2, true, true>), grid, dim3(256), 0, s, a);      // decode, 17..32 rows (K = 3, 4)
        else if (merge && a.n_sub == 8) hipLaunchKernelGGL((attn_fwd_kernel<T, 1, true, true, 8>), grid, dim3(512), 0, s, a);
        else if (merge) hipLaunchKernelGGL((attn_fwd_kernel<T, 1, true, true>), grid, dim3(256), 0, s, a);
        else if (big) hipLaunchKernelGGL((attn_fwd_kernel<T, 2, false, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<T, 1, false, true>), grid, dim3(256), 0, s, a);
    } else if (merge && a.n_rows > 16) hipLaunchKernelGGL((attn_fwd_kernel<T, 2, true, false>), grid, dim3(256), 0, s, a);
    else if (merge) hipLaunchKernelGGL((attn_fwd_kernel<T, 1, true, false>), grid, dim3(256), 0, s, a);
    else if (big) hipLaunchKernelGGL((attn_fwd_kernel<T, 2, false, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<T, 1, false, false>), grid, dim3(256), 0, s, a);
    if (a.n_splits > 1 && !a.skip_combine) {
        dim3 g2((a.n_rows + 3) / 4, a.heads, a.batch);
        hipLaunchKernelGGL((attn_combine_kernel<T>), g2, dim3(256), 0, s, a);
    }
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("attention launch failed"), -1);
}

int launch_attention(const AttnArgs& a_in, hipStream_t s) {
    AttnArgs a = a_in;
    if (a.n_rows <= 0 || a.batch <= 0) return 0;
    if (a.n_splits < 1) a.n_splits = 1;

    if (a.kn < 1) a.kn = a.n_rows;
    if (a.n_sub != 8 || !a.kv_frag || a.n_rows > 16 || a.dtype != DT_BF16) a.n_sub = 4;       // (the 8-wave form exists for the one-tile bf16 decode over fragment-order caches)
    if (a.n_splits == 1 || a.n_rows > 32 || a.sub_chunk * a.n_sub != a.split_chunk || (a.sub_chunk & 31)) a.sub_chunk = 0;
    if (a.kv_frag && (a.chunk > 0 || !a.causal)) return set_error("launch_attention: fragment-order caches serve the causal LLM calls only"), -1;
    if (a.o_frag_kt > 0 && a.dtype == DT_F32) return set_error("launch_attention: fragment-order output is 16-bit only"), -1;
    if ((a.v_ld & 31) || (a.n_splits > 1 && ((a.split_chunk & 31) || !a.part_o || !a.part_ml || a.n_rows_pad < a.n_rows))) {
        set_error("launch_attention: bad geometry v_ld=%d split_chunk=%d", a.v_ld, a.split_chunk);
        return -1;
    }
    return a.dtype == DT_BF16 ? launch_t<bf16_t>(a, s) : launch_t<float>(a, s);
}

}  // namespace hvx
