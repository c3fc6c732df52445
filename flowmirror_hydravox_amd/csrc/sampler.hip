// sampler.hip — repetition-aware top-k/top-p sampler on gfx950 (see hvx_kernels.h: SampleArgs).
//
// Restates server/model_utils/cosyvoice/utils/common.py:138-166 (ras_sampling / nucleus_sampling /
// random_sampling) and the EOS-rejection loop + shared history snapshot of
// cosyvoice/llm/llm_multi_head_v3.py:151-166, 890-900.  One workgroup per sequence; the K heads are
// sampled one after the other because each consumes a data-dependent amount of the sequence's
// Exp(1) noise stream (torch CPU generator, produced by the host: `multinomial(1)` == argmax(p / q)).
//   softmax      : fp32, block reduction (max, then sum in fp64), p kept in LDS
//   top-k / top-p: tournament selection — every thread keeps its best (value desc, index asc) candidate,
//                  a wave-shuffle + LDS reduction picks the winner, only the winner's owner rescans;
//                  stops exactly like the reference loop `cum < top_p and n < top_k` (cum in fp32, sorted order)
//   draw         : lane-parallel exponential race over the <= 64 candidates (first max)
//   repetition   : count of the drawn id in the last win_size tokens; >= rep_thresh -> full-vocabulary race
#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

struct Best {
    float v;
    int i;
};
__device__ __forceinline__ bool better(float v, int i, float v2, int i2) { return (v > v2) || (v == v2 && i < i2); }

// Wave-wide best on the DPP network instead of 12 ds_bpermute round trips: (value, index) becomes one 64-bit key whose unsigned order is
// the `better` order (values here are -1 or >= 0, so the float bits + 1 are monotonic; the inverted index breaks ties towards the lower
// index), the maximum is folded to lane 63 with row shifts and row broadcasts (max is idempotent: overlapping contributions are
// harmless) and read back through a scalar register.
__device__ __forceinline__ unsigned long long best_key(Best b) {
    const unsigned hi = b.v < 0.0f ? 0u : __float_as_uint(b.v) + 1u;
    return ((unsigned long long)hi << 32) | (unsigned)(~b.i);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_step(unsigned long long k) {
    const unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);
    const unsigned lo2 = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi2 = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROW_MASK, 0xf, false);
    const unsigned long long k2 = ((unsigned long long)hi2 << 32) | lo2;
    return k2 > k ? k2 : k;
}
__device__ __forceinline__ Best wave_best(Best b) {
    unsigned long long k = best_key(b);
    k = dpp_max_step<0x111, 0xf>(k);          // row_shr:1
    k = dpp_max_step<0x112, 0xf>(k);          // row_shr:2
    k = dpp_max_step<0x114, 0xf>(k);          // row_shr:4
    k = dpp_max_step<0x118, 0xf>(k);          // row_shr:8   -> lane 15 of every row holds its row's maximum
    k = dpp_max_step<0x142, 0xa>(k);          // row_bcast:15 into rows 1 and 3
    k = dpp_max_step<0x143, 0xc>(k);          // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), 63);
    Best r;
    r.v = hi == 0u ? -1.0f : __uint_as_float(hi - 1u);
    r.i = (int)~lo;
    return r;
}

// block-wide best; every thread returns the same winner.  red_v/red_i: LDS [4]
__device__ __forceinline__ Best block_best(Best b, float* red_v, int* red_i) {
    b = wave_best(b);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red_v[wave] = b.v;
        red_i[wave] = b.i;
    }
    __syncthreads();
    Best r = {red_v[0], red_i[0]};
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (better(red_v[w], red_i[w], r.v, r.i)) {
            r.v = red_v[w];
            r.i = red_i[w];
        }
    return r;
}

constexpr int MAX_CAND = 64;
constexpr int BIG_IDX = 0x7fffffff;

constexpr int VPT = 32;                    // vocabulary entries per thread: V <= 256 * 32 = 8192

__global__ __launch_bounds__(256) void ras_sample_kernel(SampleArgs a) {
    // Each thread keeps its strided slice of the distribution (index i = u * 256 + tid) in REGISTERS for the whole head:
    // softmax, the tournament rescans and the full-vocabulary race never go back to memory.
    const int V = a.V;
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    __shared__ double red_d[4];
    __shared__ float cand_v[MAX_CAND];
    __shared__ int cand_i[MAX_CAND];
    __shared__ int sh_pick;

    const int s = blockIdx.x, tid = threadIdx.x;
    if (a.active && !a.active[s]) return;
    const long long cursor0 = a.cursor[s];
    long long cursor = cursor0;
    const float* noise = a.noise + (long long)s * a.noise_ss;
    // noise is a ring of noise_len values indexed by the ABSOLUTE stream position (position % noise_len); positions below `limit` are
    // valid.  Without a limit array the buffer is a plain window [0, noise_len).
    const long long cap = a.noise_len;
    const long long limit = a.noise_limit ? a.noise_limit[s] : cap;
    const int hist_len = a.hist_len[s];
    const int* hist = a.hist + (long long)s * a.hist_ss;
    const int min_len = a.min_len[s];
    const int win = (a.win_size == 0) ? hist_len : min(a.win_size, hist_len);
    bool overflow = false;
    // Everything the trial loops will touch is requested up front, beside the first head's logits: the repetition window (shared by
    // the K heads, llm_multi_head_v3.py:891-900) and the next NW noise values (a step without EOS retries consumes <= K * top_k).
    constexpr int NW = 128, HW = 256;
    __shared__ float s_noise[NW];
    __shared__ int s_hist[HW];
    if (tid < NW) s_noise[tid] = (cursor0 + tid < limit) ? noise[(cursor0 + tid) % cap] : 1.0f;
    if (tid < HW && tid < win) s_hist[tid] = hist[hist_len - win + tid];
    auto noise_at = [&](long long i) -> float { return (i - cursor0) < NW ? s_noise[i - cursor0] : noise[i % cap]; };

    for (int j = 0; j < a.head_k && !overflow; ++j) {
        const float* lp = a.logp + (long long)s * a.logp_ss + (long long)j * a.logp_hs;
        // ---- softmax(logp) (common.py:149 / :165): 32 independent loads per thread, then registers only ------------
        float pv[VPT];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int i = u * 256 + tid;
            pv[u] = i < V ? lp[i] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < VPT; ++u) mx = fmaxf(mx, pv[u]);
        {
            Best b = block_best(Best{mx, tid}, red_v, red_i);
            mx = b.v;
        }
        double sum = 0.0;
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const float e = (u * 256 + tid) < V ? expf(pv[u] - mx) : 0.0f;
            pv[u] = e;
            sum += (double)e;
        }
        sum = wave_sum_d(sum);
        __syncthreads();
        if ((tid & 63) == 0) red_d[tid >> 6] = sum;
        __syncthreads();
        const float denom = (float)(red_d[0] + red_d[1] + red_d[2] + red_d[3]);
#pragma unroll
        for (int u = 0; u < VPT; ++u) pv[u] = (u * 256 + tid) < V ? pv[u] / denom : -1.0f;       // padding can never win

        // ---- nucleus candidates: stable descending order, `cum < top_p and n < top_k` (common.py:146-157)
        unsigned taken = 0;                                      // bit u: entry u of this thread already selected
        auto local_best = [&]() {
            Best m = {-1.0f, BIG_IDX};
#pragma unroll
            for (int u = 0; u < VPT; ++u) {
                const float v = ((taken >> u) & 1u) ? -1.0f : pv[u];
                if (v > m.v) m = Best{v, u * 256 + tid};         // ascending u == ascending index: ties keep the lower index
            }
            return m;
        };
        Best mine = local_best();
        int n = 0;
        float cum = 0.0f;
        const int kmax = min(min(a.top_k, MAX_CAND), V);
        while (cum < a.top_p && n < kmax) {
            const Best wv = block_best(mine, red_v, red_i);
            if (tid == 0) {
                cand_v[n] = wv.v;
                cand_i[n] = wv.i;
            }
            cum = cum + wv.v;                                   // fp32, in sorted order, like the reference's 0-dim tensor
            ++n;
            if ((wv.i & 255) == tid) {                          // owner: retire the winner and rescan its registers
                taken |= 1u << (wv.i >> 8);
                mine = local_best();
            }
        }
        __syncthreads();

        const bool ignore_eos = (hist_len + j) < min_len;
        int result = -1;
        for (int trial = 0;; ++trial) {
            // ---- multinomial(1) over the candidates == first argmax of p / Exp(1) ---------------------
            if (cursor + n > limit) {
                overflow = true;
                break;
            }
            if (tid < 64) {
                Best b = {-1.0f, BIG_IDX};
                if (tid < n) b = Best{cand_v[tid] / noise_at(cursor + tid), tid};
                b = wave_best(b);
                if (tid == 0) sh_pick = cand_i[b.i];
            }
            __syncthreads();
            int c = sh_pick;
            cursor += n;
            // ---- repetition check over the shared history snapshot (common.py:140-142) ----------------
            int rep = 0;
            for (int i = tid; i < win; i += 256) rep += ((i < HW ? s_hist[i] : hist[hist_len - win + i]) == c) ? 1 : 0;
            rep = (int)wave_sum((float)rep);
            __syncthreads();
            if ((tid & 63) == 0) red_i[tid >> 6] = rep;
            __syncthreads();
            rep = red_i[0] + red_i[1] + red_i[2] + red_i[3];
            if (rep >= a.rep_thresh) {
                // random_sampling: race over the full vocabulary (common.py:164-166)
                if (cursor + V > limit) {
                    overflow = true;
                    break;
                }
                float q[VPT];
#pragma unroll
                for (int u = 0; u < VPT; ++u) {
                    const int i = u * 256 + tid;
                    q[u] = i < V ? noise[(cursor + i) % cap] : 1.0f;
                }
                Best b = {-1.0f, BIG_IDX};
#pragma unroll
                for (int u = 0; u < VPT; ++u) {
                    const int i = u * 256 + tid;
                    const float r = pv[u] / q[u];
                    if (i < V && better(r, i, b.v, b.i)) b = Best{r, i};
                }
                b = block_best(b, red_v, red_i);
                c = b.i;
                cursor += V;
            }
            __syncthreads();
            if (!ignore_eos || c < a.Vs) {
                result = c;
                break;
            }
            if (trial + 1 > a.max_trials) {                     // num_trials > max_trials -> RuntimeError in the reference
                result = -1;
                break;
            }
        }
        if (overflow) break;
        if (tid == 0) a.out_ids[(long long)s * a.head_k + j] = result;
        __syncthreads();
    }
    if (overflow) {
        if (tid < a.head_k) a.out_ids[(long long)s * a.head_k + tid] = -2;
    } else if (tid == 0) {
        a.cursor[s] = cursor;
    }
}

int launch_ras_sample(const SampleArgs& a, hipStream_t s) {
    if (a.n_seq <= 0) return 0;
    if (a.top_k < 1 || a.top_k > MAX_CAND) {
        set_error("ras_sample: top_k=%d outside [1,%d]", a.top_k, MAX_CAND);
        return -1;
    }
    if (a.head_k < 1 || a.head_k > 256) {
        set_error("ras_sample: head_k=%d", a.head_k);
        return -1;
    }
    if (a.V > 256 * VPT) {
        set_error("ras_sample: vocab %d exceeds %d", a.V, 256 * VPT);
        return -1;
    }
    const size_t lds = 0;
    const int slot = prof_begin(PK_SAMPLER, (double)a.n_seq * a.head_k * a.V * 4.0, s);
    hipLaunchKernelGGL(ras_sample_kernel, dim3(a.n_seq), dim3(256), lds, s, a);
    prof_end(slot, s);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("ras_sample launch failed"), -1);
}

// ---------------------------------------------------------------------------------------------------------------------
// decode_advance: the reference's host logic between two steps, one thread per sequence (K <= 8 tokens each).
//   for t in ids: t == -1 -> RuntimeError (err = 1); t >= speech_token_size -> stop; else append; len(out) >= max_len -> stop
//   an empty accepted group stops the sequence; the accepted group is what the next step feeds (llm_multi_head_v3.py:898-905)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void decode_advance_kernel(AdvanceArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_seq) return;
    int* st = a.seq_state + i * 8;
    const int K = a.head_k;
    int pos = st[0], out_len = st[1], done = st[2];
    const int min_len = st[3], max_len = st[4];
    int n_next = 0;
    if (!done && a.ids[i * K] == -2) {
        // the sampler ran out of pre-generated noise for this sequence (all K ids are -2, its cursor is untouched): the step is void.
        // Nothing advances — the same rows are fed again (rewriting the same KV entries) until the host has refilled the window.
        st[6] = 2;
        return;
    }
    if (!done) {
        st[6] = 0;                                           // (a stall flag from an earlier void step is history once a step counts)
        pos += a.ctrl[2 * a.n_seq + i];                      // rows fed by the step that just ran
        for (int j = 0; j < K; ++j) {
            const int t = a.ids[i * K + j];
            if (t < 0) {                                     // -1: max_trials exhausted (the reference raises RuntimeError)
                st[6] = 1;
                done = 1;
                break;
            }
            if (t >= a.speech_tokens) {                      // any stop id
                done = 1;
                break;
            }
            if (out_len < a.max_out) a.out_tokens[(long long)i * a.max_out + out_len] = t;
            a.hist[(long long)i * a.win_cap + (out_len % a.win_cap)] = t;
            a.tok[i * K + n_next] = t;
            ++out_len;
            ++n_next;
            if (out_len >= max_len) {
                done = 1;
                break;
            }
        }
        if (n_next == 0) done = 1;
        st[5] += 1;
    }
    const int nn = done ? 0 : n_next;
    for (int j = nn; j < K; ++j) a.tok[i * K + j] = -1;
    a.ctrl[0 * a.n_seq + i] = i;
    a.ctrl[1 * a.n_seq + i] = pos;
    a.ctrl[2 * a.n_seq + i] = nn;
    a.ctrl[3 * a.n_seq + i] = pos + nn;
    a.ctrl[4 * a.n_seq + i] = nn ? (i * K + nn - 1) : -1;
    const int hl = out_len < a.win_cap ? out_len : a.win_cap;
    a.hist_len[i] = hl;
    a.min_adj[i] = min_len - (out_len - hl);
    a.active[i] = done ? 0 : 1;
    st[0] = pos;
    st[1] = out_len;
    st[2] = done;
}

// A new sequence takes over slot i of a running decode grid (continuous batching): what the host-side set-up of a batch writes for
// every sequence, for one slot, ordered on the stream between two steps.  Its prefix but the last row is already in the slot's KV cache.
__global__ void decode_join_kernel(AdvanceArgs a, long long* cursor, int i, int first_tok, int pos, int min_len, int max_len) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int K = a.head_k;
    a.tok[i * K] = first_tok;
    for (int j = 1; j < K; ++j) a.tok[i * K + j] = -1;
    a.ctrl[0 * a.n_seq + i] = i;
    a.ctrl[1 * a.n_seq + i] = pos;
    a.ctrl[2 * a.n_seq + i] = 1;
    a.ctrl[3 * a.n_seq + i] = pos + 1;
    a.ctrl[4 * a.n_seq + i] = i * K;
    a.hist_len[i] = 0;
    a.min_adj[i] = min_len;
    a.active[i] = 1;
    int* st = a.seq_state + i * 8;
    st[0] = pos; st[1] = 0; st[2] = 0; st[3] = min_len; st[4] = max_len; st[5] = 0; st[6] = 0; st[7] = 0;
    cursor[i] = 0;
}

int launch_decode_join(const AdvanceArgs& a, long long* cursor, int slot, int first_tok, int pos, int min_len, int max_len, hipStream_t s) {
    hipLaunchKernelGGL(decode_join_kernel, dim3(1), dim3(64), 0, s, a, cursor, slot, first_tok, pos, min_len, max_len);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("decode_join launch failed"), -1);
}

int launch_decode_advance(const AdvanceArgs& a, hipStream_t s) {
    if (a.n_seq <= 0) return 0;
    hipLaunchKernelGGL(decode_advance_kernel, dim3((a.n_seq + 63) / 64), dim3(64), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : (set_error("decode_advance launch failed"), -1);
}

}  // namespace hvx
