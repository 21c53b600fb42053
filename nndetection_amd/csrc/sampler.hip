// Batch-level hard-negative sampling on the device for gfx950 -- replaces DetectionHeadHNM.select_indices +
// HardNegativeSamplerBatched.__call__ / select_positives / select_negatives (nndet/arch/heads/comb.py:247-276,
// nndet/core/boxes/sampler.py:57-98,154-185,237-270). The reference does torch.where x3 (one host sync each), a topk over
// all negatives (4.7 M anchors at batch 4) and two randperm calls whose sizes come back from the device. Here every count
// stays on the device and the outputs have fixed capacities:
//   positives : the num_pos anchors with label >= 1 that have the smallest selection keys   (== positive[randperm(n)[:num_pos]])
//   pool      : the `pool` anchors with label == 0 and the highest foreground probability (ties: lowest index), sorted
//   negatives : the num_neg pool POSITIONS with the smallest selection keys                 (== pool[randperm(pool)[:num_neg]])
// Selection key = hash(seed, index) -- a uniformly random subset in random order, the distribution of randperm(n)[:k] -- or,
// in deterministic mode (parity tests), the reversed index: exactly what randperm := arange(n-1, -1, -1) selects.
// Both selections are exact radix selects (radix_select.h) over the label / logit arrays; nothing of size N is written.
#include "common.h"
#include "radix_select.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

struct SpArgs {
    const float* labels;   // [N]
    const float* scores;   // [N, C] logits (is_prob == 0) or [N] probabilities (is_prob == 1, C == 1)
    int64_t N;
    int32_t C, is_prob;
    int32_t P_cap, NEG_cap, POOL_cap, min_neg;
    double ratio, pool_size;
    u64 seed;
    int32_t det;
};
// params (device int32[8]): 0 n_pos, 1 n_neg, 2 num_pos, 3 num_neg, 4 pool, 5 cnt_pos, 6 cnt_pool

#define SP_ITEMS 8

__device__ __forceinline__ uint32_t sp_hash(u64 seed, uint32_t i) {
    u64 x = ((u64)i + seed) * 0x9E3779B97F4A7C15ULL;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    return (uint32_t)(x >> 32) ^ (uint32_t)x;
}
__device__ __forceinline__ u64 sp_key_pos(const SpArgs& A, uint32_t i) {
    return ((u64)(A.det ? ~i : sp_hash(A.seed, i)) << 32) | (u64)i;
}
__device__ __forceinline__ float sp_fgprob(const SpArgs& A, int64_t i) {
    if (A.is_prob) return A.scores[i];
    float m = -INFINITY;
    for (int c = 0; c < A.C; ++c) m = fmaxf(m, A.scores[i * A.C + c]);
    return 1.f / (1.f + expf(-m));                       // max(sigmoid(x)) == sigmoid(max(x)) (comb.py:262-266)
}
__device__ __forceinline__ u64 sp_key_neg(const SpArgs& A, int64_t i) {
    return ((u64)(~f32_sortable(sp_fgprob(A, i))) << 32) | (u64)(uint32_t)i;
}
__device__ __forceinline__ u64 sp_key_neg_p(float prob, int64_t i) { return ((u64)(~f32_sortable(prob)) << 32) | (u64)(uint32_t)i; }
// The SP_ITEMS labels and (for probabilities: C == 1) scores of a thread, requested TOGETHER: the passes below used to load a label, branch
// on it, and only then load the score -- 16 dependent round trips per thread, 26-47 us per pass over 38 MB (round 6). Rows beyond N read
// row 0 and get the label -1 (neither positive nor negative).
__device__ __forceinline__ void sp_load_items(const SpArgs& A, int64_t i0, float* lab, float* prob) {
#pragma unroll
    for (int t = 0; t < SP_ITEMS; ++t) {
        const int64_t i = i0 + (int64_t)t * 256;
        lab[t] = A.labels[i < A.N ? i : 0];
    }
    if (A.is_prob) {
#pragma unroll
        for (int t = 0; t < SP_ITEMS; ++t) {
            const int64_t i = i0 + (int64_t)t * 256;
            prob[t] = A.scores[i < A.N ? i : 0];
        }
    }
#pragma unroll
    for (int t = 0; t < SP_ITEMS; ++t) if (i0 + (int64_t)t * 256 >= A.N) lab[t] = -1.f;
}

__global__ void k_sp_init(int32_t* params, u64* prefix, unsigned* hist) {
    const int t = threadIdx.x;
    if (t < 8) params[t] = 0;
    if (t < 2) prefix[t] = 0;
    for (int i = t; i < 512; i += blockDim.x) hist[i] = 0;
}

// grid-stride over the labels, ONE pair of atomics per workgroup: with one pair per WAVE of a 2 318-workgroup grid the ~9 000 adds
// to params[1] (nearly every anchor is background) serialised in L2 -- 112 us for a 19 MB read
__global__ __launch_bounds__(256) void k_sp_count(SpArgs A, int32_t* params) {
    __shared__ int red[2][4];
    int np = 0, nn = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < A.N; i += (int64_t)gridDim.x * 256) {
        const float l = A.labels[i];
        np += (l >= 1.f); nn += (l == 0.f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { np += __shfl_xor(np, o, 64); nn += __shfl_xor(nn, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = np; red[1][threadIdx.x >> 6] = nn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        np = red[0][0] + red[0][1] + red[0][2] + red[0][3]; nn = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        if (np) atomicAdd(&params[0], np);
        if (nn) atomicAdd(&params[1], nn);
    }
}

// sampler.py:154-185,237-262 -- the same integer / double arithmetic as the Python code
__global__ void k_sp_derive(SpArgs A, int32_t* params, int* krem) {
    const int n_pos = params[0], n_neg = params[1];
    const int num_pos = n_pos < A.P_cap ? n_pos : A.P_cap;
    int num_neg = (int)((double)(num_pos > 1 ? num_pos : 1) * A.ratio);
    if (num_neg < A.min_neg) num_neg = A.min_neg;
    if (num_neg > n_neg) num_neg = n_neg;
    int pool = (int)((double)num_neg * A.pool_size);
    if (pool > n_neg) pool = n_neg;
    if (pool > A.POOL_cap) pool = A.POOL_cap;
    if (num_neg > pool) num_neg = pool;
    params[2] = num_pos; params[3] = num_neg; params[4] = pool;
    krem[0] = num_pos; krem[1] = pool;
    krem[2] = 0; krem[3] = 0;                 // "selection decided" flags of the two problems (radix_pick_wave)
}

// two problems in one pass over the anchors: 0 = positives by selection key, 1 = negatives by (probability desc, index)
__global__ __launch_bounds__(256) void k_sp_hist(SpArgs A, const u64* __restrict__ prefix, unsigned* __restrict__ hist, int shift,
                                                 const int* __restrict__ done) {
    const bool d0 = done[0] != 0, d1 = done[1] != 0;     // a problem whose selection is decided takes no part in the remaining passes
    if (d0 && d1) return;                                 // (scores / hashes are nearly unique: usually after the 4 passes over them)
    __shared__ unsigned h[2][256];
    h[0][threadIdx.x] = 0; h[1][threadIdx.x] = 0;
    __syncthreads();
    const u64 p0 = prefix[0], p1 = prefix[1];
    // (a workgroup can walk several blocks of 256 * SP_ITEMS anchors -- NNDET_SP_HIST_WGS caps the grid -- but one block per workgroup is the
    // fastest: 2 318 workgroups 0.41 ms for the detection-loss phase, 1 024: 0.41, 512: 0.43, 256: 0.53. The passes share the chip with the
    // segmentation branch's forward kernel on its side stream; the same-address atomics at their end are not what they wait for)
    for (int64_t blk = blockIdx.x; blk * (256 * SP_ITEMS) < A.N; blk += gridDim.x) {
        const int64_t i0 = (blk * 256) * SP_ITEMS + threadIdx.x;
        float lab[SP_ITEMS], prob[SP_ITEMS];
        sp_load_items(A, i0, lab, prob);
#pragma unroll
        for (int t = 0; t < SP_ITEMS; ++t) {
            const int64_t i = i0 + (int64_t)t * 256;
            const float l = lab[t];
            if (l >= 1.f) {
                if (!d0) {
                    const u64 key = sp_key_pos(A, (uint32_t)i);
                    if ((shift >= 56) || ((key >> (shift + 8)) == (p0 >> (shift + 8)))) atomicAdd(&h[0][(unsigned)(key >> shift) & 255u], 1u);
                }
            } else if (l == 0.f && !d1) {
                const u64 key = A.is_prob ? sp_key_neg_p(prob[t], i) : sp_key_neg(A, i);
                if ((shift >= 56) || ((key >> (shift + 8)) == (p1 >> (shift + 8)))) atomicAdd(&h[1][(unsigned)(key >> shift) & 255u], 1u);
            }
        }
    }
    __syncthreads();
    const unsigned v0 = h[0][threadIdx.x], v1 = h[1][threadIdx.x];
    if (v0) atomicAdd(&hist[threadIdx.x], v0);
    if (v1) atomicAdd(&hist[256 + threadIdx.x], v1);
}

__global__ __launch_bounds__(128) void k_sp_pick(u64* __restrict__ prefix, int* __restrict__ krem, unsigned* __restrict__ hist, int shift) {
    const int i = threadIdx.x >> 6;
    if (krem[2 + i]) return;                              // decided in an earlier pass (its histogram stayed empty)
    radix_pick_wave(prefix + i, krem + i, hist + i * 256, shift, threadIdx.x & 63, krem + 2 + i);
}

__global__ __launch_bounds__(256) void k_sp_collect(SpArgs A, const u64* __restrict__ kth, int32_t* __restrict__ params,
                                                    u64* __restrict__ list_pos, u64* __restrict__ list_pool) {
    const u64 k0 = kth[0], k1 = kth[1];
    const int num_pos = params[2], pool = params[4];
    const int64_t i0 = ((int64_t)blockIdx.x * 256) * SP_ITEMS + threadIdx.x;
    float lab[SP_ITEMS], prob[SP_ITEMS];
    sp_load_items(A, i0, lab, prob);
#pragma unroll
    for (int t = 0; t < SP_ITEMS; ++t) {
        const int64_t i = i0 + (int64_t)t * 256;
        const float l = lab[t];
        if (l >= 1.f && num_pos > 0) {
            const u64 key = sp_key_pos(A, (uint32_t)i);
            if (key <= k0) { const int p = atomicAdd(&params[5], 1); if (p < A.P_cap) list_pos[p] = key; }
        } else if (l == 0.f && pool > 0) {
            const u64 key = A.is_prob ? sp_key_neg_p(prob[t], i) : sp_key_neg(A, i);
            if (key <= k1) { const int p = atomicAdd(&params[6], 1); if (p < A.POOL_cap) list_pool[p] = key; }
        }
    }
}

__global__ void k_sp_fill(u64* a, int na, u64* b, int nb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) a[i] = ~0ULL;
    if (i < nb) b[i] = ~0ULL;
}

// selection keys over the POSITIONS of the sorted pool (randperm(pool)[:num_neg] of sampler.py:205-207)
__global__ void k_sp_poolkeys(SpArgs A, const int32_t* __restrict__ params, u64* __restrict__ sel) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= A.POOL_cap) return;
    const int pool = params[4];
    sel[j] = j < pool ? (((u64)(A.det ? (uint32_t)(pool - 1 - j) : sp_hash(A.seed ^ 0xA5A5A5A55A5A5A5AULL, (uint32_t)j)) << 32) | (u64)(uint32_t)j) : ~0ULL;
}

// The reference turns the selections into masks and reads them back with torch.where (comb.py:268-276): the index lists it
// trains on are in ASCENDING anchor order. k_sp_negidx resolves the chosen pool positions to anchor indices; both lists are
// then sorted by index (the low 32 bits) and emitted.
__global__ void k_sp_negidx(SpArgs A, const int32_t* __restrict__ params, const u64* __restrict__ pool_sorted,
                            const u64* __restrict__ sel_sorted, u64* __restrict__ neg_tmp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.NEG_cap) return;
    neg_tmp[i] = i < params[3] ? (u64)(uint32_t)pool_sorted[(uint32_t)sel_sorted[i]] : ~0ULL;
}

__global__ void k_sp_emit(SpArgs A, const int32_t* __restrict__ params, const u64* __restrict__ pos_sorted,
                          const u64* __restrict__ neg_sorted, int64_t* __restrict__ pos_idx, int64_t* __restrict__ neg_idx,
                          int64_t* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int num_pos = params[2], num_neg = params[3];
    if (i < A.P_cap) pos_idx[i] = i < num_pos ? (int64_t)(uint32_t)pos_sorted[i] : -1;
    if (i < A.NEG_cap) neg_idx[i] = i < num_neg ? (int64_t)(uint32_t)neg_sorted[i] : -1;
    if (i == 0) { counts[0] = num_pos; counts[1] = num_neg; counts[2] = params[0]; counts[3] = params[1]; }
}

// ---- the whole tail in ONE workgroup (round 5): the three <= POOL_cap-key sorts, the pool-position keys, the index resolution and
// the emit were 4 rocPRIM radix sorts (3-4 launches each) + 4 small kernels = ~16 launches of 5-14 us in a row on the critical chain
// between the forward and the backward pass (~0.12 ms). LDS bitonic network (the post-processing's, csrc/postproc.hip: k_pp_bitonic) on
// keys masked like the radix sorts' bit ranges; every sorted key set is duplicate-free apart from its ~0 padding, so the result is the
// radix sorts' bit for bit. NNDET_SP_TAIL_FUSED=0 or capacities above 4096: the rocPRIM path.
__device__ __forceinline__ void sp_bitonic(u64* s, int N, u64 mask, int tid) {
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (N >> 1); t += 1024) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // pair index t -> element with bit j clear
                const int l = i | j;
                const u64 a = s[i], b = s[l];
                const bool up = (i & k) == 0;
                if (((a & mask) > (b & mask)) == up) { s[i] = b; s[l] = a; }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(1024) void k_sp_tail(SpArgs A, const int32_t* __restrict__ params, const u64* __restrict__ list_pos,
                                                  const u64* __restrict__ list_pool, int NP2, int NQ2, int NN2,
                                                  int64_t* __restrict__ pos_idx, int64_t* __restrict__ neg_idx, int64_t* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) char sp_smem[];
    u64* a = reinterpret_cast<u64*>(sp_smem);            // [max(NP2, NQ2)]: positives, then the pool
    u64* b = a + (NP2 > NQ2 ? NP2 : NQ2);                // [NQ2]: selection keys over the pool positions
    u64* c = b + NQ2;                                    // [NN2]: the chosen negatives
    const int tid = threadIdx.x;
    const int num_pos = params[2], num_neg = params[3], pool = params[4];
    // positives: everything collected IS the selection -> ascending anchor index (low 32 bits; padding sorts last)
    for (int i = tid; i < NP2; i += 1024) a[i] = i < A.P_cap ? list_pos[i] : ~0ULL;
    __syncthreads();
    sp_bitonic(a, NP2, 0xffffffffULL, tid);
    for (int i = tid; i < A.P_cap; i += 1024) pos_idx[i] = i < num_pos ? (int64_t)(uint32_t)a[i] : -1;
    __syncthreads();
    // pool: ascending key = descending foreground probability, ties by index
    for (int i = tid; i < NQ2; i += 1024) a[i] = i < A.POOL_cap ? list_pool[i] : ~0ULL;
    // selection keys over the POSITIONS of the sorted pool (randperm(pool)[:num_neg], sampler.py:205-207)
    for (int j = tid; j < NQ2; j += 1024)
        b[j] = (j < A.POOL_cap && j < pool)
                   ? (((u64)(A.det ? (uint32_t)(pool - 1 - j) : sp_hash(A.seed ^ 0xA5A5A5A55A5A5A5AULL, (uint32_t)j)) << 32) | (u64)(uint32_t)j) : ~0ULL;
    __syncthreads();
    sp_bitonic(a, NQ2, ~0ULL, tid);
    sp_bitonic(b, NQ2, ~0ULL, tid);
    // chosen pool positions -> anchor indices, ascending (comb.py:268-276: torch.where on the masks)
    for (int i = tid; i < NN2; i += 1024) c[i] = (i < A.NEG_cap && i < num_neg) ? (u64)(uint32_t)a[(uint32_t)b[i]] : ~0ULL;
    __syncthreads();
    sp_bitonic(c, NN2, 0x1ffffffffULL, tid);
    for (int i = tid; i < A.NEG_cap; i += 1024) neg_idx[i] = i < num_neg ? (int64_t)(uint32_t)c[i] : -1;
    if (tid == 0) { counts[0] = num_pos; counts[1] = num_neg; counts[2] = params[0]; counts[3] = params[1]; }
}

struct SpWs {
    int32_t* params; u64* prefix; int* krem; unsigned* hist;
    u64 *list_pos, *pos_sorted, *list_pool, *pool_sorted, *sel, *sel_sorted, *neg_tmp, *neg_sorted;
    void* sort_tmp; size_t sort_tmp_bytes; size_t total;
};

static int sp_layout(int P_cap, int NEG_cap, int POOL_cap, char* base, SpWs* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    size_t o_pa = take(32), o_pr = take(16), o_kr = take(16), o_h = take(512 * 4);   // krem: 2 remaining ranks + 2 'decided' flags (k_sp_derive / k_sp_pick / k_sp_hist)
    size_t o_lp = take((size_t)P_cap * 8), o_ps = take((size_t)P_cap * 8);
    size_t o_lq = take((size_t)POOL_cap * 8), o_qs = take((size_t)POOL_cap * 8), o_se = take((size_t)POOL_cap * 8), o_ss = take((size_t)POOL_cap * 8);
    size_t o_nt = take((size_t)NEG_cap * 8), o_ns = take((size_t)NEG_cap * 8);
    size_t tmp = 0;
    const size_t nmax = (size_t)(P_cap > POOL_cap ? P_cap : POOL_cap);
    hipError_t e = rocprim::radix_sort_keys<rocprim::default_config, const u64*, u64*>(nullptr, tmp, nullptr, nullptr, nmax, 0, 64, (hipStream_t)0, false);
    if (e != hipSuccess) return (int)e;
    size_t o_tmp = take(tmp > 0 ? tmp : 256);
    w->sort_tmp_bytes = tmp; w->total = off;
    if (!base) return 0;                       // size query (nndet_hnm_sample_workspace_bytes): no pointer arithmetic on a null base (UBSan, round 6)
    w->params = (int32_t*)(base + o_pa); w->prefix = (u64*)(base + o_pr); w->krem = (int*)(base + o_kr); w->hist = (unsigned*)(base + o_h);
    w->list_pos = (u64*)(base + o_lp); w->pos_sorted = (u64*)(base + o_ps); w->list_pool = (u64*)(base + o_lq);
    w->pool_sorted = (u64*)(base + o_qs); w->sel = (u64*)(base + o_se); w->sel_sorted = (u64*)(base + o_ss);
    w->neg_tmp = (u64*)(base + o_nt); w->neg_sorted = (u64*)(base + o_ns);
    w->sort_tmp = base + o_tmp; w->sort_tmp_bytes = tmp; w->total = off;
    return 0;
}

static int sp_caps(int32_t pos_cap, double ratio, int32_t min_neg, double pool_size, int* NEG_cap, int* POOL_cap) {
    if (pos_cap < 0 || ratio < 0 || min_neg < 0 || pool_size < 1.0) return NNDET_EINVAL;
    int neg = (int)((double)(pos_cap > 1 ? pos_cap : 1) * ratio);
    if (neg < min_neg) neg = min_neg;
    if (neg < 1) neg = 1;
    const double pool = (double)neg * pool_size;
    if (pool > (double)(1 << 24)) return NNDET_EINVAL;
    *NEG_cap = neg; *POOL_cap = (int)pool > 0 ? (int)pool : 1;
    return 0;
}

extern "C" size_t nndet_hnm_sample_workspace_bytes(int32_t pos_cap, double neg_pos_ratio, int32_t min_neg, double pool_size) {
    int NEG_cap, POOL_cap;
    if (sp_caps(pos_cap, neg_pos_ratio, min_neg, pool_size, &NEG_cap, &POOL_cap)) return 0;
    SpWs w;
    if (sp_layout(pos_cap > 0 ? pos_cap : 1, NEG_cap, POOL_cap, nullptr, &w)) return 0;
    return w.total;
}

extern "C" int32_t nndet_hnm_neg_capacity(int32_t pos_cap, double neg_pos_ratio, int32_t min_neg) {
    int NEG_cap, POOL_cap;
    if (sp_caps(pos_cap, neg_pos_ratio, min_neg, 1.0, &NEG_cap, &POOL_cap)) return -1;
    return NEG_cap;
}

extern "C" int nndet_hnm_sample_f32(const float* labels, const float* scores, int32_t scores_are_probs, int64_t N, int32_t C,
                                    int32_t pos_cap, double neg_pos_ratio, int32_t min_neg, double pool_size, uint64_t seed,
                                    int32_t deterministic, int64_t* pos_idx, int64_t* neg_idx, int64_t* counts,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = as_stream(stream);
    int NEG_cap, POOL_cap;
    int rc = sp_caps(pos_cap, neg_pos_ratio, min_neg, pool_size, &NEG_cap, &POOL_cap);
    if (rc) return rc;
    if (N <= 0 || N >= (1LL << 32) || C <= 0 || (scores_are_probs && C != 1)) return NNDET_EINVAL;
    if (!labels || !scores || !pos_idx || !neg_idx || !counts || !workspace) return NNDET_EINVAL;
    const int P_cap = pos_cap > 0 ? pos_cap : 1;
    SpWs w;
    rc = sp_layout(P_cap, NEG_cap, POOL_cap, (char*)workspace, &w);
    if (rc) return rc;
    if (w.total > workspace_bytes) return NNDET_EWORKSPACE;
    SpArgs A;
    A.labels = labels; A.scores = scores; A.N = N; A.C = C; A.is_prob = scores_are_probs;
    A.P_cap = pos_cap; A.NEG_cap = NEG_cap; A.POOL_cap = POOL_cap; A.min_neg = min_neg;
    A.ratio = neg_pos_ratio; A.pool_size = pool_size; A.seed = seed; A.det = deterministic;
    const unsigned nb = (unsigned)ceil_div64(N, 256 * SP_ITEMS);
    k_sp_init<<<1, 256, 0, st>>>(w.params, w.prefix, w.hist);
    LAUNCH_CHECK();
    k_sp_count<<<nb < 512u ? nb : 512u, 256, 0, st>>>(A, w.params);
    LAUNCH_CHECK();
    k_sp_derive<<<1, 1, 0, st>>>(A, w.params, w.krem);
    LAUNCH_CHECK();
    int idx_bits = 1;
    while (((int64_t)1 << idx_bits) < N) ++idx_bits;
    const unsigned hist_wgs = getenv("NNDET_SP_HIST_WGS") ? (unsigned)atoi(getenv("NNDET_SP_HIST_WGS")) : 0u;     // (read per call: A/B, tests; 0 = one block per workgroup)
    const unsigned nb_hist = nb < hist_wgs || hist_wgs < 1 ? nb : hist_wgs;
    for (int shift = 56; shift >= 0; shift -= 8) {
        if (shift < 32 && shift >= idx_bits) continue;
        k_sp_hist<<<nb_hist, 256, 0, st>>>(A, w.prefix, w.hist, shift, w.krem + 2);
        LAUNCH_CHECK();
        k_sp_pick<<<1, 128, 0, st>>>(w.prefix, w.krem, w.hist, shift);
        LAUNCH_CHECK();
    }
    k_sp_fill<<<ceil_div(P_cap > POOL_cap ? P_cap : POOL_cap, 256), 256, 0, st>>>(w.list_pos, P_cap, w.list_pool, POOL_cap);
    LAUNCH_CHECK();
    k_sp_collect<<<nb, 256, 0, st>>>(A, w.prefix, w.params, w.list_pos, w.list_pool);
    LAUNCH_CHECK();
    const char* tf_env = getenv("NNDET_SP_TAIL_FUSED");          // (read per call: the tests run both paths in one process)
    const int tail_fused = tf_env ? atoi(tf_env) : 1;
    if (tail_fused && P_cap <= 4096 && POOL_cap <= 4096) {
        auto p2 = [](int n) { int v = 2; while (v < n) v <<= 1; return v; };
        const int NP2 = p2(P_cap), NQ2 = p2(POOL_cap), NN2 = p2(NEG_cap);
        const size_t lds = ((size_t)(NP2 > NQ2 ? NP2 : NQ2) + NQ2 + NN2) * 8;
        static NndetDevOnce attr;
        if (attr.need()) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sp_tail), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 4096 * 8));
            attr.done();
        }
        k_sp_tail<<<1, 1024, lds, st>>>(A, w.params, w.list_pos, w.list_pool, NP2, NQ2, NN2, pos_idx, neg_idx, counts);
        LAUNCH_CHECK();
        return 0;
    }
    size_t tmp = w.sort_tmp_bytes;
    // positives: everything collected IS the selection -> sort by anchor index only (low 32 bits; padding sorts last)
    HIP_TRY((rocprim::radix_sort_keys<rocprim::default_config, const u64*, u64*>(w.sort_tmp, tmp, w.list_pos, w.pos_sorted, (size_t)P_cap, 0, 32, st, false)));
    tmp = w.sort_tmp_bytes;
    HIP_TRY((rocprim::radix_sort_keys<rocprim::default_config, const u64*, u64*>(w.sort_tmp, tmp, w.list_pool, w.pool_sorted, (size_t)POOL_cap, 0, 64, st, false)));
    k_sp_poolkeys<<<ceil_div(POOL_cap, 256), 256, 0, st>>>(A, w.params, w.sel);
    LAUNCH_CHECK();
    tmp = w.sort_tmp_bytes;
    HIP_TRY((rocprim::radix_sort_keys<rocprim::default_config, const u64*, u64*>(w.sort_tmp, tmp, w.sel, w.sel_sorted, (size_t)POOL_cap, 0, 64, st, false)));
    k_sp_negidx<<<ceil_div(NEG_cap, 256), 256, 0, st>>>(A, w.params, w.pool_sorted, w.sel_sorted, w.neg_tmp);
    LAUNCH_CHECK();
    tmp = w.sort_tmp_bytes;
    HIP_TRY((rocprim::radix_sort_keys<rocprim::default_config, const u64*, u64*>(w.sort_tmp, tmp, w.neg_tmp, w.neg_sorted, (size_t)NEG_cap, 0, 33, st, false)));
    const int ne = P_cap > NEG_cap ? P_cap : NEG_cap;
    k_sp_emit<<<ceil_div(ne, 256), 256, 0, st>>>(A, w.params, w.pos_sorted, w.neg_sorted, pos_idx, neg_idx, counts);
    LAUNCH_CHECK();
    return 0;
}
