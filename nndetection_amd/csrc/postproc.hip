// Fused post-processing front end for gfx950 -- replaces, for a whole batch and without a host round trip,
//   DetectionHeadHNM.postprocess_for_inference (decode ALL anchors + sigmoid of ALL logits, nndet/arch/heads/comb.py:140-158)
//   BaseRetinaNet.postprocess_detections_single_image (clip, full descending sort, top-k, score threshold, small-box filter,
//   batched NMS, top detections_per_img; nndet/core/retina.py:332-379)
// The reference decodes 1.19 M anchors per image and sorts 1.19 M scores to keep 10 000 of them. Here:
//   1. exact radix SELECT (radix_select.h) of the K = min(topk, M) best (score desc, flat index asc) candidates per image
//      straight from the logits: sigmoid is recomputed per pass (4.7 MB of logits per image and pass), nothing is decoded;
//   2. the K selected 64-bit keys are collected and sorted (rocPRIM segmented radix sort, K <= 10 000 per image);
//   3. k_pp_finish (one workgroup per image): decode + clip ONLY the K survivors, score threshold, small-box filter,
//      order-preserving compaction, max coordinate of the survivors (the class offset of batched_nms, nms.py:103-105);
//   4. class offsets, then the mask + scan NMS of nms3d.hip on the already sorted boxes (no second sort, no .item());
//   5. gather of the first detections_per_img kept boxes / scores / labels into fixed-capacity outputs + a count per image.
// Arithmetic of decode / clip / sigmoid / IoU is the reference's expression order (compile with -ffp-contract=off).
#include "common.h"
#include "radix_select.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

size_t nms_presorted_workspace_bytes(int64_t n_cap);
size_t nms_presorted_batched_workspace_bytes(int64_t n_cap, int B);
int nms_presorted_batched_run(const float* boxes, int64_t n_cap, int B, const int64_t* n_valid_dev, float thr, int64_t* keep_out,
                              int64_t* n_keep_out, void* workspace, size_t workspace_bytes, hipStream_t st);
int nms_presorted_run(const float* boxes, int64_t n_cap, const int64_t* n_valid_dev, float thr, int64_t* keep_out,
                      int64_t* n_keep_out, void* workspace, size_t workspace_bytes, hipStream_t st);

struct PpArgs {
    const float* scores;    // [B, M*C] logits (is_prob == 0) or probabilities (is_prob == 1)
    const float* deltas;    // [B, M, 6] regression deltas (anchors != NULL) or already decoded boxes (anchors == NULL)
    const float* anchors;   // [M, 6] shared by all images, or NULL
    const int32_t* labels;  // [B, M] explicit class per row (C must be 1), or NULL: label = flat index % C
    int32_t B, C, K, is_prob;
    int64_t M, MC;
    float clip_exp, ix, iy, iz;       // ix <= 0: no clipping
    float score_thresh; int32_t use_thresh;
    float min_size; int32_t use_min_size;
    int32_t max_det;
};

#define PP_ITEMS 8   // candidates per thread in the select passes

__device__ __forceinline__ float pp_score(const PpArgs& A, float v) {
    // torch.sigmoid's fp32 expression (1 / (1 + exp(-x))); the SAME code in every pass, so keys are consistent
    return A.is_prob ? v : 1.f / (1.f + expf(-v));
}
// ascending key order == (score descending, flat index ascending); NaN scores sort first, like torch.sort(descending=True)
__device__ __forceinline__ u64 pp_key(float score, uint32_t idx) {
    return ((u64)(~f32_sortable(score)) << 32) | (u64)idx;
}

__global__ void k_pp_init(int B, int K, u64* prefix, int* krem, int* cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    prefix[i] = 0; krem[i] = K; krem[B + i] = 0; cnt[i] = 0;
}

// grid (ceil(MC / (256 * PP_ITEMS)), B)
__global__ __launch_bounds__(256) void k_pp_hist(PpArgs A, const u64* __restrict__ prefix, unsigned* __restrict__ hist, int shift,
                                                 const int* __restrict__ done) {
    __shared__ unsigned h[256];
    const int b = blockIdx.y;
    if (done[b]) return;                         // the K-th key of this image is decided (scores are nearly unique: after the 4 score passes)
    h[threadIdx.x] = 0;
    __syncthreads();
    const u64 pre = prefix[b];
    const float* s = A.scores + (int64_t)b * A.MC;
    const int64_t i0 = ((int64_t)blockIdx.x * 256) * PP_ITEMS + threadIdx.x;
#pragma unroll
    for (int t = 0; t < PP_ITEMS; ++t) {
        const int64_t i = i0 + (int64_t)t * 256;
        if (i < A.MC) {
            const u64 key = pp_key(pp_score(A, s[i]), (uint32_t)i);
            const bool match = (shift >= 56) || ((key >> (shift + 8)) == (pre >> (shift + 8)));
            if (match) atomicAdd(&h[(unsigned)(key >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    const unsigned v = h[threadIdx.x];
    if (v) atomicAdd(&hist[(int64_t)b * 256 + threadIdx.x], v);
}

__global__ __launch_bounds__(256) void k_pp_pick(int B, u64* __restrict__ prefix, int* __restrict__ krem,
                                                 unsigned* __restrict__ hist, int shift) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    if (krem[B + i]) return;                     // decided in an earlier pass: its histogram stayed empty (radix_select.h)
    radix_pick_wave(prefix + i, krem + i, hist + (int64_t)i * 256, shift, threadIdx.x & 63, krem + B + i);
}

// every key <= the K-th smallest goes into the candidate list (exactly K per image: keys are distinct).
// ONE global atomic per workgroup (round 4): the K = 10 000 returning adds per image onto the same counter serialise in L2 at ~30 ns
// each -- 250-330 us for a 19 MB pass whose histogram twin takes 14 us; the hits are sparse (1 in ~150), so aggregating per wave does
// not help. Threads rank their hits with an LDS atomic, thread 0 reserves the block's slots.
__global__ __launch_bounds__(256) void k_pp_collect(PpArgs A, const u64* __restrict__ kth, int* __restrict__ cnt, u64* __restrict__ cand) {
    __shared__ int lcnt, lbase;
    const int b = blockIdx.y;
    const u64 lim = kth[b];
    const float* s = A.scores + (int64_t)b * A.MC;
    const int64_t i0 = ((int64_t)blockIdx.x * 256) * PP_ITEMS + threadIdx.x;
    if (threadIdx.x == 0) lcnt = 0;
    __syncthreads();
    u64 keys[PP_ITEMS];
    unsigned hits = 0;
#pragma unroll
    for (int t = 0; t < PP_ITEMS; ++t) {
        const int64_t i = i0 + (int64_t)t * 256;
        keys[t] = 0;
        if (i < A.MC) {
            keys[t] = pp_key(pp_score(A, s[i]), (uint32_t)i);
            if (keys[t] <= lim) hits |= 1u << t;
        }
    }
    int lofs = 0;
    if (hits) lofs = atomicAdd(&lcnt, __popc(hits));
    __syncthreads();
    if (lcnt == 0) return;
    if (threadIdx.x == 0) lbase = atomicAdd(&cnt[b], lcnt);
    __syncthreads();
    if (!hits) return;
    int pos = lbase + lofs;
#pragma unroll
    for (int t = 0; t < PP_ITEMS; ++t)
        if (hits & (1u << t)) {
            if (pos < A.K) cand[(int64_t)b * A.K + pos] = keys[t];
            ++pos;
        }
}

// Sort of the K <= 16 384 candidate keys of an image, ascending (= descending score, ties by index): ONE workgroup per image, bitonic
// network in 128 KB of LDS (round 4). rocprim's segmented radix sort gives each of the B segments to one block and walks 8 digit passes
// over it: 350 us for 4 x 10 000 keys -- a fifth of the whole post-processing. N = K rounded up to a power of two, padded with ~0.
__global__ __launch_bounds__(1024) void k_pp_bitonic(const u64* __restrict__ cand, u64* __restrict__ out, int K, int N) {
    extern __shared__ __attribute__((aligned(16))) char pp_smem[];
    u64* s = reinterpret_cast<u64*>(pp_smem);
    const int tid = threadIdx.x;
    const u64* src = cand + (int64_t)blockIdx.x * K;
    for (int i = tid; i < N; i += 1024) s[i] = i < K ? src[i] : ~0ULL;
    __syncthreads();
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (N >> 1); t += 1024) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // pair index t -> element with bit j clear
                const int l = i | j;
                const u64 a = s[i], b = s[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { s[i] = b; s[l] = a; }
            }
            __syncthreads();
        }
    }
    u64* dst = out + (int64_t)blockIdx.x * K;
    for (int i = tid; i < K; i += 1024) dst[i] = s[i];
}

__global__ void k_pp_offsets(int B, int K, int* off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= B) off[i] = i * K;
}

// One workgroup (1024 threads) per image over its K sorted candidates: decode + clip, filters, ordered compaction.
__global__ __launch_bounds__(1024) void k_pp_finish(PpArgs A, const u64* __restrict__ cand, float* __restrict__ cboxes,
                                                    float* __restrict__ cscores, int32_t* __restrict__ clabels,
                                                    uint32_t* __restrict__ cidx, int64_t* __restrict__ n_valid, float* __restrict__ maxc) {
    __shared__ int wsum[16];
    __shared__ float wmax[16];
    __shared__ int running;
    __shared__ float runmax;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) { running = 0; runmax = -INFINITY; }
    __syncthreads();
    for (int base = 0; base < A.K; base += 1024) {
        const int i = base + tid;
        bool ok = false;
        float bx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float sc = 0.f;
        int lab = 0;
        uint32_t idx = 0;
        if (i < A.K) {
            const u64 key = cand[(int64_t)b * A.K + i];
            idx = (uint32_t)key;
            sc = f32_unsortable(~(uint32_t)(key >> 32));
            const int64_t a = idx / (uint32_t)A.C;
            lab = A.labels ? A.labels[(int64_t)b * A.M + a] : (int)(idx - (uint32_t)a * (uint32_t)A.C);
            const float* r = A.deltas + ((int64_t)b * A.M + a) * 6;
            if (A.anchors) {
                // decode_single, nndet/core/boxes/coder.py:107-151 with unit weights (x / 1 == x)
                const float* an = A.anchors + a * 6;
                const float wd = an[2] - an[0], ht = an[3] - an[1], dp = an[5] - an[4];
                const float cx = an[0] + 0.5f * wd, cy = an[1] + 0.5f * ht, cz = an[4] + 0.5f * dp;
                const float dw = fminf(r[2], A.clip_exp), dh = fminf(r[3], A.clip_exp), dd = fminf(r[5], A.clip_exp);
                const float pcx = r[0] * wd + cx, pcy = r[1] * ht + cy, pcz = r[4] * dp + cz;
                const float pw = expf(dw) * wd, ph = expf(dh) * ht, pd = expf(dd) * dp;
                bx[0] = pcx - 0.5f * pw; bx[1] = pcy - 0.5f * ph; bx[2] = pcx + 0.5f * pw; bx[3] = pcy + 0.5f * ph;
                bx[4] = pcz - 0.5f * pd; bx[5] = pcz + 0.5f * pd;
            } else {
#pragma unroll
                for (int q = 0; q < 6; ++q) bx[q] = r[q];
            }
            if (A.ix > 0.f) {   // clip_boxes_to_image_3d_, nndet/core/boxes/clip.py:95-100
                bx[0] = fminf(fmaxf(bx[0], 0.f), A.ix); bx[2] = fminf(fmaxf(bx[2], 0.f), A.ix);
                bx[1] = fminf(fmaxf(bx[1], 0.f), A.iy); bx[3] = fminf(fmaxf(bx[3], 0.f), A.iy);
                bx[4] = fminf(fmaxf(bx[4], 0.f), A.iz); bx[5] = fminf(fmaxf(bx[5], 0.f), A.iz);
            }
            ok = !A.use_thresh || (sc > A.score_thresh);                       // retina.py:358-360 (NaN > t is false)
            if (A.use_min_size)                                                  // remove_small_boxes, ops.py:241-259
                ok = ok && (bx[2] - bx[0] >= A.min_size) && (bx[3] - bx[1] >= A.min_size) && (bx[5] - bx[4] >= A.min_size);
        }
        const unsigned long long vote = __ballot(ok);
        const int before = __popcll(vote & ((1ULL << lane) - 1ULL));
        if (lane == 0) wsum[w] = __popcll(vote);
        float m = -INFINITY;
        if (ok) m = fmaxf(fmaxf(fmaxf(bx[0], bx[1]), fmaxf(bx[2], bx[3])), fmaxf(bx[4], bx[5]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) wmax[w] = m;
        __syncthreads();
        int woff = 0, tot = 0;
        float bm = -INFINITY;
        for (int k = 0; k < 16; ++k) { if (k < w) woff += wsum[k]; tot += wsum[k]; bm = fmaxf(bm, wmax[k]); }
        if (ok) {
            const int64_t pos = (int64_t)b * A.K + running + woff + before;
#pragma unroll
            for (int q = 0; q < 6; ++q) cboxes[pos * 6 + q] = bx[q];
            cscores[pos] = sc;
            clabels[pos] = lab;
            cidx[pos] = idx;
        }
        __syncthreads();
        if (tid == 0) { running += tot; runmax = fmaxf(runmax, bm); }
        __syncthreads();
    }
    if (tid == 0) { n_valid[b] = running; maxc[b] = runmax; }
}

// batched_nms' coordinate trick (nndet/core/boxes/nms.py:101-106): box + label * (max_coordinate + 1) in fp32; padding rows
// (>= n_valid) become all-zero boxes. grid (ceil(K / 256), B)
__global__ void k_pp_offset_boxes(int K, const float* __restrict__ cboxes, const int32_t* __restrict__ clabels,
                                  const int64_t* __restrict__ n_valid, const float* __restrict__ maxc, float* __restrict__ nboxes) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const int64_t p = (int64_t)b * K + i;
    if (i < n_valid[b]) {
        const float off = (float)clabels[p] * (maxc[b] + 1.f);
#pragma unroll
        for (int q = 0; q < 6; ++q) nboxes[p * 6 + q] = cboxes[p * 6 + q] + off;
    } else {
#pragma unroll
        for (int q = 0; q < 6; ++q) nboxes[p * 6 + q] = 0.f;
    }
}

// grid (B), block 256
__global__ void k_pp_gather(int K, int max_det, const int64_t* __restrict__ keep, const int64_t* __restrict__ n_keep,
                            const float* __restrict__ cboxes, const float* __restrict__ cscores, const int32_t* __restrict__ clabels,
                            const uint32_t* __restrict__ cidx, float* __restrict__ out_boxes, float* __restrict__ out_scores,
                            int64_t* __restrict__ out_labels, int64_t* __restrict__ out_index, int64_t* __restrict__ out_counts) {
    const int b = blockIdx.x;
    const int64_t nk = n_keep[b] < max_det ? n_keep[b] : max_det;
    for (int j = threadIdx.x; j < max_det; j += blockDim.x) {
        const int64_t o = (int64_t)b * max_det + j;
        if (j < nk) {
            const int64_t p = (int64_t)b * K + keep[(int64_t)b * K + j];
#pragma unroll
            for (int q = 0; q < 6; ++q) out_boxes[o * 6 + q] = cboxes[p * 6 + q];
            out_scores[o] = cscores[p];
            out_labels[o] = (int64_t)clabels[p];
            if (out_index) out_index[o] = (int64_t)cidx[p];
        } else {
#pragma unroll
            for (int q = 0; q < 6; ++q) out_boxes[o * 6 + q] = 0.f;
            out_scores[o] = 0.f;
            out_labels[o] = -1;
            if (out_index) out_index[o] = -1;
        }
    }
    if (threadIdx.x == 0) out_counts[b] = nk;
}

struct PpWs {
    u64* prefix; int* krem; int* cnt; unsigned* hist; int* seg_off;
    u64* cand; u64* cand_sorted; float* cboxes; float* cscores; int32_t* clabels; uint32_t* cidx; float* nboxes;
    int64_t* n_valid; float* maxc; int64_t* keep; int64_t* n_keep;
    void* sort_tmp; size_t sort_tmp_bytes; void* nms_ws; size_t nms_ws_bytes; size_t total;
};

static int pp_layout(int B, int K, char* base, PpWs* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t BK = (size_t)B * K;
    size_t o_p = take((size_t)B * 8), o_k = take((size_t)B * 8) /* ranks + "decided" flags */, o_c = take((size_t)B * 4), o_h = take((size_t)B * 256 * 4);
    size_t o_so = take((size_t)(B + 1) * 4);
    size_t o_ca = take(BK * 8), o_cs = take(BK * 8), o_cb = take(BK * 24), o_sc = take(BK * 4), o_cl = take(BK * 4), o_ci = take(BK * 4), o_nb = take(BK * 24);
    size_t o_nv = take((size_t)B * 8), o_mx = take((size_t)B * 4), o_kp = take(BK * 8), o_nk = take((size_t)B * 8);
    size_t tmp = 0;
    hipError_t e = rocprim::segmented_radix_sort_keys<rocprim::default_config, const u64*, u64*, const int*>(
        nullptr, tmp, nullptr, nullptr, (unsigned)BK, (unsigned)B, nullptr, nullptr, 0, 64, (hipStream_t)0, false);
    if (e != hipSuccess) return (int)e;
    size_t o_tmp = take(tmp > 0 ? tmp : 256);
    // (round 4) the NMS of all B images runs in the same launches: B x the single-image workspace (NNDET_PP_BATCHED_NMS=0: one by one)
    size_t nms_b = nms_presorted_workspace_bytes(K);
    if (nms_b == 0) return NNDET_EINVAL;
    { const size_t bb = nms_presorted_batched_workspace_bytes(K, B); if (bb > nms_b) nms_b = bb; }
    size_t o_nms = take(nms_b);
    w->sort_tmp_bytes = tmp; w->nms_ws_bytes = nms_b; w->total = off;
    if (!base) return 0;                       // size query: no pointer arithmetic on a null base (UBSan, round 6)
    w->prefix = (u64*)(base + o_p); w->krem = (int*)(base + o_k); w->cnt = (int*)(base + o_c); w->hist = (unsigned*)(base + o_h);
    w->seg_off = (int*)(base + o_so);
    w->cand = (u64*)(base + o_ca); w->cand_sorted = (u64*)(base + o_cs); w->cboxes = (float*)(base + o_cb);
    w->cscores = (float*)(base + o_sc); w->clabels = (int32_t*)(base + o_cl); w->cidx = (uint32_t*)(base + o_ci); w->nboxes = (float*)(base + o_nb);
    w->n_valid = (int64_t*)(base + o_nv); w->maxc = (float*)(base + o_mx); w->keep = (int64_t*)(base + o_kp); w->n_keep = (int64_t*)(base + o_nk);
    w->sort_tmp = base + o_tmp; w->sort_tmp_bytes = tmp; w->nms_ws = base + o_nms; w->nms_ws_bytes = nms_b; w->total = off;
    return 0;
}

static int pp_K(int64_t M, int32_t C, int32_t topk, int64_t* K) {
    const int64_t MC = M * C;
    if (M <= 0 || C <= 0 || MC >= (1LL << 32)) return NNDET_EINVAL;
    int64_t k = topk > 0 ? (topk < M ? topk : M) : MC;      // retina.py:351-354: min(topk_candidates, boxes.size(0))
    if (k > MC) k = MC;
    if (k >= (1 << 24)) return NNDET_EINVAL;                  // one workgroup per image finishes the K survivors
    *K = k;
    return 0;
}

extern "C" size_t nndet_postprocess3d_workspace_bytes(int32_t B, int64_t M, int32_t C, int32_t topk) {
    int64_t K;
    if (B <= 0 || pp_K(M, C, topk, &K)) return 0;
    PpWs w;
    if (pp_layout(B, (int)K, nullptr, &w)) return 0;
    return w.total;
}

static int pp_run(const float* scores, int32_t scores_are_probs, const float* deltas, const float* anchors, const int32_t* labels,
                  int32_t B, int64_t M, int32_t C, float clip_exp, float img_x, float img_y, float img_z,
                  int32_t topk, float score_thresh, int32_t use_score_thresh, float min_size,
                  int32_t use_min_size, float nms_thresh, int32_t max_det, float* out_boxes,
                  float* out_scores, int64_t* out_labels, int64_t* out_index, int64_t* out_counts, void* workspace,
                  size_t workspace_bytes, void* stream);

extern "C" int nndet_postprocess3d_f32(const float* scores, int32_t scores_are_probs, const float* deltas, const float* anchors,
                                       int32_t B, int64_t M, int32_t C, float clip_exp, float img_x, float img_y, float img_z,
                                       int32_t topk, float score_thresh, int32_t use_score_thresh, float min_size,
                                       int32_t use_min_size, float nms_thresh, int32_t max_det, float* out_boxes,
                                       float* out_scores, int64_t* out_labels, int64_t* out_counts, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    return pp_run(scores, scores_are_probs, deltas, anchors, nullptr, B, M, C, clip_exp, img_x, img_y, img_z, topk, score_thresh,
                  use_score_thresh, min_size, use_min_size, nms_thresh, max_det, out_boxes, out_scores, out_labels, nullptr, out_counts,
                  workspace, workspace_bytes, stream);
}

extern "C" int nndet_postprocess3d_rows_f32(const float* probs, const float* boxes, const int32_t* labels, int32_t B, int64_t M,
                                            float img_x, float img_y, float img_z, int32_t topk, float score_thresh,
                                            int32_t use_score_thresh, float min_size, int32_t use_min_size, float nms_thresh,
                                            int32_t max_det, float* out_boxes, float* out_scores, int64_t* out_labels,
                                            int64_t* out_index, int64_t* out_counts, void* workspace, size_t workspace_bytes,
                                            void* stream) {
    if (!labels || !out_index) return NNDET_EINVAL;
    return pp_run(probs, 1, boxes, nullptr, labels, B, M, 1, 0.f, img_x, img_y, img_z, topk, score_thresh, use_score_thresh, min_size,
                  use_min_size, nms_thresh, max_det, out_boxes, out_scores, out_labels, out_index, out_counts, workspace,
                  workspace_bytes, stream);
}

static int pp_run(const float* scores, int32_t scores_are_probs, const float* deltas, const float* anchors, const int32_t* labels,
                  int32_t B, int64_t M, int32_t C, float clip_exp, float img_x, float img_y, float img_z,
                  int32_t topk, float score_thresh, int32_t use_score_thresh, float min_size,
                  int32_t use_min_size, float nms_thresh, int32_t max_det, float* out_boxes,
                  float* out_scores, int64_t* out_labels, int64_t* out_index, int64_t* out_counts, void* workspace,
                  size_t workspace_bytes, void* stream) {
    hipStream_t st = as_stream(stream);
    int64_t K64;
    if (B <= 0 || max_det <= 0 || !out_counts) return NNDET_EINVAL;
    int rc = pp_K(M, C, topk, &K64);
    if (rc) return rc;
    if (!scores || !deltas || !out_boxes || !out_scores || !out_labels || !workspace) return NNDET_EINVAL;
    const int K = (int)K64;
    PpWs w;
    rc = pp_layout(B, K, (char*)workspace, &w);
    if (rc) return rc;
    if (w.total > workspace_bytes) return NNDET_EWORKSPACE;
    PpArgs A;
    A.scores = scores; A.deltas = deltas; A.anchors = anchors; A.labels = labels; A.B = B; A.C = C; A.K = K; A.is_prob = scores_are_probs;
    A.M = M; A.MC = M * C; A.clip_exp = clip_exp; A.ix = img_x; A.iy = img_y; A.iz = img_z;
    A.score_thresh = score_thresh; A.use_thresh = use_score_thresh; A.min_size = min_size; A.use_min_size = use_min_size;
    A.max_det = max_det;
    k_pp_init<<<ceil_div(B, 64), 64, 0, st>>>(B, K, w.prefix, w.krem, w.cnt);
    LAUNCH_CHECK();
    HIP_TRY(hipMemsetAsync(w.hist, 0, (size_t)B * 256 * 4, st));
    const dim3 grid((unsigned)ceil_div64(A.MC, 256 * PP_ITEMS), B);
    if ((int64_t)K < A.MC) {
        int idx_bits = 1;
        while (((int64_t)1 << idx_bits) < A.MC) ++idx_bits;
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (shift < 32 && shift >= idx_bits) continue;   // index digits above the highest set bit are zero for every key
            k_pp_hist<<<grid, 256, 0, st>>>(A, w.prefix, w.hist, shift, w.krem + B);
            LAUNCH_CHECK();
            k_pp_pick<<<ceil_div(B, 4), 256, 0, st>>>(B, w.prefix, w.krem, w.hist, shift);
            LAUNCH_CHECK();
        }
    } else {
        HIP_TRY(hipMemsetAsync(w.prefix, 0xFF, (size_t)B * 8, st));   // everything is a candidate
    }
    k_pp_collect<<<grid, 256, 0, st>>>(A, w.prefix, w.cnt, w.cand);
    LAUNCH_CHECK();
    static const int bitonic = getenv("NNDET_PP_BITONIC") ? atoi(getenv("NNDET_PP_BITONIC")) : 1;
    if (bitonic && K <= 16384) {
        int N = 2;
        while (N < K) N <<= 1;
        static NndetDevOnce attr;
        if (attr.need()) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pp_bitonic), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
            attr.done();
        }
        k_pp_bitonic<<<B, 1024, (size_t)N * 8, st>>>(w.cand, w.cand_sorted, (int)K, N);
        LAUNCH_CHECK();
    } else {
        k_pp_offsets<<<ceil_div(B + 1, 64), 64, 0, st>>>(B, K, w.seg_off);
        LAUNCH_CHECK();
        size_t tmp = w.sort_tmp_bytes;
        HIP_TRY((rocprim::segmented_radix_sort_keys<rocprim::default_config, const u64*, u64*, const int*>(
            w.sort_tmp, tmp, w.cand, w.cand_sorted, (unsigned)((size_t)B * K), (unsigned)B, w.seg_off, w.seg_off + 1, 0, 64, st, false)));
    }
    k_pp_finish<<<B, 1024, 0, st>>>(A, w.cand_sorted, w.cboxes, w.cscores, w.clabels, w.cidx, w.n_valid, w.maxc);
    LAUNCH_CHECK();
    k_pp_offset_boxes<<<dim3(ceil_div(K, 256), B), 256, 0, st>>>(K, w.cboxes, w.clabels, w.n_valid, w.maxc, w.nboxes);
    LAUNCH_CHECK();
    static const int batched_nms = getenv("NNDET_PP_BATCHED_NMS") ? atoi(getenv("NNDET_PP_BATCHED_NMS")) : 1;
    if (batched_nms && B > 1) {
        rc = nms_presorted_batched_run(w.nboxes, K, B, w.n_valid, nms_thresh, w.keep, w.n_keep, w.nms_ws, w.nms_ws_bytes, st);
        if (rc) return rc;
    } else {
        for (int b = 0; b < B; ++b) {
            rc = nms_presorted_run(w.nboxes + (size_t)b * K * 6, K, w.n_valid + b, nms_thresh, w.keep + (size_t)b * K, w.n_keep + b,
                                   w.nms_ws, w.nms_ws_bytes, st);
            if (rc) return rc;
        }
    }
    k_pp_gather<<<B, 256, 0, st>>>(K, max_det, w.keep, w.n_keep, w.cboxes, w.cscores, w.clabels, w.cidx, out_boxes, out_scores,
                                   out_labels, out_index, out_counts);
    LAUNCH_CHECK();
    return 0;
}
