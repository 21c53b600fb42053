// Weighted box clustering for gfx950 -- replaces nndet.inference.detection.wbc.wbc / batched_wbc
// (nndet/inference/detection/wbc.py:22-160,163-199), the ensembling step that merges the predictions of several models /
// tiles / test-time augmentations (nndet/inference/ensembler/detection.py:166-217,476-537). The reference materialises the
// full [N, N] IoU matrix and runs a Python while-loop with torch.where per cluster.
//
// The clustering recurrence is the one of greedy NMS: walking the boxes by descending score, a box founds a cluster iff no
// earlier cluster HEAD has IoU > thr with it (wbc.py:120-144); every other box joins the FIRST head that has IoU > thr with it.
// So: (1) stable descending sort, (2) the upper-triangle IoU bit mask and (3) the on-device greedy scan of nms3d.hip give the
// heads; (4) k_wbc_assign: every head atomically min-writes its sorted position into the boxes of its mask row -> first head
// per box; (5) the members are grouped by head with one key sort (head << 32 | position) and (6) one thread per cluster adds
// its members up in pool order (wbc.py:163-199: score = sum(iou w s) / (sum(iou w) + n_missing * mean(iou w) * missing_weight),
// box = sum(box * iou w s) / sum(iou w s)); clusters with score <= score_thresh are dropped, the rest is emitted in cluster order.
// Per-class clustering (batched_wbc) = the same pass with the IoU bit masked by label equality; output sorted by (label,
// cluster order) like the reference's loop over labels.unique().
#include "common.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

typedef unsigned long long u64;

// nms3d.hip
size_t nms_presorted_workspace_bytes(int64_t n_cap);
int nms_heads_run(const float* sboxes, const int32_t* slabels, int64_t n, float thr, u64** mask_out, u64** keepbits_out,
                  void* workspace, size_t workspace_bytes, hipStream_t st);

__device__ __forceinline__ float wbc_iou(const float* a, const float* b) {   // box_iou_union_3d, nndet/core/boxes/ops.py:131-159
    const float va = (a[2] - a[0]) * (a[3] - a[1]) * (a[5] - a[4]);
    const float vb = (b[2] - b[0]) * (b[3] - b[1]) * (b[5] - b[4]);
    const float x1 = fmaxf(a[0], b[0]), y1 = fmaxf(a[1], b[1]), x2 = fminf(a[2], b[2]), y2 = fminf(a[3], b[3]);
    const float z1 = fmaxf(a[4], b[4]), z2 = fminf(a[5], b[5]);
    const float inter = fmaxf(x2 - x1, 0.f) * fmaxf(y2 - y1, 0.f) * fmaxf(z2 - z1, 0.f);
    return inter / ((va + vb) - inter);
}

__global__ void k_wbc_prep(const float* __restrict__ scores, const int64_t* __restrict__ labels, int64_t n, float* __restrict__ skey,
                           int32_t* __restrict__ iota) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    skey[i] = scores[i];
    iota[i] = (int32_t)i;
}

__global__ void k_wbc_gather(const float* __restrict__ boxes, const int64_t* __restrict__ labels, const int32_t* __restrict__ order,
                             int64_t n, float* __restrict__ sboxes, int32_t* __restrict__ slabels, uint32_t* __restrict__ assign) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t s = order[i];
#pragma unroll
    for (int q = 0; q < 6; ++q) sboxes[i * 6 + q] = boxes[s * 6 + q];
    slabels[i] = labels ? (int32_t)labels[s] : 0;
    assign[i] = 0xffffffffu;
}

// grid (col_blocks, n_heads_upper_bound = col_blocks * 64 / 4): one wave per row i; only heads do work
__global__ __launch_bounds__(256) void k_wbc_assign(const u64* __restrict__ mask, const u64* __restrict__ keepbits, int64_t n,
                                                    int col_blocks, uint32_t* __restrict__ assign) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    if (!((keepbits[i >> 6] >> (i & 63)) & 1ULL)) return;
    const int rb = (int)(i >> 6);
    for (int cb = rb + lane; cb < col_blocks; cb += 64) {
        u64 w = mask[i * col_blocks + cb];
        if (cb == rb) w &= ~((2ULL << (i & 63)) - 1ULL);   // diagonal tiles carry both directions: only LOWER-scored boxes can join
        while (w) {
            const int b = __ffsll((long long)w) - 1;
            w &= w - 1;
            atomicMin(&assign[(int64_t)cb * 64 + b], (uint32_t)i);
        }
    }
}

__global__ void k_wbc_keys(const uint32_t* __restrict__ assign, int64_t n, u64* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = assign[i];
    if (h == 0xffffffffu) h = (uint32_t)i;        // a head (or a box whose IoU with everything is NaN): its own cluster
    keys[i] = ((u64)h << 32) | (u64)(uint32_t)i;
}

// one thread per sorted key position that STARTS a cluster (head == position): sequential fp32 sums over its members in pool order
__global__ void k_wbc_consolidate(const u64* __restrict__ keys, int64_t n, const float* __restrict__ sboxes,
                                  const int32_t* __restrict__ order, const float* __restrict__ scores, const float* __restrict__ weights,
                                  const float* __restrict__ n_exp, const u64* __restrict__ keepbits, int use_area, float missing_weight,
                                  float iou_thresh, float score_thresh, float* __restrict__ cbox, float* __restrict__ cscore, int32_t* __restrict__ cvalid) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t h = (uint32_t)(keys[k] >> 32);
    if (k > 0 && (uint32_t)(keys[k - 1] >> 32) == h) return;                  // not the first member of its cluster
    // clusters are only founded by heads of the scan; a box with NaN IoU against its would-be head is neither matched nor
    // kept in the reference's pool (wbc.py:122,141: both comparisons are false) -- it simply disappears. Same here.
    const bool is_head = (keepbits[h >> 6] >> (h & 63)) & 1ULL;
    cvalid[h] = 0;
    if (!is_head) return;
    const float* hb = sboxes + (int64_t)h * 6;
    float s_w = 0.f, s_ws = 0.f, s_nexp = 0.f, bx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    for (int64_t m = k; m < n && (uint32_t)(keys[m] >> 32) == h; ++m) {
        const uint32_t j = (uint32_t)keys[m];
        const int64_t src = order[j];
        const float* jb = sboxes + (int64_t)j * 6;
        const float iou = wbc_iou(hb, jb);
        if (!(iou > iou_thresh)) continue;                // NaN IoU: left the pool without joining the cluster (wbc.py:122,141)
        float w = weights[src];
        if (use_area) w = w * ((jb[2] - jb[0]) * (jb[3] - jb[1]) * (jb[5] - jb[4]));   // box_area (wbc.py:110-112)
        const float msw = iou * w;                        // match_score_weights (wbc.py:186)
        const float ms = msw * scores[src];               // match_scores
        s_w += msw; s_ws += ms; s_nexp += n_exp[src];
#pragma unroll
        for (int q = 0; q < 6; ++q) bx[q] += jb[q] * ms;
        ++cnt;
    }
    const float n_expected = s_nexp / (float)cnt;
    const float n_missing = fmaxf(0.f, n_expected - (float)cnt);
    const float denom = s_w + n_missing * (s_w / (float)cnt) * missing_weight;
    const float sc = s_ws / denom;
#pragma unroll
    for (int q = 0; q < 6; ++q) cbox[(int64_t)h * 6 + q] = bx[q] / s_ws;
    cscore[h] = sc;
    cvalid[h] = sc > score_thresh ? 1 : 0;
}

// single block: ordered compaction of the valid clusters (cluster order = head position = descending head score)
__global__ __launch_bounds__(1024) void k_wbc_emit(int64_t n, const int32_t* __restrict__ cvalid, const float* __restrict__ cbox,
                                                   const float* __restrict__ cscore, const int32_t* __restrict__ slabels,
                                                   const u64* __restrict__ keepbits, float* __restrict__ out_boxes,
                                                   float* __restrict__ out_scores, int64_t* __restrict__ out_labels,
                                                   int64_t* __restrict__ out_count) {
    __shared__ int wsum[16];
    __shared__ int running;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) running = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + tid;
        const bool ok = i < n && ((keepbits[i >> 6] >> (i & 63)) & 1ULL) && cvalid[i];
        const unsigned long long vote = __ballot(ok);
        const int before = __popcll(vote & ((1ULL << lane) - 1ULL));
        if (lane == 0) wsum[w] = __popcll(vote);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int k = 0; k < 16; ++k) { if (k < w) woff += wsum[k]; tot += wsum[k]; }
        if (ok) {
            const int64_t p = running + woff + before;
#pragma unroll
            for (int q = 0; q < 6; ++q) out_boxes[p * 6 + q] = cbox[i * 6 + q];
            out_scores[p] = cscore[i];
            out_labels[p] = (int64_t)slabels[i];
        }
        __syncthreads();
        if (tid == 0) running += tot;
        __syncthreads();
    }
    if (tid == 0) *out_count = running;
}

struct WbcWs {
    float* skey; float* skey_out; int32_t* iota; int32_t* order; float* sboxes; int32_t* slabels; uint32_t* assign;
    u64* keys; u64* keys_sorted; float* cbox; float* cscore; int32_t* cvalid;
    void* sort_tmp; size_t sort_tmp_bytes; void* nms_ws; size_t nms_ws_bytes; size_t total;
};

static int wbc_layout(int64_t n, char* base, WbcWs* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    size_t o1 = take(n * 4), o2 = take(n * 4), o3 = take(n * 4), o4 = take(n * 4), o5 = take(n * 24), o6 = take(n * 4), o7 = take(n * 4);
    size_t o8 = take(n * 8), o9 = take(n * 8), o10 = take(n * 24), o11 = take(n * 4), o12 = take(n * 4);
    size_t t1 = 0, t2 = 0;
    hipError_t e = rocprim::radix_sort_pairs_desc<rocprim::default_config, const float*, float*, const int32_t*, int32_t*>(
        nullptr, t1, nullptr, nullptr, nullptr, nullptr, (size_t)n, 0, 32, (hipStream_t)0, false);
    if (e != hipSuccess) return (int)e;
    e = rocprim::radix_sort_keys<rocprim::default_config, const u64*, u64*>(nullptr, t2, nullptr, nullptr, (size_t)n, 0, 64, (hipStream_t)0, false);
    if (e != hipSuccess) return (int)e;
    const size_t tmp = t1 > t2 ? t1 : t2;
    size_t o_tmp = take(tmp > 0 ? tmp : 256);
    const size_t nb = nms_presorted_workspace_bytes(n);
    if (nb == 0) return NNDET_EINVAL;
    size_t o_nms = take(nb);
    w->sort_tmp_bytes = tmp; w->nms_ws_bytes = nb; w->total = off;
    if (!base) return 0;                       // size query: no pointer arithmetic on a null base (UBSan, round 6)
    w->skey = (float*)(base + o1); w->skey_out = (float*)(base + o2); w->iota = (int32_t*)(base + o3); w->order = (int32_t*)(base + o4);
    w->sboxes = (float*)(base + o5); w->slabels = (int32_t*)(base + o6); w->assign = (uint32_t*)(base + o7);
    w->keys = (u64*)(base + o8); w->keys_sorted = (u64*)(base + o9); w->cbox = (float*)(base + o10); w->cscore = (float*)(base + o11);
    w->cvalid = (int32_t*)(base + o12);
    w->sort_tmp = base + o_tmp; w->sort_tmp_bytes = tmp; w->nms_ws = base + o_nms; w->nms_ws_bytes = nb; w->total = off;
    return 0;
}

extern "C" size_t nndet_wbc3d_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    WbcWs w;
    if (wbc_layout(n, nullptr, &w)) return 0;
    return w.total;
}

extern "C" int nndet_wbc3d_f32(const float* boxes, const float* scores, const int64_t* labels, const float* weights,
                               const float* n_exp_preds, int64_t n, float iou_thresh, float score_thresh, int32_t use_area,
                               float missing_weight, float* out_boxes, float* out_scores, int64_t* out_labels,
                               int64_t* out_count, void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = as_stream(stream);
    if (n < 0 || !out_count) return NNDET_EINVAL;
    if (n == 0) return (int)hipMemsetAsync(out_count, 0, 8, st);
    if (n >= (1LL << 31)) return NNDET_EINVAL;
    if (!boxes || !scores || !weights || !n_exp_preds || !out_boxes || !out_scores || !out_labels || !workspace) return NNDET_EINVAL;
    WbcWs w;
    int rc = wbc_layout(n, (char*)workspace, &w);
    if (rc) return rc;
    if (w.total > workspace_bytes) return NNDET_EWORKSPACE;
    const unsigned nb = (unsigned)ceil_div64(n, 256);
    k_wbc_prep<<<nb, 256, 0, st>>>(scores, labels, n, w.skey, w.iota);
    LAUNCH_CHECK();
    size_t tmp = w.sort_tmp_bytes;
    HIP_TRY((rocprim::radix_sort_pairs_desc<rocprim::default_config, const float*, float*, const int32_t*, int32_t*>(
        w.sort_tmp, tmp, w.skey, w.skey_out, w.iota, w.order, (size_t)n, 0, 32, st, false)));
    k_wbc_gather<<<nb, 256, 0, st>>>(boxes, labels, w.order, n, w.sboxes, w.slabels, w.assign);
    LAUNCH_CHECK();
    u64 *mask = nullptr, *keepbits = nullptr;
    rc = nms_heads_run(w.sboxes, labels ? w.slabels : nullptr, n, iou_thresh, &mask, &keepbits, w.nms_ws, w.nms_ws_bytes, st);
    if (rc) return rc;
    const int cb = (int)ceil_div64(n, 64);
    k_wbc_assign<<<(unsigned)ceil_div64(n, 4), 256, 0, st>>>(mask, keepbits, n, cb, w.assign);
    LAUNCH_CHECK();
    k_wbc_keys<<<nb, 256, 0, st>>>(w.assign, n, w.keys);
    LAUNCH_CHECK();
    tmp = w.sort_tmp_bytes;
    HIP_TRY((rocprim::radix_sort_keys<rocprim::default_config, const u64*, u64*>(w.sort_tmp, tmp, w.keys, w.keys_sorted, (size_t)n, 0, 64, st, false)));
    k_wbc_consolidate<<<nb, 256, 0, st>>>(w.keys_sorted, n, w.sboxes, w.order, scores, weights, n_exp_preds, keepbits, use_area,
                                          missing_weight, iou_thresh, score_thresh, w.cbox, w.cscore, w.cvalid);
    LAUNCH_CHECK();
    k_wbc_emit<<<1, 1024, 0, st>>>(n, w.cvalid, w.cbox, w.cscore, w.slabels, keepbits, out_boxes, out_scores, out_labels, out_count);
    LAUNCH_CHECK();
    return 0;
}
