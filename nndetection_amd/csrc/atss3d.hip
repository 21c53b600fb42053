// Fused ATSS matcher for gfx950 -- replaces ATSSMatcher.compute_matches (center_in_gt=False),
// nndet/core/boxes/matcher/atss.py:48-122, and never materialises the [G, M] distance / IoU /
// overlaps_inf matrices (3 x 4*G*M bytes + a [G,M,3] temporary in the reference).
//
// Per (GT g, level l) the candidate set is "the k anchors with the smallest (centre distance, index)".
// It is found with an exact MSB-first radix SELECT on the 64-bit key (dist_bits << 32 | anchor index):
// 8-bit digits, LDS histograms per workgroup flushed with global atomics, a tiny "pick" kernel between
// passes. Digits that are known to be zero (index bytes above log2(M)) are skipped. Distances are
// recomputed in every pass (24 B per anchor re-read from L2/HBM; no [G,M] storage). Then
//   stats  : sum / sum-of-squares of the candidate IoUs per GT (fp64 atomics) -> thr = mean + std(unbiased)
//   assign : per anchor arg-max over the GTs for which it is a candidate with IoU >= thr, else -1.
// Distance: sqrt((dx^2 + dy^2) + dz^2) with centres (hi + lo) / 2, fp32, correctly rounded sqrt
// (ops.py:262-287,314-327); IoU as box_iou_union_3d (ops.py:131-159). Compile with -ffp-contract=off.
#include "common.h"
#include "radix_select.h"

#define GT_TILE 16
#define MAXL 8
#define MAXB 64

struct AtssArgs {
    int64_t lvl_off[MAXL + 1];   // anchor offsets per level
    int32_t blk_off[MAXL + 1];   // workgroup offsets per level (256 anchors per workgroup)
    int32_t k[MAXL];             // candidates per level (clamped)
    int32_t L, G;
    int64_t M;
    int32_t B;                   // images in the batch (they share the anchors)
    int32_t img_off[MAXB + 1];   // GT offsets per image into the concatenated GT list
    int32_t center_in_gt;        // atss.py:101-107: a positive's centre must lie inside its GT box, further than min_dist from every face
    float min_dist;
};

// center_in_boxes (nndet/core/boxes/ops.py:290-311) of the anchor's centre (box_center, ops.py:314-327) and GT box g: the smallest of
// the six centre-to-face distances must exceed eps. Same fp32 expressions, no contraction (-ffp-contract=off).
__device__ __forceinline__ bool ctr_in_box(const float* g, const float* a, float eps) {
    const float cx = (a[2] + a[0]) / 2.f, cy = (a[3] + a[1]) / 2.f, cz = (a[5] + a[4]) / 2.f;
    const float m = fminf(fminf(fminf(cx - g[0], cy - g[1]), fminf(g[2] - cx, g[3] - cy)), fminf(cz - g[4], g[5] - cz));
    return m > eps;
}

__device__ __forceinline__ float ctr_dist(const float* g, const float* a) {
    float gx = (g[2] + g[0]) / 2.f, gy = (g[3] + g[1]) / 2.f, gz = (g[5] + g[4]) / 2.f;
    float ax = (a[2] + a[0]) / 2.f, ay = (a[3] + a[1]) / 2.f, az = (a[5] + a[4]) / 2.f;
    float dx = gx - ax, dy = gy - ay, dz = gz - az;
    return __fsqrt_rn((dx * dx + dy * dy) + dz * dz);
}

__device__ __forceinline__ float iou3(const float* a, const float* b) {
    float va = (a[2] - a[0]) * (a[3] - a[1]) * (a[5] - a[4]);
    float vb = (b[2] - b[0]) * (b[3] - b[1]) * (b[5] - b[4]);
    float x1 = fmaxf(a[0], b[0]), y1 = fmaxf(a[1], b[1]), x2 = fminf(a[2], b[2]), y2 = fminf(a[3], b[3]);
    float z1 = fmaxf(a[4], b[4]), z2 = fminf(a[5], b[5]);
    float inter = fmaxf(x2 - x1, 0.f) * fmaxf(y2 - y1, 0.f) * fmaxf(z2 - z1, 0.f);
    return inter / ((va + vb) - inter);
}

__device__ __forceinline__ int level_of_block(const AtssArgs& A, int b) {
    int l = 0;
    while (l + 1 < A.L && b >= A.blk_off[l + 1]) ++l;
    return l;
}

// state layout per (g, l): prefix (u64), krem (int). hist: [G][L][256] u32
// grid (total_blocks, ceil(G / GT_TILE)), block 256
__global__ __launch_bounds__(256) void k_atss_hist(AtssArgs A, const float* __restrict__ gt,
                                                   const float* __restrict__ anchors, const u64* __restrict__ prefix,
                                                   unsigned* __restrict__ hist, int shift) {
    __shared__ unsigned h[GT_TILE][256];
    __shared__ float g_s[GT_TILE][6];
    __shared__ u64 p_s[GT_TILE];
    const int l = level_of_block(A, blockIdx.x);
    const int g0 = blockIdx.y * GT_TILE;
    const int ng = min(GT_TILE, A.G - g0);
    for (int i = threadIdx.x; i < GT_TILE * 256; i += 256) (&h[0][0])[i] = 0;
    if ((int)threadIdx.x < ng * 6) (&g_s[0][0])[threadIdx.x] = gt[g0 * 6 + threadIdx.x];
    if ((int)threadIdx.x < ng) p_s[threadIdx.x] = prefix[(int64_t)(g0 + threadIdx.x) * A.L + l];
    __syncthreads();
    const int64_t a = A.lvl_off[l] + (int64_t)(blockIdx.x - A.blk_off[l]) * 256 + threadIdx.x;
    if (a < A.lvl_off[l + 1]) {
        float ab[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) ab[q] = anchors[a * 6 + q];
        for (int g = 0; g < ng; ++g) {
            const u64 key = ((u64)__float_as_uint(ctr_dist(g_s[g], ab)) << 32) | (u64)(uint32_t)a;
            const bool match = (shift >= 56) || ((key >> (shift + 8)) == (p_s[g] >> (shift + 8)));
            if (match) atomicAdd(&h[g][(unsigned)(key >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ng * 256; i += 256) {
        const unsigned v = (&h[0][0])[i];
        if (v) atomicAdd(&hist[((int64_t)(g0 + (i >> 8)) * A.L + l) * 256 + (i & 255)], v);
    }
}

// one WAVE per (g, l): choose the digit, update prefix / remaining rank, clear the histogram (radix_select.h; the serial
// 256-bin scan of v1 took 20 us per pass).
__global__ __launch_bounds__(256) void k_atss_pick(int GL, u64* __restrict__ prefix, int* __restrict__ krem,
                                                   unsigned* __restrict__ hist, int shift) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= GL) return;
    radix_pick_wave(prefix + i, krem + i, hist + (int64_t)i * 256, shift, threadIdx.x & 63);
}

__global__ void k_atss_init(int G, int L, AtssArgs A, u64* prefix, int* krem) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * L) return;
    prefix[i] = 0;
    krem[i] = A.k[i % L];
}

// candidate IoU statistics per GT. grid (total_blocks, ceil(G/GT_TILE))
__global__ __launch_bounds__(256) void k_atss_stats(AtssArgs A, const float* __restrict__ gt,
                                                    const float* __restrict__ anchors, const u64* __restrict__ kth,
                                                    double* __restrict__ sums /* [G][2] */) {
    __shared__ float g_s[GT_TILE][6];
    __shared__ u64 p_s[GT_TILE];
    __shared__ double red[GT_TILE][2];
    const int l = level_of_block(A, blockIdx.x);
    const int g0 = blockIdx.y * GT_TILE;
    const int ng = min(GT_TILE, A.G - g0);
    if ((int)threadIdx.x < ng * 6) (&g_s[0][0])[threadIdx.x] = gt[g0 * 6 + threadIdx.x];
    if ((int)threadIdx.x < ng) p_s[threadIdx.x] = kth[(int64_t)(g0 + threadIdx.x) * A.L + l];
    if ((int)threadIdx.x < GT_TILE * 2) (&red[0][0])[threadIdx.x] = 0.0;
    __syncthreads();
    const int64_t a = A.lvl_off[l] + (int64_t)(blockIdx.x - A.blk_off[l]) * 256 + threadIdx.x;
    if (a < A.lvl_off[l + 1]) {
        float ab[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) ab[q] = anchors[a * 6 + q];
        for (int g = 0; g < ng; ++g) {
            const u64 key = ((u64)__float_as_uint(ctr_dist(g_s[g], ab)) << 32) | (u64)(uint32_t)a;
            if (key <= p_s[g]) {
                const double v = (double)iou3(g_s[g], ab);
                atomicAdd(&red[g][0], v);
                atomicAdd(&red[g][1], v * v);
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < ng * 2) {
        const double v = (&red[0][0])[threadIdx.x];
        if (v != 0.0) atomicAdd(&sums[(int64_t)g0 * 2 + threadIdx.x], v);
    }
}

__global__ void k_atss_thr(int G, int ncand, const double* __restrict__ sums, float* __restrict__ thr) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const double n = (double)ncand;
    const double mean = sums[g * 2] / n;
    float t;
    if (ncand > 1) {
        double var = (sums[g * 2 + 1] - n * mean * mean) / (n - 1.0);
        if (var < 0.0) var = 0.0;
        t = (float)mean + (float)sqrt(var);   // atss.py:97-99: fp32 mean + fp32 std
    } else {
        t = __uint_as_float(0x7fc00000u);     // unbiased std of one sample is NaN -> nothing is positive
    }
    thr[g] = t;
}

// per anchor arg-max over the GTs of ONE image. grid (total_blocks, B), block 256
// labels (optional): the anchor's training label as BaseRetinaNet.assign_targets_to_anchors forms it (core/retina.py:262-287 of the
// reference): class of the matched GT + 1, 0 for an unmatched anchor -- written here instead of by a clamp / gather / compare / multiply
// chain over the [B, M] match tensor.
__global__ __launch_bounds__(256) void k_atss_assign(AtssArgs A, const float* __restrict__ gt,
                                                     const float* __restrict__ anchors, const u64* __restrict__ kth,
                                                     const float* __restrict__ thr, int64_t* __restrict__ matches,
                                                     const float* __restrict__ gt_cls, float* __restrict__ labels) {
    __shared__ float g_s[GT_TILE][6];
    __shared__ u64 p_s[GT_TILE];
    __shared__ float t_s[GT_TILE];
    const int l = level_of_block(A, blockIdx.x);
    const int img = blockIdx.y;
    const int gbeg = A.img_off[img], gend = A.img_off[img + 1];
    const int64_t a = A.lvl_off[l] + (int64_t)(blockIdx.x - A.blk_off[l]) * 256 + threadIdx.x;
    const bool valid = a < A.lvl_off[l + 1];
    float ab[6] = {0, 0, 0, 0, 0, 0};
    if (valid) {
#pragma unroll
        for (int q = 0; q < 6; ++q) ab[q] = anchors[a * 6 + q];
    }
    float best = -100.f;  // -INF of the reference (atss.py:16)
    int bi = -1;
    for (int g0 = gbeg; g0 < gend; g0 += GT_TILE) {
        const int ng = min(GT_TILE, gend - g0);
        __syncthreads();
        if ((int)threadIdx.x < ng * 6) (&g_s[0][0])[threadIdx.x] = gt[g0 * 6 + threadIdx.x];
        if ((int)threadIdx.x < ng) {
            p_s[threadIdx.x] = kth[(int64_t)(g0 + threadIdx.x) * A.L + l];
            t_s[threadIdx.x] = thr[g0 + threadIdx.x];
        }
        __syncthreads();
        if (valid) {
            for (int g = 0; g < ng; ++g) {
                const u64 key = ((u64)__float_as_uint(ctr_dist(g_s[g], ab)) << 32) | (u64)(uint32_t)a;
                if (key <= p_s[g]) {
                    const float v = iou3(g_s[g], ab);
                    if (v >= t_s[g] && v > best && (!A.center_in_gt || ctr_in_box(g_s[g], ab, A.min_dist))) {
                        best = v; bi = g0 + g - gbeg;                                // index local to the image
                    }
                }
            }
        }
    }
    if (valid) {
        matches[(int64_t)img * A.M + a] = (int64_t)bi;
        if (labels) labels[(int64_t)img * A.M + a] = bi >= 0 ? (gt_cls ? gt_cls[gbeg + bi] : 0.f) + 1.f : 0.f;
    }
}

__global__ void k_fill_i64(int64_t* p, int64_t n, int64_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct AtssWs { u64* prefix; int* krem; unsigned* hist; double* sums; float* thr; size_t total; };

static void atss_layout(int64_t G, int32_t L, char* base, AtssWs* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    size_t o_p = take((size_t)G * L * 8), o_k = take((size_t)G * L * 4), o_h = take((size_t)G * L * 256 * 4);
    size_t o_s = take((size_t)G * 2 * 8), o_t = take((size_t)G * 4);
    w->total = off;
    if (!base) return;                         // size query: no pointer arithmetic on a null base (UBSan, round 6)
    w->prefix = (u64*)(base + o_p); w->krem = (int*)(base + o_k); w->hist = (unsigned*)(base + o_h);
    w->sums = (double*)(base + o_s); w->thr = (float*)(base + o_t); w->total = off;
}

extern "C" size_t nndet_atss3d_workspace_bytes(int64_t G, int64_t M, int32_t L, int32_t k) {
    (void)M; (void)k;
    AtssWs w;
    atss_layout(G > 0 ? G : 1, L > 0 ? L : 1, nullptr, &w);
    return w.total;
}

static int atss_batched(const float* gt, const float* gt_cls, int64_t G, const int32_t* img_off_host, int32_t B,
                        const float* anchors, int64_t M, const int64_t* level_offsets_host, int32_t L,
                        int32_t k, int64_t* matches, float* labels, void* workspace, size_t workspace_bytes, void* stream,
                        int32_t center_in_gt = 0, float min_dist = 0.01f) {
    hipStream_t st = as_stream(stream);
    if (G < 0 || M < 0 || L <= 0 || L > MAXL || k <= 0 || !level_offsets_host || B <= 0 || B > MAXB || !img_off_host) return NNDET_EINVAL;
    if (M == 0) return 0;
    if (!matches || !anchors) return NNDET_EINVAL;
    if (M >= (1LL << 32) || img_off_host[0] != 0 || img_off_host[B] != G) return NNDET_EINVAL;
    if (G == 0) {  // Matcher.__call__ fast path (matcher/base.py:51-56) for every image
        k_fill_i64<<<(unsigned)ceil_div64(M * B, 256), 256, 0, st>>>(matches, M * B, -1);
        LAUNCH_CHECK();
        if (labels) HIP_TRY(hipMemsetAsync(labels, 0, (size_t)M * B * sizeof(float), st));
        return 0;
    }
    if (!gt || !workspace) return NNDET_EINVAL;
    if (level_offsets_host[0] != 0 || level_offsets_host[L] != M) return NNDET_EINVAL;
    AtssWs w;
    atss_layout(G, L, (char*)workspace, &w);
    if (w.total > workspace_bytes) return NNDET_EWORKSPACE;
    AtssArgs A;
    A.L = L; A.G = (int32_t)G; A.M = M; A.B = B; A.center_in_gt = center_in_gt; A.min_dist = min_dist;
    for (int b = 0; b <= B; ++b) {
        A.img_off[b] = img_off_host[b];
        if (b && img_off_host[b] < img_off_host[b - 1]) return NNDET_EINVAL;
    }
    int ncand = 0;
    A.blk_off[0] = 0;
    for (int l = 0; l < L; ++l) {
        A.lvl_off[l] = level_offsets_host[l];
        const int64_t sz = level_offsets_host[l + 1] - level_offsets_host[l];
        if (sz <= 0) return NNDET_EINVAL;
        A.k[l] = (int32_t)(sz < k ? sz : k);
        ncand += A.k[l];
        A.blk_off[l + 1] = A.blk_off[l] + (int32_t)ceil_div64(sz, 256);
    }
    A.lvl_off[L] = M;
    const int nblk = A.blk_off[L];
    const int GL = (int)G * L;
    const dim3 grid(nblk, ceil_div((int)G, GT_TILE));
    k_atss_init<<<ceil_div(GL, 256), 256, 0, st>>>((int)G, L, A, w.prefix, w.krem);
    LAUNCH_CHECK();
    HIP_TRY(hipMemsetAsync(w.hist, 0, (size_t)GL * 256 * 4, st));
    HIP_TRY(hipMemsetAsync(w.sums, 0, (size_t)G * 16, st));
    // digits of the low (index) word above the highest set bit of (M - 1) are zero for every key: skip them
    int idx_bits = 1;
    while (((int64_t)1 << idx_bits) < M) ++idx_bits;
    for (int shift = 56; shift >= 0; shift -= 8) {
        if (shift < 32 && shift >= idx_bits) continue;
        k_atss_hist<<<grid, 256, 0, st>>>(A, gt, anchors, w.prefix, w.hist, shift);
        LAUNCH_CHECK();
        k_atss_pick<<<ceil_div(GL, 4), 256, 0, st>>>(GL, w.prefix, w.krem, w.hist, shift);
        LAUNCH_CHECK();
    }
    k_atss_stats<<<grid, 256, 0, st>>>(A, gt, anchors, w.prefix, w.sums);
    LAUNCH_CHECK();
    k_atss_thr<<<ceil_div((int)G, 64), 64, 0, st>>>((int)G, ncand, w.sums, w.thr);
    LAUNCH_CHECK();
    k_atss_assign<<<dim3(nblk, B), 256, 0, st>>>(A, gt, anchors, w.prefix, w.thr, matches, gt_cls, labels);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_atss3d_match_batched_f32(const float* gt, int64_t G, const int32_t* img_off_host, int32_t B,
                                              const float* anchors, int64_t M, const int64_t* level_offsets_host, int32_t L,
                                              int32_t k, int64_t* matches, void* workspace, size_t workspace_bytes, void* stream) {
    return atss_batched(gt, nullptr, G, img_off_host, B, anchors, M, level_offsets_host, L, k, matches, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int nndet_atss3d_assign_batched_f32(const float* gt, const float* gt_classes, int64_t G, const int32_t* img_off_host, int32_t B,
                                               const float* anchors, int64_t M, const int64_t* level_offsets_host, int32_t L,
                                               int32_t k, int32_t center_in_gt, float min_dist, int64_t* matches, float* labels_out,
                                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!labels_out) return NNDET_EINVAL;
    return atss_batched(gt, gt_classes, G, img_off_host, B, anchors, M, level_offsets_host, L, k, matches, labels_out, workspace, workspace_bytes, stream,
                        center_in_gt ? 1 : 0, min_dist);
}

extern "C" int nndet_atss3d_match_f32(const float* gt, int64_t G, const float* anchors, int64_t M,
                                      const int64_t* level_offsets_host, int32_t L, int32_t k, int64_t* matches,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (G < 0 || G > 0x7fffffff) return NNDET_EINVAL;
    const int32_t off[2] = {0, (int32_t)G};
    return nndet_atss3d_match_batched_f32(gt, G, off, 1, anchors, M, level_offsets_host, L, k, matches, workspace, workspace_bytes, stream);
}


// ------------------------------------------------------------------------------------------------ IoU-threshold matcher
// IoUMatcher.compute_matches (nndet/core/boxes/matcher/iou.py:43-107; the matcher of the reference's skeleton module -- RetinaUNetV001
// uses ATSS): per anchor the GT with the highest IoU (ties: lowest GT index), BELOW_LOW_THRESHOLD (-1) if that IoU < low,
// BETWEEN_THRESHOLDS (-2) if low <= IoU < high; with allow_low_quality_matches every GT additionally claims the anchor it overlaps most
// (ties: lowest anchor index; two GTs claiming one anchor: the higher GT index wins, as the reference's in-order index assignment).
// The [G, M] IoU matrix is never written: pass 1 keeps, per GT, the key (IoU bits << 32 | ~anchor) of its best anchor by atomicMax.
__global__ __launch_bounds__(256) void k_ioum_assign(const float* __restrict__ gt, int G, const float* __restrict__ anchors, int64_t M,
                                                     float low, float high, int64_t* __restrict__ matches, u64* __restrict__ best_key) {
    __shared__ float g_s[GT_TILE][6];
    __shared__ u64 k_s[GT_TILE];
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = a < M;
    float ab[6] = {0, 0, 0, 0, 0, 0};
    if (valid) {
#pragma unroll
        for (int q = 0; q < 6; ++q) ab[q] = anchors[a * 6 + q];
    }
    float best = -1.f;
    int bi = 0;
    for (int g0 = 0; g0 < G; g0 += GT_TILE) {
        const int ng = min(GT_TILE, G - g0);
        __syncthreads();
        if ((int)threadIdx.x < ng * 6) (&g_s[0][0])[threadIdx.x] = gt[g0 * 6 + threadIdx.x];
        if ((int)threadIdx.x < ng) k_s[threadIdx.x] = 0ull;
        __syncthreads();
        if (valid) {
            for (int g = 0; g < ng; ++g) {
                const float v = iou3(g_s[g], ab);
                if (v > best) { best = v; bi = g0 + g; }                       // first maximum: lowest GT index (NaN never wins)
                if (best_key && v >= 0.f) atomicMax(&k_s[g], ((u64)__float_as_uint(v) << 32) | (u64)(~(uint32_t)a));
            }
        }
        __syncthreads();
        if (best_key && (int)threadIdx.x < ng && k_s[threadIdx.x]) atomicMax(&best_key[g0 + threadIdx.x], k_s[threadIdx.x]);
    }
    if (valid) {
        int64_t m = bi;
        if (best < low) m = -1;                                                 // (an all-NaN column keeps best = -1 -> below)
        else if (best < high) m = -2;
        matches[a] = m;
    }
}

__global__ void k_ioum_lowq(int G, const u64* __restrict__ best_key, int64_t* __restrict__ matches) {
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int g = 0; g < G; ++g) {
            const u64 k = best_key[g];
            if (k) matches[(int64_t)(~(uint32_t)(k & 0xffffffffull))] = (int64_t)g;
        }
}

extern "C" int nndet_iou_match3d_f32(const float* gt, int64_t G, const float* anchors, int64_t M, float low_threshold, float high_threshold,
                                     int32_t allow_low_quality_matches, int64_t* matches, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    hipStream_t st = as_stream(stream);
    if (G < 0 || G > 0x7fffffff || M < 0 || M >= (1LL << 32) || low_threshold > high_threshold) return NNDET_EINVAL;
    if (M == 0) return 0;
    if (!matches || !anchors) return NNDET_EINVAL;
    if (G == 0) {                                                              // Matcher.__call__ fast path (matcher/base.py:51-56)
        k_fill_i64<<<(unsigned)ceil_div64(M, 256), 256, 0, st>>>(matches, M, -1);
        LAUNCH_CHECK();
        return 0;
    }
    if (!gt) return NNDET_EINVAL;
    u64* keys = nullptr;
    if (allow_low_quality_matches) {
        if (!workspace || workspace_bytes < (size_t)G * 8) return NNDET_EWORKSPACE;
        keys = reinterpret_cast<u64*>(workspace);
        HIP_TRY(hipMemsetAsync(keys, 0, (size_t)G * 8, st));
    }
    k_ioum_assign<<<(unsigned)ceil_div64(M, 256), 256, 0, st>>>(gt, (int)G, anchors, M, low_threshold, high_threshold, matches, keys);
    LAUNCH_CHECK();
    if (keys) {
        k_ioum_lowq<<<1, 64, 0, st>>>((int)G, keys, matches);
        LAUNCH_CHECK();
    }
    return 0;
}
