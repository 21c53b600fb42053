// Box-op kernels for gfx950: pairwise IoU / GIoU, diagonal GIoU (+grad), anchor grid, decode+clip, sigmoid-max.
// Reference: nndet/core/boxes/ops.py:75-185, anchors.py:337-377, coder.py:90-155, clip.py:83-101.
// All fp32; the operation order follows the reference expression by expression (bit-exact for + - * / min max
// with -ffp-contract=off and IEEE division). These kernels are HBM-bound: IoU writes 4 B per pair
// (float4 stores, 1 KiB per wave-instruction), anchors write 24 B per anchor.
#include "common.h"

struct Box { float x1, y1, x2, y2, z1, z2; };

__device__ __forceinline__ Box ldbox(const float* p) { return Box{p[0], p[1], p[2], p[3], p[4], p[5]}; }
__device__ __forceinline__ float vol3(const Box& b) { return (b.x2 - b.x1) * (b.y2 - b.y1) * (b.z2 - b.z1); }

// inter / union exactly as box_iou_union_3d (ops.py:131-159); returns union through *u
__device__ __forceinline__ float iou_union(const Box& a, float va, const Box& b, float vb, float eps, float* u) {
    float x1 = fmaxf(a.x1, b.x1), y1 = fmaxf(a.y1, b.y1), x2 = fminf(a.x2, b.x2), y2 = fminf(a.y2, b.y2);
    float z1 = fmaxf(a.z1, b.z1), z2 = fminf(a.z2, b.z2);
    float inter = (fmaxf(x2 - x1, 0.f) * fmaxf(y2 - y1, 0.f) * fmaxf(z2 - z1, 0.f)) + eps;
    float un = (va + vb) - inter;
    *u = un;
    return inter / un;
}

__device__ __forceinline__ float giou_val(const Box& a, float va, const Box& b, float vb, float eps) {
    float un;
    float iou = iou_union(a, va, b, vb, 0.f, &un);  // eps is NOT forwarded to the inner IoU (ops.py:175)
    float x1 = fminf(a.x1, b.x1), y1 = fminf(a.y1, b.y1), x2 = fmaxf(a.x2, b.x2), y2 = fmaxf(a.y2, b.y2);
    float z1 = fminf(a.z1, b.z1), z2 = fmaxf(a.z2, b.z2);
    float vol = (fmaxf(x2 - x1, 0.f) * fmaxf(y2 - y1, 0.f) * fmaxf(z2 - z1, 0.f)) + eps;
    return iou - (vol - un) / vol;
}

// grid (ceil(m/1024), ceil(n/ROWS)); block 256; each thread owns 4 consecutive columns and ROWS rows.
template <bool GIOU, int ROWS>
__global__ __launch_bounds__(256) void k_pairwise(const float* __restrict__ a, int64_t n, const float* __restrict__ b,
                                                  int64_t m, float eps, float* __restrict__ out) {
    __shared__ float arow[ROWS * 6];
    const int64_t r0 = (int64_t)blockIdx.y * ROWS;
    const int nr = (int)min((int64_t)ROWS, n - r0);
    if ((int)threadIdx.x < nr * 6) arow[threadIdx.x] = a[r0 * 6 + threadIdx.x];
    __syncthreads();
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c0 >= m) return;
    Box bb[4];
    float vb[4];
    const int nc = (int)min((int64_t)4, m - c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bb[j] = ldbox(b + (c0 + (j < nc ? j : 0)) * 6);
        vb[j] = vol3(bb[j]);
    }
    const bool vec = (nc == 4) && ((m & 3) == 0);
    for (int r = 0; r < nr; ++r) {
        Box ab = ldbox(arow + r * 6);
        float va = vol3(ab);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float un;
            v[j] = GIOU ? giou_val(ab, va, bb[j], vb[j], eps) : iou_union(ab, va, bb[j], vb[j], eps, &un);
        }
        float* o = out + (r0 + r) * m + c0;
        if (vec) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int j = 0; j < nc; ++j) o[j] = v[j];
        }
    }
}

template <bool GIOU>
static int pairwise(const float* a, int64_t n, const float* b, int64_t m, float eps, float* out, void* stream) {
    if (n < 0 || m < 0) return NNDET_EINVAL;
    if (n == 0 || m == 0) return 0;
    if (!a || !b || !out) return NNDET_EINVAL;
    constexpr int ROWS = 16;
    dim3 grid((unsigned)ceil_div64(m, 1024), (unsigned)ceil_div64(n, ROWS));
    k_pairwise<GIOU, ROWS><<<grid, 256, 0, as_stream(stream)>>>(a, n, b, m, eps, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_iou3d_pairwise_f32(const float* a, int64_t n, const float* b, int64_t m, float eps, float* out,
                                        void* stream) { return pairwise<false>(a, n, b, m, eps, out, stream); }
extern "C" int nndet_giou3d_pairwise_f32(const float* a, int64_t n, const float* b, int64_t m, float eps, float* out,
                                         void* stream) { return pairwise<true>(a, n, b, m, eps, out, stream); }

// ------------------------------------------------------------------ row maximum of the IoU matrix (anchor search objective)
// out[i] = max_j IoU(a[i], b[j]) without the [n, m] matrix: the planner's anchor optimisation evaluates
// box_iou(gt, anchors).max(dim=1)[0].mean() 15 000 times (nndet/planning/architecture/boxes/base.py:424-484).
// One workgroup per 256 rows; the columns stream through LDS in tiles of 256. NaN handling follows torch.max (NaN propagates).
__global__ __launch_bounds__(256) void k_iou_rowmax(const float* __restrict__ a, int64_t n, const float* __restrict__ b, int64_t m,
                                                    float eps, float* __restrict__ out) {
    __shared__ float tile[256 * 6];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = i < n;
    Box ab = ldbox(a + (ok ? i : 0) * 6);
    const float va = vol3(ab);
    float best = -INFINITY;
    bool nan = false;
    for (int64_t c0 = 0; c0 < m; c0 += 256) {
        const int nc = (int)min((int64_t)256, m - c0);
        __syncthreads();
        for (int k = threadIdx.x; k < nc * 6; k += 256) tile[k] = b[c0 * 6 + k];
        __syncthreads();
        for (int j = 0; j < nc; ++j) {
            Box bb = ldbox(tile + j * 6);
            float un;
            const float v = iou_union(ab, va, bb, vol3(bb), eps, &un);
            nan = nan || (v != v);
            best = fmaxf(best, v);
        }
    }
    if (ok) out[i] = nan ? __uint_as_float(0x7fc00000u) : best;
}

extern "C" int nndet_iou3d_rowmax_f32(const float* a, int64_t n, const float* b, int64_t m, float eps, float* out, void* stream) {
    if (n < 0 || m <= 0) return NNDET_EINVAL;
    if (n == 0) return 0;
    if (!a || !b || !out) return NNDET_EINVAL;
    k_iou_rowmax<<<(unsigned)ceil_div64(n, 256), 256, 0, as_stream(stream)>>>(a, n, b, m, eps, out);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ diagonal GIoU (loss) + gradient w.r.t. a
__global__ void k_giou_diag_fwd(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float eps,
                                float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Box ab = ldbox(a + i * 6), bb = ldbox(b + i * 6);
    out[i] = giou_val(ab, vol3(ab), bb, vol3(bb), eps);
}

// Analytic gradient (same sub-gradient conventions as autograd of the reference expression:
// max/min route the gradient to the selected operand -- on ties torch.max/min(a, b) split 0.5/0.5 --
// and clamp(min=0) passes gradient where the argument is > 0... torch passes it for >= 0? No: clamp's
// backward mask is (x >= min), so exactly-0 extents still receive gradient.
__device__ __forceinline__ void giou_grad(const float* A, const float* B, float g, float eps, float* gout) {
    // axis k: lo index, hi index
    const int LO[3] = {0, 1, 4}, HI[3] = {2, 3, 5};
    float ext[3], iw[3], hw[3];           // box extent, clamped intersection extent, clamped hull extent
    float ilo_w[3], ihi_w[3], hlo_w[3], hhi_w[3];  // d(unclamped extents)/d(a.lo / a.hi) weights (0, 0.5 or 1)
    float iraw[3], hraw[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float alo = A[LO[k]], ahi = A[HI[k]], blo = B[LO[k]], bhi = B[HI[k]];
        ext[k] = ahi - alo;
        // intersection: lo = max(alo, blo), hi = min(ahi, bhi)
        ilo_w[k] = alo > blo ? 1.f : (alo == blo ? 0.5f : 0.f);
        ihi_w[k] = ahi < bhi ? 1.f : (ahi == bhi ? 0.5f : 0.f);
        iraw[k] = fminf(ahi, bhi) - fmaxf(alo, blo);
        iw[k] = fmaxf(iraw[k], 0.f);
        // hull: lo = min(alo, blo), hi = max(ahi, bhi)
        hlo_w[k] = alo < blo ? 1.f : (alo == blo ? 0.5f : 0.f);
        hhi_w[k] = ahi > bhi ? 1.f : (ahi == bhi ? 0.5f : 0.f);
        hraw[k] = fmaxf(ahi, bhi) - fminf(alo, blo);
        hw[k] = fmaxf(hraw[k], 0.f);
    }
    float va = ext[0] * ext[1] * ext[2];
    float vb = (B[2] - B[0]) * (B[3] - B[1]) * (B[5] - B[4]);
    float inter = iw[0] * iw[1] * iw[2];
    float un = va + vb - inter;
    float vol = hw[0] * hw[1] * hw[2] + eps;
    // giou = inter/un - (vol - un)/vol = inter/un - 1 + un/vol
    float d_inter = g * (1.f / un);
    float d_un = g * (-inter / (un * un) + 1.f / vol);
    float d_vol = g * (-un / (vol * vol));
    // un = va + vb - inter
    float d_va = d_un;
    d_inter -= d_un;
#pragma unroll
    for (int k = 0; k < 6; ++k) gout[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        // d va / d ext_k
        float d_ext = d_va * ext[k1] * ext[k2];
        gout[HI[k]] += d_ext;
        gout[LO[k]] -= d_ext;
        // d inter / d iw_k (clamp passes gradient when raw >= 0)
        float d_iw = (iraw[k] >= 0.f) ? d_inter * iw[k1] * iw[k2] : 0.f;
        gout[HI[k]] += d_iw * ihi_w[k];
        gout[LO[k]] -= d_iw * ilo_w[k];
        float d_hw = (hraw[k] >= 0.f) ? d_vol * hw[k1] * hw[k2] : 0.f;
        gout[HI[k]] += d_hw * hhi_w[k];
        gout[LO[k]] -= d_hw * hlo_w[k];
    }
}

__global__ void k_giou_diag_bwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ go,
                                int64_t n, float eps, float* __restrict__ ga) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gout[6];
    giou_grad(a + i * 6, b + i * 6, go[i], eps, gout);
#pragma unroll
    for (int k = 0; k < 6; ++k) ga[i * 6 + k] = gout[k];
}

// ------------------------------------------------------------------ gradient of the full [n, m] GIoU matrix (ops.py:106-128,162-185)
// generalized_box_iou is differentiable w.r.t. both box sets in the reference (plain autograd through min / max / clamp); GIoULoss
// (losses/regression.py:158-161) back-propagates through it. grad_a[i] = sum_j d giou(a_i, b_j) / d a_i * g[i, j]; the expression is
// symmetric in its arguments (min / max / + commute bit for bit, ties split 0.5 / 0.5), so grad_b[j] = sum_i giou_grad(b_j, a_i) * g[i, j].
// One workgroup per output row; the 256 threads stride over the other box set, partial sums in float64, combined in a fixed order
// (deterministic; one rounding per output).
template <bool TR>
__global__ __launch_bounds__(256) void k_giou_pairwise_bwd(const float* __restrict__ a, int64_t n, const float* __restrict__ b, int64_t m,
                                                           const float* __restrict__ go, float eps, float* __restrict__ gout) {
    // TR = false: row = a_i, partners b_j, cotangent go[i * m + j];  TR = true: row = b_j, partners a_i, cotangent go[i * m + j]
    __shared__ double red[256 * 6];
    const int64_t row = blockIdx.x;
    const float* self = (TR ? b : a) + row * 6;
    const float* other = TR ? a : b;
    const int64_t cnt = TR ? n : m;
    float S[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) S[k] = self[k];
    double acc[6] = {0., 0., 0., 0., 0., 0.};
    for (int64_t j = threadIdx.x; j < cnt; j += 256) {
        float O[6], G[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) O[k] = other[j * 6 + k];
        const float g = TR ? go[j * m + row] : go[row * m + j];
        giou_grad(S, O, g, eps, G);
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += (double)G[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) red[k * 256 + threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[k * 256 + threadIdx.x] += red[k * 256 + threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 6) gout[row * 6 + threadIdx.x] = (float)red[threadIdx.x * 256];
}

extern "C" int nndet_giou3d_pairwise_bwd_f32(const float* a, int64_t n, const float* b, int64_t m, const float* grad_out, float eps,
                                             float* grad_a, float* grad_b, void* stream) {
    if (n < 0 || m < 0) return NNDET_EINVAL;
    if (n == 0 || m == 0) return 0;
    if (!a || !b || !grad_out || (!grad_a && !grad_b)) return NNDET_EINVAL;
    if (n > 0x7fffffff || m > 0x7fffffff) return NNDET_EINVAL;
    if (grad_a) { k_giou_pairwise_bwd<false><<<(unsigned)n, 256, 0, as_stream(stream)>>>(a, n, b, m, grad_out, eps, grad_a); LAUNCH_CHECK(); }
    if (grad_b) { k_giou_pairwise_bwd<true><<<(unsigned)m, 256, 0, as_stream(stream)>>>(a, n, b, m, grad_out, eps, grad_b); LAUNCH_CHECK(); }
    return 0;
}

extern "C" int nndet_giou3d_diag_fwd_f32(const float* a, const float* b, int64_t n, float eps, float* out, void* stream) {
    if (n < 0) return NNDET_EINVAL;
    if (n == 0) return 0;
    k_giou_diag_fwd<<<(unsigned)ceil_div64(n, 256), 256, 0, as_stream(stream)>>>(a, b, n, eps, out);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int nndet_giou3d_diag_bwd_f32(const float* a, const float* b, const float* grad_out, int64_t n, float eps,
                                         float* grad_a, void* stream) {
    if (n < 0) return NNDET_EINVAL;
    if (n == 0) return 0;
    k_giou_diag_bwd<<<(unsigned)ceil_div64(n, 256), 256, 0, as_stream(stream)>>>(a, b, grad_out, n, eps, grad_a);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ anchors
// one thread per (cell, anchor): writes 6 floats = 24 B; consecutive threads write consecutive 24-B records.
__global__ void k_anchor_grid(const float* __restrict__ cell, int A, int sx, int sy, int sz, float stx, float sty,
                              float stz, float* __restrict__ out, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int a = (int)(i % A);
    int64_t c = i / A;
    int z = (int)(c % sz);
    int64_t t = c / sz;
    int y = (int)(t % sy);
    int x = (int)(t / sy);
    // shifts = arange(size) * stride (anchors.py:361-363), anchors = shifts + base (anchors.py:371)
    float fx = (float)x * stx, fy = (float)y * sty, fz = (float)z * stz;
    const float* b = cell + a * 6;
    float* o = out + i * 6;
    o[0] = fx + b[0]; o[1] = fy + b[1]; o[2] = fx + b[2]; o[3] = fy + b[3]; o[4] = fz + b[4]; o[5] = fz + b[5];
}

extern "C" int nndet_anchors3d_grid_f32(const float* cell, int32_t A, int32_t sx, int32_t sy, int32_t sz,
                                        int32_t stride_x, int32_t stride_y, int32_t stride_z, float* out, void* stream) {
    if (A <= 0 || sx <= 0 || sy <= 0 || sz <= 0 || !cell || !out) return NNDET_EINVAL;
    int64_t total = (int64_t)sx * sy * sz * A;
    k_anchor_grid<<<(unsigned)ceil_div64(total, 256), 256, 0, as_stream(stream)>>>(
        cell, A, sx, sy, sz, (float)stride_x, (float)stride_y, (float)stride_z, out, total);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ decode + clip
__global__ void k_decode_clip(const float* __restrict__ rel, const float* __restrict__ anchors, int64_t n,
                              int64_t n_anchor, float clip_exp, float ix, float iy, float iz, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rel + i * 6;
    const float* b = anchors + (i % n_anchor) * 6;
    // coder.py:107-151 with weights == 1 (x / 1 == x)
    float w = b[2] - b[0], h = b[3] - b[1], d = b[5] - b[4];
    float cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * h, cz = b[4] + 0.5f * d;
    float dw = fminf(r[2], clip_exp), dh = fminf(r[3], clip_exp), dd = fminf(r[5], clip_exp);
    float pcx = r[0] * w + cx, pcy = r[1] * h + cy, pcz = r[4] * d + cz;
    float pw = expf(dw) * w, ph = expf(dh) * h, pd = expf(dd) * d;
    float o0 = pcx - 0.5f * pw, o1 = pcy - 0.5f * ph, o2 = pcx + 0.5f * pw, o3 = pcy + 0.5f * ph;
    float o4 = pcz - 0.5f * pd, o5 = pcz + 0.5f * pd;
    if (ix > 0.f) {  // clip.py:95-100
        o0 = fminf(fmaxf(o0, 0.f), ix); o2 = fminf(fmaxf(o2, 0.f), ix);
        o1 = fminf(fmaxf(o1, 0.f), iy); o3 = fminf(fmaxf(o3, 0.f), iy);
        o4 = fminf(fmaxf(o4, 0.f), iz); o5 = fminf(fmaxf(o5, 0.f), iz);
    }
    float* o = out + i * 6;
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3; o[4] = o4; o[5] = o5;
}

extern "C" int nndet_decode_clip3d_f32(const float* rel, const float* anchors, int64_t n, int64_t n_anchor,
                                       float clip_exp, float img_x, float img_y, float img_z, float* out, void* stream) {
    if (n < 0 || n_anchor <= 0) return NNDET_EINVAL;
    if (n == 0) return 0;
    k_decode_clip<<<(unsigned)ceil_div64(n, 256), 256, 0, as_stream(stream)>>>(rel, anchors, n, n_anchor, clip_exp,
                                                                              img_x, img_y, img_z, out);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ sigmoid + max over classes
__global__ void k_sigmoid_max(const float* __restrict__ logits, int64_t n, int C, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, logits[i * C + c]);
    out[i] = 1.f / (1.f + expf(-m));  // sigmoid is monotone: max(sigmoid(x)) == sigmoid(max(x))
}

extern "C" int nndet_sigmoid_max_f32(const float* logits, int64_t n, int32_t C, float* out, void* stream) {
    if (n < 0 || C <= 0) return NNDET_EINVAL;
    if (n == 0) return 0;
    k_sigmoid_max<<<(unsigned)ceil_div64(n, 256), 256, 0, as_stream(stream)>>>(logits, n, C, out);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ fused detection loss on the sampled anchors
// DetectionHeadHNM.compute_loss (nndet/arch/heads/comb.py:351-405) after the sampler: decode of the <= pos_cap positives
// (coder.py:90-155, unit weights) -> -GIoU of (decoded, matched GT) (losses/regression.py:118-162) and BCE-with-logits of the
// positives + negatives against the one-hot labels without the background column (losses/classification.py:137-181), forward AND
// the per-row gradients in ONE single-workgroup launch. In torch this is ~250 element-wise launches of 1-170 elements (index,
// decode, where, one_hot, bce, sum, ... and their autograd mirror images): 1.5-3 ms of pure launch latency on the critical path of
// every training step between the forward and the backward pass (profiles/round2_v4_timeline_one_step.txt).
// Index lists are the device sampler's: fixed capacity, padded with -1; counts = {num_pos, num_neg, ...} (int64, device).
struct DetLossArgs {
    const float* logits; const float* deltas; const int64_t* pos; const int64_t* neg; const int64_t* counts;
    const float* labels; const float* gt; const float* anchors;
    int64_t m_anchors;
    int32_t P, Q, C, reg_mean, cls_mean;
    int32_t compact;        // deltas are [P][6] rows in the order of pos (nndet_detloss_compact_f32) instead of [B * M][6]
    // matched GT boxes either as the dense [B * M][6] tensor `gt` (the reference's matched_gt_boxes, retina.py:262-287) or INDIRECT:
    // `matches` [B * M] (index of the matched GT local to the image, the ATSS kernel's output) + `gt` = the concatenated GT boxes
    // [G][6] + gt_base[image] = first GT row of the image: the [B, M, 6] gather (114 MB per step at 160x160x96, batch 4) is never made
    const int64_t* matches;
    int32_t gt_base[65];
    float eps, clip, reg_w, cls_w;
    float* losses; float* g_deltas; float* g_logits;
};

__global__ __launch_bounds__(256) void k_detloss(const DetLossArgs A) {
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    const float n_pos = (float)A.counts[0], n_neg = (float)A.counts[1];
    const float n_pos_f = fmaxf(n_pos, 1.f);
    // loss = reg_w * -1 * (sum giou [/ n_pos_f if mean]) / n_pos_f  +  cls_w * (sum bce [/ (max(n_pos + n_neg, 1) * C) if mean])
    const float reg_coef = -A.reg_w / n_pos_f / (A.reg_mean ? n_pos_f : 1.f);
    const float cls_coef = A.cls_w / (A.cls_mean ? fmaxf(n_pos + n_neg, 1.f) * (float)A.C : 1.f);
    float reg_part = 0.f, cls_part = 0.f;
    for (int r = tid; r < A.P; r += 256) {
        const int64_t idx = A.pos[r];
        float gd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (idx >= 0) {
            const float* d = A.deltas + (A.compact ? (int64_t)r : idx) * 6;
            const float* b = A.anchors + (idx % A.m_anchors) * 6;
            // coder.py:107-151 with weights == 1 (the expression order of k_decode_clip)
            const float w = b[2] - b[0], h = b[3] - b[1], dd_ = b[5] - b[4];
            const float cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * h, cz = b[4] + 0.5f * dd_;
            const float dw = fminf(d[2], A.clip), dh = fminf(d[3], A.clip), dz = fminf(d[5], A.clip);
            const float pcx = d[0] * w + cx, pcy = d[1] * h + cy, pcz = d[4] * dd_ + cz;
            const float pw = expf(dw) * w, ph = expf(dh) * h, pd = expf(dz) * dd_;
            const float p[6] = {pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph, pcz - 0.5f * pd, pcz + 0.5f * pd};
            const float* t = A.matches ? A.gt + (int64_t)(A.gt_base[idx / A.m_anchors] + (A.matches[idx] > 0 ? A.matches[idx] : 0)) * 6
                                       : A.gt + idx * 6;
            const Box pb = ldbox(p), tb = ldbox(t);
            reg_part += giou_val(pb, vol3(pb), tb, vol3(tb), A.eps);
            float G[6];
            giou_grad(p, t, reg_coef, A.eps, G);                       // d loss / d decoded box
            // decode backward: clamp(max) passes the gradient where x <= max
            gd[0] = (G[0] + G[2]) * w; gd[1] = (G[1] + G[3]) * h; gd[4] = (G[4] + G[5]) * dd_;
            gd[2] = d[2] <= A.clip ? (G[2] - G[0]) * 0.5f * pw : 0.f;
            gd[3] = d[3] <= A.clip ? (G[3] - G[1]) * 0.5f * ph : 0.f;
            gd[5] = d[5] <= A.clip ? (G[5] - G[4]) * 0.5f * pd : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) A.g_deltas[r * 6 + k] = gd[k];
    }
    for (int r = tid; r < A.P + A.Q; r += 256) {
        const int64_t idx = r < A.P ? A.pos[r] : A.neg[r - A.P];
        const float lab = idx >= 0 ? A.labels[idx] : 0.f;
        for (int c = 0; c < A.C; ++c) {
            float g = 0.f;
            if (idx >= 0) {
                const float x = A.logits[idx * A.C + c];
                const float y = ((int)lab == c + 1) ? 1.f : 0.f;
                // binary_cross_entropy_with_logits: (1 - y) * x + max(-x, 0) + log(exp(-max(-x, 0)) + exp(-x - max(-x, 0)))
                const float mv = fmaxf(-x, 0.f);
                cls_part += (1.f - y) * x + mv + logf(expf(-mv) + expf(-x - mv));
                g = (1.f / (1.f + expf(-x)) - y) * cls_coef;
            }
            A.g_logits[r * A.C + c] = g;
        }
    }
    reg_part = wave_sum_f32(reg_part); cls_part = wave_sum_f32(cls_part);
    if ((tid & 63) == 0) { red[0][tid >> 6] = reg_part; red[1][tid >> 6] = cls_part; }
    __syncthreads();
    if (tid == 0) {
        const float gs = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), bs = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        A.losses[0] = gs * reg_coef;
        A.losses[1] = bs * cls_coef;
    }
}

// d_deltas / d_logits (dense, zeroed by the caller): the sampled rows scaled by the upstream gradients of the two losses
__global__ __launch_bounds__(256) void k_detloss_scatter(const int64_t* __restrict__ pos, int P, const int64_t* __restrict__ neg, int Q, int C,
                                                         const float* __restrict__ g_deltas, const float* __restrict__ g_logits,
                                                         const float* __restrict__ up, float* __restrict__ d_deltas,
                                                         float* __restrict__ d_logits) {
    const float u_reg = up[0], u_cls = up[1];
    for (int r = threadIdx.x; r < P + Q; r += 256) {
        const int64_t idx = r < P ? pos[r] : neg[r - P];
        if (idx < 0) continue;
        if (r < P && d_deltas) {
#pragma unroll
            for (int k = 0; k < 6; ++k) d_deltas[idx * 6 + k] = u_reg * g_deltas[r * 6 + k];
        }
        for (int c = 0; c < C; ++c) d_logits[idx * C + c] = u_cls * g_logits[r * C + c];
    }
}

extern "C" int nndet_detloss_f32(const float* logits, const float* deltas, const int64_t* pos, int32_t pos_cap, const int64_t* neg,
                                 int32_t neg_cap, const int64_t* counts, const float* labels, const float* matched_gt,
                                 const float* anchors, int64_t m_anchors, int32_t C, float eps, float clip, float reg_weight,
                                 int32_t reg_mean, float cls_weight, int32_t cls_mean, float* losses_out, float* g_deltas_out,
                                 float* g_logits_out, void* stream) {
    if (!logits || !deltas || !pos || !neg || !counts || !labels || !matched_gt || !anchors || !losses_out || !g_deltas_out || !g_logits_out)
        return NNDET_EINVAL;
    if (pos_cap < 1 || neg_cap < 0 || C < 1 || m_anchors < 1) return NNDET_EINVAL;
    DetLossArgs a;
    a.logits = logits; a.deltas = deltas; a.pos = pos; a.neg = neg; a.counts = counts; a.labels = labels; a.gt = matched_gt;
    a.anchors = anchors; a.m_anchors = m_anchors; a.P = pos_cap; a.Q = neg_cap; a.C = C; a.reg_mean = reg_mean; a.cls_mean = cls_mean;
    a.eps = eps; a.clip = clip; a.reg_w = reg_weight; a.cls_w = cls_weight; a.compact = 0; a.matches = nullptr;
    a.losses = losses_out; a.g_deltas = g_deltas_out; a.g_logits = g_logits_out;
    k_detloss<<<1, 256, 0, as_stream(stream)>>>(a);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_detloss_compact_f32(const float* logits, const float* deltas_compact, const int64_t* pos, int32_t pos_cap,
                                         const int64_t* neg, int32_t neg_cap, const int64_t* counts, const float* labels,
                                         const float* matched_gt, const float* anchors, int64_t m_anchors, int32_t C, float eps,
                                         float clip, float reg_weight, int32_t reg_mean, float cls_weight, int32_t cls_mean,
                                         float* losses_out, float* g_deltas_out, float* g_logits_out, void* stream) {
    if (!logits || !deltas_compact || !pos || !neg || !counts || !labels || !matched_gt || !anchors || !losses_out || !g_deltas_out || !g_logits_out)
        return NNDET_EINVAL;
    if (pos_cap < 1 || neg_cap < 0 || C < 1 || m_anchors < 1) return NNDET_EINVAL;
    DetLossArgs a;
    a.logits = logits; a.deltas = deltas_compact; a.pos = pos; a.neg = neg; a.counts = counts; a.labels = labels; a.gt = matched_gt;
    a.anchors = anchors; a.m_anchors = m_anchors; a.P = pos_cap; a.Q = neg_cap; a.C = C; a.reg_mean = reg_mean; a.cls_mean = cls_mean;
    a.eps = eps; a.clip = clip; a.reg_w = reg_weight; a.cls_w = cls_weight; a.compact = 1; a.matches = nullptr;
    a.losses = losses_out; a.g_deltas = g_deltas_out; a.g_logits = g_logits_out;
    k_detloss<<<1, 256, 0, as_stream(stream)>>>(a);
    LAUNCH_CHECK();
    return 0;
}

// Same losses with the matched GT boxes given INDIRECTLY (see DetLossArgs.matches): gt_all [G][6] = the batch's GT boxes concatenated,
// matches [B * m_anchors] = nndet_atss3d_*_batched_f32's output, gt_base_host [B] = first row of each image in gt_all.
// deltas_compact != 0: `deltas` are the [pos_cap][6] rows of the sampled positives (as nndet_detloss_compact_f32).
extern "C" int nndet_detloss_matched_f32(const float* logits, const float* deltas, int32_t deltas_compact, const int64_t* pos, int32_t pos_cap,
                                         const int64_t* neg, int32_t neg_cap, const int64_t* counts, const float* labels,
                                         const float* gt_all, const int64_t* matches, const int32_t* gt_base_host, int32_t B,
                                         const float* anchors, int64_t m_anchors, int32_t C, float eps, float clip, float reg_weight,
                                         int32_t reg_mean, float cls_weight, int32_t cls_mean, float* losses_out, float* g_deltas_out,
                                         float* g_logits_out, void* stream) {
    if (!logits || !deltas || !pos || !neg || !counts || !labels || !gt_all || !matches || !gt_base_host || !anchors || !losses_out ||
        !g_deltas_out || !g_logits_out)
        return NNDET_EINVAL;
    if (pos_cap < 1 || neg_cap < 0 || C < 1 || m_anchors < 1 || B < 1 || B > 64) return NNDET_EINVAL;
    DetLossArgs a;
    a.logits = logits; a.deltas = deltas; a.pos = pos; a.neg = neg; a.counts = counts; a.labels = labels; a.gt = gt_all;
    a.anchors = anchors; a.m_anchors = m_anchors; a.P = pos_cap; a.Q = neg_cap; a.C = C; a.reg_mean = reg_mean; a.cls_mean = cls_mean;
    a.eps = eps; a.clip = clip; a.reg_w = reg_weight; a.cls_w = cls_weight; a.compact = deltas_compact ? 1 : 0; a.matches = matches;
    for (int b = 0; b < 65; ++b) a.gt_base[b] = b < B ? gt_base_host[b] : 0;
    a.losses = losses_out; a.g_deltas = g_deltas_out; a.g_logits = g_logits_out;
    k_detloss<<<1, 256, 0, as_stream(stream)>>>(a);
    LAUNCH_CHECK();
    return 0;
}

// The same with everything the Python backward node did around it in five more launches (round 6: they sit on the serial chain between the
// loss and the backward pass): the two upstream gradients are read from their own device scalars (NULL = 0), and the scaled compact rows
// val_deltas [P, 6] / val_logits [P + Q, C] and the concatenated index list idx [P + Q] -- what the sparse consumers take instead of the
// dense tensors -- are written on the way (each may be NULL).
__global__ __launch_bounds__(256) void k_detloss_scatter2(const int64_t* __restrict__ pos, int P, const int64_t* __restrict__ neg, int Q, int C,
                                                          const float* __restrict__ g_deltas, const float* __restrict__ g_logits,
                                                          const float* __restrict__ up_reg, const float* __restrict__ up_cls,
                                                          float* __restrict__ d_deltas, float* __restrict__ d_logits,
                                                          float* __restrict__ val_deltas, float* __restrict__ val_logits, int64_t* __restrict__ idx_out) {
    const float u_reg = up_reg ? *up_reg : 0.f, u_cls = up_cls ? *up_cls : 0.f;
    for (int r = threadIdx.x; r < P + Q; r += 256) {
        const int64_t idx = r < P ? pos[r] : neg[r - P];
        if (idx_out) idx_out[r] = idx;
        if (r < P) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float v = u_reg * g_deltas[r * 6 + k];            // (the product the caller formed with a torch multiply: same rounding)
                if (val_deltas) val_deltas[r * 6 + k] = v;
                if (d_deltas && idx >= 0) d_deltas[idx * 6 + k] = v;
            }
        }
        for (int c = 0; c < C; ++c) {
            const float v = u_cls * g_logits[r * C + c];
            if (val_logits) val_logits[r * C + c] = v;
            if (idx >= 0) d_logits[idx * C + c] = v;
        }
    }
}

extern "C" int nndet_detloss_scatter2_f32(const int64_t* pos, int32_t pos_cap, const int64_t* neg, int32_t neg_cap, int32_t C,
                                          const float* g_deltas, const float* g_logits, const float* up_reg, const float* up_cls,
                                          float* d_deltas, float* d_logits, float* val_deltas, float* val_logits, int64_t* idx_out,
                                          void* stream) {
    if (!pos || !neg || !g_deltas || !g_logits || !d_logits || pos_cap < 1 || neg_cap < 0 || C < 1) return NNDET_EINVAL;
    k_detloss_scatter2<<<1, 256, 0, as_stream(stream)>>>(pos, pos_cap, neg, neg_cap, C, g_deltas, g_logits, up_reg, up_cls, d_deltas, d_logits,
                                                         val_deltas, val_logits, idx_out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_detloss_scatter_f32(const int64_t* pos, int32_t pos_cap, const int64_t* neg, int32_t neg_cap, int32_t C,
                                         const float* g_deltas, const float* g_logits, const float* upstream, float* d_deltas,
                                         float* d_logits, void* stream) {
    if (!pos || !neg || !g_deltas || !g_logits || !upstream || !d_logits || pos_cap < 1 || neg_cap < 0 || C < 1) return NNDET_EINVAL;   // d_deltas may be NULL (compact deltas)
    k_detloss_scatter<<<1, 256, 0, as_stream(stream)>>>(pos, pos_cap, neg, neg_cap, C, g_deltas, g_logits, upstream, d_deltas, d_logits);
    LAUNCH_CHECK();
    return 0;
}
