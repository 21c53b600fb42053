// Fused 2-class segmentation loss (0.5 * CE + 0.5 * SoftDice) statistics and gradient for gfx950.
// Replaces DiCESegmenterFgBg.compute_loss (nndet/arch/heads/segmenter.py:184-206,273-289) with
// SoftDiceLoss(softmax, batch_dice=True, do_bg=False) + CrossEntropyLoss (nndet/losses/segmentation.py:32-151):
// one streaming pass produces {sum CE, tp, fp, fn}; the scalar loss algebra on those 4 numbers stays in the
// caller (autograd), whose gradient w.r.t. the 4 sums drives the second streaming pass that writes dlogits.
// logits: [nvox][c_p] NDHWC, channels 0 (background) and 1 (foreground) are used. HBM-bound.
#include "common.h"

template <typename T> __device__ __forceinline__ void ld2(const T* p, float& a, float& b) {      // 16-bit types
    const uint32_t v = *reinterpret_cast<const uint32_t*>(p);
    a = H16<T>::lo(v); b = H16<T>::hi(v);
}
template <> __device__ __forceinline__ void ld2<float>(const float* p, float& a, float& b) {
    const float2 v = *reinterpret_cast<const float2*>(p); a = v.x; b = v.y;
}

__device__ __forceinline__ float softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

template <typename T>
__global__ __launch_bounds__(256) void k_segloss_fwd(const T* __restrict__ logits, const uint8_t* __restrict__ target,
                                                     int64_t nvox, int c_p, double* __restrict__ sums) {
    __shared__ double red[4];
    if (threadIdx.x < 4) red[threadIdx.x] = 0.0;
    __syncthreads();
    float ce = 0.f, tp = 0.f, fp = 0.f, fn = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * 256) {
        float l0, l1;
        ld2<T>(logits + v * c_p, l0, l1);
        const float z = l1 - l0;
        const bool t = target[v] > 0;
        const float p1 = 1.f / (1.f + expf(-z));
        ce += t ? softplus(-z) : softplus(z);          // -log softmax(l)[t]
        if (t) { tp += p1; fn += 1.f - p1; } else { fp += p1; }
    }
    double d[4] = {(double)ce, (double)tp, (double)fp, (double)fn};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double s = wave_sum_f64(d[k]);
        if ((threadIdx.x & 63) == 0) atomicAdd(&red[k], s);
    }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&sums[threadIdx.x], red[threadIdx.x]);
}

// one thread per voxel writes the whole c_p-channel row (channels >= 2 are zero)
template <typename T>
__global__ __launch_bounds__(256) void k_segloss_bwd(const T* __restrict__ logits, const uint8_t* __restrict__ target,
                                                     int64_t nvox, int c_p, const float* __restrict__ coeffs,
                                                     T* __restrict__ dlogits) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvox) return;
    const float g_ce = coeffs[0], g_tp = coeffs[1], g_fp = coeffs[2], g_fn = coeffs[3];
    float l0, l1;
    ld2<T>(logits + v * c_p, l0, l1);
    const float z = l1 - l0;
    const bool t = target[v] > 0;
    const float p1 = 1.f / (1.f + expf(-z));
    const float dp = p1 * (1.f - p1);                      // d p1 / d l1 = - d p1 / d l0
    float d1 = g_ce * (p1 - (t ? 1.f : 0.f));
    d1 += dp * (t ? (g_tp - g_fn) : g_fp);
    T* o = dlogits + v * c_p;
    constexpr int E = 16 / (int)sizeof(T);
    uint4 first = make_uint4(0u, 0u, 0u, 0u);
    if (sizeof(T) == 2) first.x = H16<T>::pack2(-d1, d1);
    else { first.x = __float_as_uint(-d1); first.y = __float_as_uint(d1); }
    reinterpret_cast<uint4*>(o)[0] = first;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    for (int i = 1; i < c_p / E; ++i) reinterpret_cast<uint4*>(o)[i] = zero;
}

extern "C" int nndet_segloss_forward(int32_t dtype, const void* logits, const uint8_t* target, int64_t nvox, int32_t c_p,
                                     double* sums_out, void* stream) {
    if (!logits || !target || !sums_out || nvox <= 0 || c_p % 32) return NNDET_EINVAL;
    int64_t nb = ceil_div64(nvox, 256 * 8);
    if (nb > 4096) nb = 4096;
    if (dtype == NNDET_BF16) k_segloss_fwd<bf16_t><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const bf16_t*)logits, target, nvox, c_p, sums_out);
    else if (dtype == NNDET_F16) k_segloss_fwd<f16_t><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const f16_t*)logits, target, nvox, c_p, sums_out);
    else k_segloss_fwd<float><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const float*)logits, target, nvox, c_p, sums_out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_segloss_backward(int32_t dtype, const void* logits, const uint8_t* target, int64_t nvox, int32_t c_p,
                                      const float* coeffs, void* dlogits, void* stream) {
    if (!logits || !target || !coeffs || !dlogits || nvox <= 0 || c_p % 32) return NNDET_EINVAL;
    const unsigned nb = (unsigned)ceil_div64(nvox, 256);
    if (dtype == NNDET_BF16) k_segloss_bwd<bf16_t><<<nb, 256, 0, as_stream(stream)>>>((const bf16_t*)logits, target, nvox, c_p, coeffs, (bf16_t*)dlogits);
    else if (dtype == NNDET_F16) k_segloss_bwd<f16_t><<<nb, 256, 0, as_stream(stream)>>>((const f16_t*)logits, target, nvox, c_p, coeffs, (f16_t*)dlogits);
    else k_segloss_bwd<float><<<nb, 256, 0, as_stream(stream)>>>((const float*)logits, target, nvox, c_p, coeffs, (float*)dlogits);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ fused head (training)
template <typename T> __device__ __forceinline__ float round_to(float v) { return H16<T>::lo(H16<T>::pack2(v, 0.f)); }    // 16-bit types
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }

// loads the 32-channel row of voxel v as fp32
template <typename T> __device__ __forceinline__ void ld_row32(const T* p, float* v) {      // 16-bit types
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 u = reinterpret_cast<const uint4*>(p)[i];
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[8 * i + 2 * k] = H16<T>::lo(w[k]); v[8 * i + 2 * k + 1] = H16<T>::hi(w[k]); }
    }
}
template <> __device__ __forceinline__ void ld_row32<float>(const float* p, float* v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float4 f = reinterpret_cast<const float4*>(p)[i]; v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
}

struct SegHeadW { float w0[32], w1[32], b0, b1; };

template <typename T>
__device__ __forceinline__ float seghead_z(const float* xv, const SegHeadW& W) {
    float l0 = 0.f, l1 = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) { l0 = fmaf(W.w0[c], xv[c], l0); l1 = fmaf(W.w1[c], xv[c], l1); }
    l0 = round_to<T>(l0 + W.b0); l1 = round_to<T>(l1 + W.b1);
    return l1 - l0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_seghead_fwd(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                     int cin, const uint8_t* __restrict__ target, int64_t nvox, double* __restrict__ sums) {
    __shared__ double red[4];
    __shared__ SegHeadW W;
    if (threadIdx.x < 4) red[threadIdx.x] = 0.0;
    if (threadIdx.x < 32) {
        W.w0[threadIdx.x] = (int)threadIdx.x < cin ? w[threadIdx.x] : 0.f;
        W.w1[threadIdx.x] = (int)threadIdx.x < cin ? w[cin + threadIdx.x] : 0.f;
    }
    if (threadIdx.x == 0) { W.b0 = bias ? bias[0] : 0.f; W.b1 = bias ? bias[1] : 0.f; }
    __syncthreads();
    float ce = 0.f, tp = 0.f, fp = 0.f, fn = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * 256) {
        float xv[32];
        ld_row32<T>(x + v * 32, xv);
        const float z = seghead_z<T>(xv, W);
        const bool t = target[v] > 0;
        const float p1 = 1.f / (1.f + expf(-z));
        ce += t ? softplus(-z) : softplus(z);
        if (t) { tp += p1; fn += 1.f - p1; } else { fp += p1; }
    }
    double d[4] = {(double)ce, (double)tp, (double)fp, (double)fn};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double s = wave_sum_f64(d[k]);
        if ((threadIdx.x & 63) == 0) atomicAdd(&red[k], s);
    }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&sums[threadIdx.x], red[threadIdx.x]);
}

template <typename T> __device__ __forceinline__ void st_row32(T* p, const float* v) {      // 16-bit types
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint4 u;
        u.x = H16<T>::pack2(v[8 * i + 0], v[8 * i + 1]); u.y = H16<T>::pack2(v[8 * i + 2], v[8 * i + 3]);
        u.z = H16<T>::pack2(v[8 * i + 4], v[8 * i + 5]); u.w = H16<T>::pack2(v[8 * i + 6], v[8 * i + 7]);
        reinterpret_cast<uint4*>(p)[i] = u;
    }
}
template <> __device__ __forceinline__ void st_row32<float>(float* p, const float* v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) reinterpret_cast<float4*>(p)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

// persistent grid: every thread accumulates sum(d1 * x[c]) for its voxels in registers, block reduction through LDS,
// one fp64 atomic per (block, channel)
// RANK1: the two logit gradients of the fg / bg softmax are d1 and -d1, so the gradient w.r.t. the input is the OUTER PRODUCT
// d1[voxel] * (w1 - w0)[channel]. Instead of writing it (32 channels, 629 MB at 160x160x96, batch 4) the kernel writes d1 alone
// (one value per voxel); the producer convolution's backward pass then works on the factorised form (arch/conv.py: _rank1_backward).
template <typename T, bool RANK1 = false>
__global__ __launch_bounds__(256) void k_seghead_bwd(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                     int cin, const uint8_t* __restrict__ target, int64_t nvox,
                                                     const float* __restrict__ coeffs, T* __restrict__ dx, double* __restrict__ dwb) {
    __shared__ SegHeadW W;
    __shared__ float red[4][33];
    if (threadIdx.x < 32) {
        W.w0[threadIdx.x] = (int)threadIdx.x < cin ? w[threadIdx.x] : 0.f;
        W.w1[threadIdx.x] = (int)threadIdx.x < cin ? w[cin + threadIdx.x] : 0.f;
    }
    if (threadIdx.x == 0) { W.b0 = bias ? bias[0] : 0.f; W.b1 = bias ? bias[1] : 0.f; }
    __syncthreads();
    const float g_ce = coeffs[0], g_tp = coeffs[1], g_fp = coeffs[2], g_fn = coeffs[3];
    float acc[33];
#pragma unroll
    for (int c = 0; c < 33; ++c) acc[c] = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * 256) {
        float xv[32];
        ld_row32<T>(x + v * 32, xv);
        const float z = seghead_z<T>(xv, W);
        const bool t = target[v] > 0;
        const float p1 = 1.f / (1.f + expf(-z));
        const float dp = p1 * (1.f - p1);
        float d1 = g_ce * (p1 - (t ? 1.f : 0.f));
        d1 += dp * (t ? (g_tp - g_fn) : g_fp);
        d1 = round_to<T>(d1);                      // the unfused path stores dlogits in T
        if constexpr (RANK1) {
#pragma unroll
            for (int c = 0; c < 32; ++c) acc[c] = fmaf(d1, xv[c], acc[c]);
            dx[v] = Elem<T>::st(d1);               // (d1 is already a value of T: exact)
        } else {
            float o[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                acc[c] = fmaf(d1, xv[c], acc[c]);
                o[c] = fmaf(W.w1[c], d1, W.w0[c] * -d1);
            }
            st_row32<T>(dx + v * 32, o);
        }
        acc[32] += d1;
    }
    // reduce 33 values over the block: wave shuffles, then the 4 waves through LDS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 33; ++c) {
        const float s = wave_sum_f32(acc[c]);
        if (lane == 0) red[wv][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 33) {
        const int c = threadIdx.x;
        const double s = (double)red[0][c] + (double)red[1][c] + (double)red[2][c] + (double)red[3][c];
        if (c < 32) {
            if (c < cin) { atomicAdd(&dwb[cin + c], s); atomicAdd(&dwb[c], -s); }     // dW[1][c] = s, dW[0][c] = -s
        } else {
            atomicAdd(&dwb[2 * cin + 1], s); atomicAdd(&dwb[2 * cin], -s);
        }
    }
}

extern "C" int nndet_seghead_forward(int32_t dtype, const void* x, int32_t c_p, int32_t cin, const float* w, const float* bias,
                                     const uint8_t* target, int64_t nvox, double* sums_out, void* stream) {
    if (!x || !w || !target || !sums_out || nvox <= 0 || c_p != 32 || cin <= 0 || cin > 32) return NNDET_EINVAL;
    int64_t nb = ceil_div64(nvox, 256 * 4);
    if (nb > 2048) nb = 2048;
    if (dtype == NNDET_BF16) k_seghead_fwd<bf16_t><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const bf16_t*)x, w, bias, cin, target, nvox, sums_out);
    else if (dtype == NNDET_F16) k_seghead_fwd<f16_t><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const f16_t*)x, w, bias, cin, target, nvox, sums_out);
    else k_seghead_fwd<float><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const float*)x, w, bias, cin, target, nvox, sums_out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_seghead_backward(int32_t dtype, const void* x, int32_t c_p, int32_t cin, const float* w, const float* bias,
                                      const uint8_t* target, int64_t nvox, const float* coeffs, void* dx, double* dwb_out, void* stream) {
    if (!x || !w || !target || !coeffs || !dx || !dwb_out || nvox <= 0 || c_p != 32 || cin <= 0 || cin > 32) return NNDET_EINVAL;
    int64_t nb = ceil_div64(nvox, 256 * 4);
    if (nb > 2048) nb = 2048;
    if (dtype == NNDET_BF16)
        k_seghead_bwd<bf16_t><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const bf16_t*)x, w, bias, cin, target, nvox, coeffs, (bf16_t*)dx, dwb_out);
    else if (dtype == NNDET_F16)
        k_seghead_bwd<f16_t><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const f16_t*)x, w, bias, cin, target, nvox, coeffs, (f16_t*)dx, dwb_out);
    else
        k_seghead_bwd<float><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const float*)x, w, bias, cin, target, nvox, coeffs, (float*)dx, dwb_out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_seghead_backward_rank1(int32_t dtype, const void* x, int32_t c_p, int32_t cin, const float* w, const float* bias,
                                            const uint8_t* target, int64_t nvox, const float* coeffs, void* d1_out, double* dwb_out,
                                            void* stream) {
    if (!x || !w || !target || !coeffs || !d1_out || !dwb_out || nvox <= 0 || c_p != 32 || cin <= 0 || cin > 32) return NNDET_EINVAL;
    int64_t nb = ceil_div64(nvox, 256 * 4);
    if (nb > 2048) nb = 2048;
    if (dtype == NNDET_BF16)
        k_seghead_bwd<bf16_t, true><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const bf16_t*)x, w, bias, cin, target, nvox, coeffs, (bf16_t*)d1_out, dwb_out);
    else if (dtype == NNDET_F16)
        k_seghead_bwd<f16_t, true><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const f16_t*)x, w, bias, cin, target, nvox, coeffs, (f16_t*)d1_out, dwb_out);
    else
        k_seghead_bwd<float, true><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const float*)x, w, bias, cin, target, nvox, coeffs, (float*)d1_out, dwb_out);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ scalar tail of the segmentation loss
// DiCESegmenterFgBg.compute_loss after the per-voxel sums (nndet/arch/heads/segmenter.py:184-206; losses/segmentation.py:32-151):
//   seg_ce = alpha * CE_sum / nvox,  seg_dice = (1 - alpha) * (1 - (2 tp + s_nom) / (2 tp + fp + fn + s_den))
// and the 2 x 4 Jacobian d(losses) / d(sums) in ONE tiny launch: as torch scalar algebra this was ~45 launches of one element each
// (forward + autograd), 0.3-0.5 ms of launch latency between the two streaming passes of the segmentation head.
__global__ void k_segloss_tail(const float* __restrict__ s, float nvox, float alpha, float sn, float sd, float* __restrict__ losses,
                               float* __restrict__ coeffs) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float ce = s[0] / nvox, tp = s[1], fp = s[2], fn = s[3];
    const float num = 2.f * tp + sn, den = 2.f * tp + fp + fn + sd;
    const float dc = num / den;
    losses[0] = alpha * ce;
    losses[1] = (1.f - alpha) * (1.f - dc);
    const float k = -(1.f - alpha);                  // d seg_dice / d dc
    coeffs[0] = alpha / nvox; coeffs[1] = 0.f; coeffs[2] = 0.f; coeffs[3] = 0.f;
    coeffs[4] = 0.f;
    coeffs[5] = k * (2.f * den - 2.f * num) / (den * den);      // d dc / d tp = (2 den - 2 num) / den^2
    coeffs[6] = k * (-num / (den * den));                       // d dc / d fp
    coeffs[7] = k * (-num / (den * den));                       // d dc / d fn
}

extern "C" int nndet_segloss_tail_f32(const float* sums, int64_t nvox, float alpha, float smooth_nom, float smooth_denom,
                                      float* losses_out, float* coeffs_out, void* stream) {
    if (!sums || !losses_out || !coeffs_out || nvox <= 0) return NNDET_EINVAL;
    k_segloss_tail<<<1, 64, 0, as_stream(stream)>>>(sums, (float)nvox, alpha, smooth_nom, smooth_denom, losses_out, coeffs_out);
    LAUNCH_CHECK();
    return 0;
}
