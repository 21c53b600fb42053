// Fused 2-class segmentation loss (0.5 * CE + 0.5 * SoftDice) statistics and gradient for gfx950.
// Replaces DiCESegmenterFgBg.compute_loss (nndet/arch/heads/segmenter.py:184-206,273-289) with
// SoftDiceLoss(softmax, batch_dice=True, do_bg=False) + CrossEntropyLoss (nndet/losses/segmentation.py:32-151):
// one streaming pass produces {sum CE, tp, fp, fn}; the scalar loss algebra on those 4 numbers stays in the
// caller (autograd), whose gradient w.r.t. the 4 sums drives the second streaming pass that writes dlogits.
// logits: [nvox][c_p] NDHWC, channels 0 (background) and 1 (foreground) are used. HBM-bound.
#include "common.h"

template <typename T> __device__ __forceinline__ void ld2(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void ld2<float>(const float* p, float& a, float& b) {
    const float2 v = *reinterpret_cast<const float2*>(p); a = v.x; b = v.y;
}
template <> __device__ __forceinline__ void ld2<bf16_t>(const bf16_t* p, float& a, float& b) {
    const uint32_t v = *reinterpret_cast<const uint32_t*>(p);
    a = __uint_as_float(v << 16); b = __uint_as_float(v & 0xffff0000u);
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
    if (sizeof(T) == 2) first.x = pack_bf16x2(-d1, d1);
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
    else k_segloss_fwd<float><<<(unsigned)nb, 256, 0, as_stream(stream)>>>((const float*)logits, target, nvox, c_p, sums_out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_segloss_backward(int32_t dtype, const void* logits, const uint8_t* target, int64_t nvox, int32_t c_p,
                                      const float* coeffs, void* dlogits, void* stream) {
    if (!logits || !target || !coeffs || !dlogits || nvox <= 0 || c_p % 32) return NNDET_EINVAL;
    const unsigned nb = (unsigned)ceil_div64(nvox, 256);
    if (dtype == NNDET_BF16) k_segloss_bwd<bf16_t><<<nb, 256, 0, as_stream(stream)>>>((const bf16_t*)logits, target, nvox, c_p, coeffs, (bf16_t*)dlogits);
    else k_segloss_bwd<float><<<nb, 256, 0, as_stream(stream)>>>((const float*)logits, target, nvox, c_p, coeffs, (float*)dlogits);
    LAUNCH_CHECK();
    return 0;
}
