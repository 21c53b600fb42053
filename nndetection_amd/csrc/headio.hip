// Detection-head output gather for gfx950: the per-level `permute(0, 2, 3, 4, 1).contiguous().view(N, -1, last)` of the classifier /
// regressor (nndet/arch/heads/classifier.py:176-181, regressor.py:165-172, the level-specific `Scale` of regressor.py:163-164 and
// layers/scale.py:21-43) and the `torch.cat(..., dim=1)` over the pyramid levels (nndet/arch/heads/comb.py:107-108) as ONE pass.
// The conv output is NDHWC with the channels padded to 32: the permuted view IS the buffer, minus the padding. One launch reads the
// padded bf16 / fp32 rows of every level and writes fp32 [N, sum_l points_l * A, last] (+ scale); the backward launch reads the
// gradient of that tensor and writes the padded NDHWC gradients of the levels (zeros in the padding) and d(scale_l) = sum(g * y).
// HBM-bound element-wise work: consecutive lanes read consecutive channels of the padded rows and write consecutive floats.
#include "common.h"
#include <string.h>

struct HgArgs {
    const void* y[NNDET_HEAD_MAX_LEVELS];
    void* dy[NNDET_HEAD_MAX_LEVELS];
    const float* scale[NNDET_HEAD_MAX_LEVELS];
    float* dscale[NNDET_HEAD_MAX_LEVELS];
    int64_t pts[NNDET_HEAD_MAX_LEVELS];     // spatial positions of the level (per image)
    int64_t off[NNDET_HEAD_MAX_LEVELS];     // first row (of `cout` values) of the level inside an image's block of the gathered tensor
    int64_t rows_total;                     // sum of pts
    int32_t N, cout, cout_p;
    uint32_t m_cout, m_coutp;               // magic multipliers: i / cout == umulhi(i, m_cout) for i < 2^32 / cout
};

static uint32_t magic_u32(int d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)d + 1ull); }
__device__ __forceinline__ uint32_t mdivu(uint32_t n, uint32_t m) { return m ? __umulhi(n, m) : n; }

// grid (chunks, level, image); a workgroup handles PB consecutive positions of one (level, image)
#define HG_PB 64
template <typename T>
__global__ __launch_bounds__(256) void k_head_gather(const HgArgs A, float* __restrict__ out) {
    const int l = blockIdx.y, n = blockIdx.z;
    const int64_t p0 = (int64_t)blockIdx.x * HG_PB;
    if (p0 >= A.pts[l]) return;
    const int np = (int)min((int64_t)HG_PB, A.pts[l] - p0);
    const T* __restrict__ y = reinterpret_cast<const T*>(A.y[l]) + ((int64_t)n * A.pts[l] + p0) * A.cout_p;
    float* __restrict__ o = out + ((int64_t)n * A.rows_total + A.off[l] + p0) * A.cout;
    const float sc = A.scale[l] ? *A.scale[l] : 1.f;
    const uint32_t tot = (uint32_t)np * (uint32_t)A.cout;
    for (uint32_t i = threadIdx.x; i < tot; i += 256) {
        const uint32_t p = mdivu(i, A.m_cout), c = i - p * A.cout;
        const float v = Elem<T>::ld(y[(int64_t)p * A.cout_p + c]);
        o[i] = A.scale[l] ? v * sc : v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_head_gather_bwd(const HgArgs A, const float* __restrict__ g) {
    const int l = blockIdx.y, n = blockIdx.z;
    const int64_t p0 = (int64_t)blockIdx.x * HG_PB;
    if (p0 >= A.pts[l]) return;
    const int np = (int)min((int64_t)HG_PB, A.pts[l] - p0);
    const int64_t row0 = ((int64_t)n * A.pts[l] + p0) * A.cout_p;
    T* __restrict__ dy = reinterpret_cast<T*>(A.dy[l]) + row0;
    const T* __restrict__ y = A.dscale[l] ? reinterpret_cast<const T*>(A.y[l]) + row0 : nullptr;
    const float* __restrict__ gi = g + ((int64_t)n * A.rows_total + A.off[l] + p0) * A.cout;
    const float sc = A.scale[l] ? *A.scale[l] : 1.f;
    const uint32_t tot = (uint32_t)np * (uint32_t)A.cout_p;
    float ds = 0.f;
    for (uint32_t i = threadIdx.x; i < tot; i += 256) {
        const uint32_t p = mdivu(i, A.m_coutp), c = i - p * A.cout_p;
        float v = 0.f;
        if ((int)c < A.cout) {
            const float gv = gi[(int64_t)p * A.cout + c];
            if (y) ds += gv * Elem<T>::ld(y[i]);
            v = A.scale[l] ? gv * sc : gv;
        }
        dy[i] = Elem<T>::st(v);
    }
    if (A.dscale[l]) {       // d(scale) = sum(g * y): wave sums, one atomic per wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ds += __shfl_xor(ds, o, 64);
        if ((threadIdx.x & 63) == 0 && ds != 0.f) atomicAdd(A.dscale[l], ds);
    }
}

static int hg_fill(HgArgs* a, const NndetHeadLevels* lv, int32_t N, int32_t cout, int32_t cout_p, int64_t* max_pts) {
    if (!lv || lv->nlev < 1 || lv->nlev > NNDET_HEAD_MAX_LEVELS || N < 1 || cout < 1 || cout_p < cout || (cout_p % 32) != 0) return NNDET_EINVAL;
    memset(a, 0, sizeof(*a));
    int64_t off = 0, mx = 0;
    for (int l = 0; l < lv->nlev; ++l) {
        if (!lv->y[l] || lv->points[l] < 1) return NNDET_EINVAL;
        a->y[l] = lv->y[l]; a->dy[l] = lv->dy[l]; a->scale[l] = lv->scale[l]; a->dscale[l] = lv->dscale[l];
        a->pts[l] = lv->points[l]; a->off[l] = off;
        off += lv->points[l];
        if (lv->points[l] > mx) mx = lv->points[l];
    }
    if ((int64_t)HG_PB * cout_p >= (1LL << 31)) return NNDET_EINVAL;
    a->rows_total = off; a->N = N; a->cout = cout; a->cout_p = cout_p;
    a->m_cout = magic_u32(cout); a->m_coutp = magic_u32(cout_p);
    *max_pts = mx;
    return 0;
}

extern "C" int nndet_head_gather_f32(int32_t dtype, const NndetHeadLevels* lv, int32_t N, int32_t cout, int32_t cout_p, float* out,
                                     void* stream) {
    HgArgs a;
    int64_t mx = 0;
    int rc = hg_fill(&a, lv, N, cout, cout_p, &mx);
    if (rc) return rc;
    if (!out || (dtype != NNDET_BF16 && dtype != NNDET_F32 && dtype != NNDET_F16)) return NNDET_EINVAL;
    const dim3 grid((unsigned)ceil_div64(mx, HG_PB), (unsigned)lv->nlev, (unsigned)N);
    if (dtype == NNDET_BF16) k_head_gather<bf16_t><<<grid, 256, 0, as_stream(stream)>>>(a, out);
    else if (dtype == NNDET_F16) k_head_gather<f16_t><<<grid, 256, 0, as_stream(stream)>>>(a, out);
    else k_head_gather<float><<<grid, 256, 0, as_stream(stream)>>>(a, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_head_gather_backward(int32_t dtype, const NndetHeadLevels* lv, int32_t N, int32_t cout, int32_t cout_p,
                                          const float* grad_out, void* stream) {
    HgArgs a;
    int64_t mx = 0;
    int rc = hg_fill(&a, lv, N, cout, cout_p, &mx);
    if (rc) return rc;
    if (!grad_out || (dtype != NNDET_BF16 && dtype != NNDET_F32 && dtype != NNDET_F16)) return NNDET_EINVAL;
    for (int l = 0; l < lv->nlev; ++l) if (!lv->dy[l] || (lv->dscale[l] && !lv->scale[l])) return NNDET_EINVAL;
    const dim3 grid((unsigned)ceil_div64(mx, HG_PB), (unsigned)lv->nlev, (unsigned)N);
    if (dtype == NNDET_BF16) k_head_gather_bwd<bf16_t><<<grid, 256, 0, as_stream(stream)>>>(a, grad_out);
    else if (dtype == NNDET_F16) k_head_gather_bwd<f16_t><<<grid, 256, 0, as_stream(stream)>>>(a, grad_out);
    else k_head_gather_bwd<float><<<grid, 256, 0, as_stream(stream)>>>(a, grad_out);
    LAUNCH_CHECK();
    return 0;
}
