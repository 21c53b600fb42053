// InstanceNorm3d / GroupNorm (+ReLU) forward/backward and column sums on NDHWC for gfx950.
// Replaces nn.InstanceNorm3d / nndet GroupNorm (nndet/arch/layers/norm.py:26-50) + nn.ReLU inside
// ConvInstanceRelu / ConvGroupRelu (nndet/arch/conv.py:146-294). All kernels are HBM-bound streaming
// passes with 16-byte loads/stores; statistics are accumulated in fp64 (fp32 inside a thread).
//
// Thread mapping used everywhere: a row = one voxel's c_p channels = PPR pieces of 16 bytes; thread t of a
// workgroup owns piece (t % PPR) of rows (t / PPR) + k * (256 / PPR), so its channels never change and the
// per-channel scale/shift live in registers.
#include "common.h"
#include "conv_common.h"

template <typename T> struct Vec16 {      // the 16-bit storage types (bf16_t, f16_t)
    static constexpr int E = 8;
    using raw_t = uint4;                  // one 16-byte piece as loaded; `cvt` widens it (split so that several loads can be in flight)
    __device__ static __forceinline__ raw_t ldraw(const T* p) { return *reinterpret_cast<const uint4*>(p); }
    __device__ static __forceinline__ void cvt(const raw_t& u, float* v) {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = H16<T>::lo(w[i]); v[2 * i + 1] = H16<T>::hi(w[i]); }
    }
    __device__ static __forceinline__ void ld(const T* p, float* v) { cvt(ldraw(p), v); }
    __device__ static __forceinline__ void st(T* p, const float* v) {
        uint4 u;
        u.x = H16<T>::pack2(v[0], v[1]);
        u.y = H16<T>::pack2(v[2], v[3]);
        u.z = H16<T>::pack2(v[4], v[5]);
        u.w = H16<T>::pack2(v[6], v[7]);
        *reinterpret_cast<uint4*>(p) = u;
    }
};
template <> struct Vec16<float> {
    static constexpr int E = 4;
    using raw_t = float4;
    __device__ static __forceinline__ raw_t ldraw(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static __forceinline__ void cvt(const raw_t& f, float* v) { v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
    __device__ static __forceinline__ void ld(const float* p, float* v) { cvt(ldraw(p), v); }
    __device__ static __forceinline__ void st(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
// Ragged batch (NndetItems): per-item voxel count and first row; n == 0 = uniform batch (image n = rows [n * spatial, (n + 1) * spatial))
struct NormItems {
    int32_t n, pad_;
    int32_t spatial[NNDET_MAX_ITEMS];
    int64_t row_off[NNDET_MAX_ITEMS];
};
static const NormItems g_norm_uniform = {};

// Per-thread constants of the element-wise / reduction kernels: (mean, rstd) of the thread's E channels from the [N][c_p][2] table
// (written for all c_p channels, zeros beyond c) and gamma / beta ([c]: clamped index, masked afterwards). ALL loads are issued
// unconditionally and independently: the `if (ci < c) load` form compiled to one exec-masked load + s_waitcnt vmcnt(0) per element,
// 32 dependent round trips (~30-60 us) in front of every workgroup's first row (round 5; the 5x5x6 level's reduction took 24-39 us).
template <int E>
__device__ __forceinline__ void norm_consts(const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, int n, int c, int c_p, int c0,
                                            float* mu, float* rs, float* ga, float* be) {
    const float2* mr = reinterpret_cast<const float2*>(mean_rstd) + ((int64_t)n * c_p + c0);
    float2 m[E];
    float g[E], b[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int cj = min(c0 + e, c - 1);
        m[e] = mr[e];
        g[e] = gamma[cj];
        b[e] = beta[cj];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool ok = c0 + e < c;
        mu[e] = ok ? m[e].x : 0.f;
        rs[e] = ok ? m[e].y : 0.f;
        ga[e] = ok ? g[e] : 0.f;
        be[e] = ok ? b[e] : 0.f;
    }
}

#define ROWS_PER_BLOCK 512    // max rows (voxels) handled by one workgroup in the element-wise (apply) kernels
#ifndef RED_U
#define RED_U 4               // rows per thread in flight in the reduction kernels (k_norm_bwd_reduce, k_norm_stats, k_colsum)
#endif
// rows per workgroup of the element-wise kernels: 512 for the big layers; the pyramid levels P3-P5 (150 ... 4800 rows per image) got
// 1-10 workgroups per image of 32 dependent load -> store iterations each (24-33 us for 0.6-5 MB): aim at >= ~1024 workgroups
static inline int apply_rows(int64_t rows_total, int c_p, int esz) {
    const int rpi = 256 / (c_p * esz / 16);          // rows covered by one iteration of the workgroup
    int64_t r = rows_total / 1024;
    if (r > ROWS_PER_BLOCK) r = ROWS_PER_BLOCK;
    const int lo = rpi > 0 ? 2 * rpi : 32;
    if (r < lo) r = lo;
    return (int)((r + 31) / 32 * 32);
}
// rows per workgroup in the reduction kernels: as many as possible (fewer LDS / global fp64 atomics per byte) while
// keeping >= ~1024 workgroups in flight; a fixed 4096 starved the small layers (40 workgroups on 256 CUs)
static inline int red_rows(int64_t rows_total) {
    int64_t r = rows_total / 1024;
    if (r < 64) r = 64;        // (256 before: 1-19 workgroups per image on the small pyramid levels, 18-45 us per launch)
    if (r > 4096) r = 4096;
    return (int)r;
}

// End of a reduction workgroup: per-thread partial sums (E channels x 2 sums) -> the workgroup's LDS table [c_p][2] in fp64. The threads of a
// channel group (same tid % ppr) used to add their partials with LDS atomics one by one: 256 / ppr threads on the same 2 E addresses -- with
// c_p = 32 that is a 64-way serialised chain of 16 fp64 atomics per thread, and the LDS was busy 29 % of k_norm_bwd_reduce's time with 47 %
// bank conflicts (profiles/round5_step_pmc_survey.txt). When ppr is a power of two the lanes of a wave that share a channel group are a
// fixed stride apart: they are added with xor shuffles first (fp32, a 16- ... 2-leaf tree), and only the first ppr lanes of each wave touch the table.
template <int E>
__device__ __forceinline__ void norm_block_sums_to_lds(double* red, int ppr, int cp, float* sa, float* sb) {
    if ((ppr & (ppr - 1)) == 0 && ppr <= 32) {
        for (int o = ppr; o < 64; o <<= 1) {
#pragma unroll
            for (int e = 0; e < E; ++e) { sa[e] += __shfl_xor(sa[e], o, 64); sb[e] += __shfl_xor(sb[e], o, 64); }
        }
        if ((int)(threadIdx.x & 63) >= ppr) return;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        atomicAdd(&red[(cp * E + e) * 2 + 0], (double)sa[e]);
        atomicAdd(&red[(cp * E + e) * 2 + 1], (double)sb[e]);
    }
}

// ------------------------------------------------------------------ statistics: sum / sumsq per (n, channel)
// grid (ceil(spatial / ROWS_PER_BLOCK), N)
template <typename T>
__global__ __launch_bounds__(256) void k_norm_stats(const T* __restrict__ x, int64_t spatial, int c_p, int N,
                                                    double* __restrict__ stats, int RED_ROWS) {
    constexpr int E = Vec16<T>::E;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem);   // [c_p][2]
    const int ppr = c_p / E;
    const int rpi = 256 / ppr;                        // rows per iteration
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < c_p * 2; i += 256) red[i] = 0.0;
    __syncthreads();
    const int cp = threadIdx.x % ppr, rr = threadIdx.x / ppr;
    float s[E], s2[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { s[e] = 0.f; s2[e] = 0.f; }
    if (rr < rpi) {
        const int64_t r0 = (int64_t)blockIdx.x * RED_ROWS;
        const int64_t r1 = min(r0 + RED_ROWS, spatial);
        const T* xb = x + ((int64_t)n * spatial) * c_p + cp * E;
        // RED_U rows of this thread in flight per iteration (see k_norm_bwd_reduce); same summation order as the plain loop
        int64_t r = r0 + rr;
        for (; r + (int64_t)(RED_U - 1) * rpi < r1; r += (int64_t)RED_U * rpi) {
            typename Vec16<T>::raw_t raw[RED_U];
#pragma unroll
            for (int u = 0; u < RED_U; ++u) raw[u] = Vec16<T>::ldraw(xb + (r + (int64_t)u * rpi) * c_p);
#pragma unroll
            for (int u = 0; u < RED_U; ++u) {
                float v[E];
                Vec16<T>::cvt(raw[u], v);
#pragma unroll
                for (int e = 0; e < E; ++e) { s[e] += v[e]; s2[e] += v[e] * v[e]; }
            }
        }
        for (; r < r1; r += rpi) {
            float v[E];
            Vec16<T>::ld(xb + r * c_p, v);
#pragma unroll
            for (int e = 0; e < E; ++e) { s[e] += v[e]; s2[e] += v[e] * v[e]; }
        }
    }
    if (rr < rpi || 256 % ppr == 0) norm_block_sums_to_lds<E>(red, ppr, cp, s, s2);      // (256 % ppr == 0: every thread has rows, the shuffles need all lanes)
    __syncthreads();
    const int rep = blockIdx.x % NNDET_STATS_REPLICAS;
    double* dst = stats + ((int64_t)rep * N + n) * c_p * 2;
    for (int i = threadIdx.x; i < c_p * 2; i += 256) atomicAdd(&dst[i], red[i]);
}

int norm_stats_run(int dtype, const void* x, int batch, int64_t spatial, int c_p, double* stats, hipStream_t st) {
    if (c_p % 32 || c_p > 1024 || batch <= 0 || spatial <= 0) return NNDET_EINVAL;
    const int rr = red_rows(spatial * batch);
    dim3 grid((unsigned)ceil_div64(spatial, rr), batch);
    const size_t lds = (size_t)c_p * 16;
    if (dtype == NNDET_BF16) k_norm_stats<bf16_t><<<grid, 256, lds, st>>>((const bf16_t*)x, spatial, c_p, batch, stats, rr);
    else if (dtype == NNDET_F16) k_norm_stats<f16_t><<<grid, 256, lds, st>>>((const f16_t*)x, spatial, c_p, batch, stats, rr);
    else k_norm_stats<float><<<grid, 256, lds, st>>>((const float*)x, spatial, c_p, batch, stats, rr);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_norm_stats(int32_t dtype, const void* x, int32_t batch, int64_t spatial, int32_t c_p, double* stats,
                                void* stream) {
    if (!x || !stats) return NNDET_EINVAL;
    return norm_stats_run(dtype, x, batch, spatial, c_p, stats, as_stream(stream));
}

// ------------------------------------------------------------------ finalize: replicas -> per-channel (mean, rstd) of its group
// grid N, block 256 (loops channels)
__global__ void k_norm_finalize(const double* __restrict__ stats, int N, int c, int c_p, int groups, int64_t spatial,
                                float eps, float* __restrict__ mean_rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                float* __restrict__ scale_shift, const NormItems IT) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ch = reinterpret_cast<double*>(smem);   // [c_p][2]
    const int n = blockIdx.x;
    if (IT.n) spatial = IT.spatial[n];
    for (int i = threadIdx.x; i < c_p * 2; i += blockDim.x) {
        double v = 0.0;
        for (int r = 0; r < NNDET_STATS_REPLICAS; ++r) v += stats[(((int64_t)r * N + n) * c_p) * 2 + i];
        ch[i] = v;
    }
    __syncthreads();
    const int cpg = c / groups;
    for (int i = threadIdx.x; i < c_p; i += blockDim.x) {
        float mean = 0.f, rstd = 0.f;
        if (i < c) {
            const int g = i / cpg;
            double s = 0.0, s2 = 0.0;
            for (int k = 0; k < cpg; ++k) { s += ch[(g * cpg + k) * 2]; s2 += ch[(g * cpg + k) * 2 + 1]; }
            const double m = (double)cpg * (double)spatial;
            const double mu = s / m;
            double var = s2 / m - mu * mu;          // biased variance, as torch's instance/group norm
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            rstd = (float)(1.0 / sqrt(var + (double)eps));
        }
        mean_rstd[((int64_t)n * c_p + i) * 2 + 0] = mean;
        mean_rstd[((int64_t)n * c_p + i) * 2 + 1] = rstd;
        if (scale_shift) {                        // the same two expressions k_norm_apply evaluates per thread
            float a = 0.f, b = 0.f;
            if (i < c) { a = rstd * gamma[i]; b = beta[i] - mean * a; }
            scale_shift[((int64_t)n * c_p + i) * 2 + 0] = a;
            scale_shift[((int64_t)n * c_p + i) * 2 + 1] = b;
        }
    }
}

// Same result from a grid (N, C_p / 32): one block per image and 32-channel block, the 32 replicas of its 64 sums read by 256
// threads in parallel (8 independent loads each) instead of 32 dependent loads per thread -- the 4-block version above took
// 6-16 us per call, 14 calls per training step on the critical path of the forward pass. Needs whole groups inside a 32-channel
// block (channels per group divides 32: InstanceNorm 1, the heads' GroupNorm 16). The fp64 replica sums are associated differently
// (4 partial sums of 8 replicas): 1e-16 relative, invisible after the rounding to fp32.
__global__ __launch_bounds__(256) void k_norm_finalize32(const double* __restrict__ stats, int N, int c, int c_p, int groups, int64_t spatial,
                                                         float eps, float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ scale_shift, const NormItems IT) {
    __shared__ double part[4][64];
    __shared__ double ch[64];
    const int n = blockIdx.x, c0 = blockIdx.y * 32, t = threadIdx.x;
    if (IT.n) spatial = IT.spatial[n];
    const int v = t & 63, rg = t >> 6;
    double ld[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ld[k] = stats[(((int64_t)(rg * 8 + k) * N + n) * c_p + c0) * 2 + v];
    double s8 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s8 += ld[k];                 // replicas rg*8 .. rg*8+7 in order
    part[rg][v] = s8;
    __syncthreads();
    if (t < 64) ch[t] = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
    __syncthreads();
    if (t < 32) {
        const int i = c0 + t, cpg = c / groups;
        float mean = 0.f, rstd = 0.f;
        if (i < c) {
            const int g0 = (t / cpg) * cpg;                  // first channel of this group inside the block
            double s = 0.0, s2 = 0.0;
            for (int k = 0; k < cpg; ++k) { s += ch[(g0 + k) * 2]; s2 += ch[(g0 + k) * 2 + 1]; }
            const double m = (double)cpg * (double)spatial;
            const double mu = s / m;
            double var = s2 / m - mu * mu;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            rstd = (float)(1.0 / sqrt(var + (double)eps));
        }
        mean_rstd[((int64_t)n * c_p + i) * 2 + 0] = mean;
        mean_rstd[((int64_t)n * c_p + i) * 2 + 1] = rstd;
        if (scale_shift) {
            float a = 0.f, b = 0.f;
            if (i < c) { a = rstd * gamma[i]; b = beta[i] - mean * a; }
            scale_shift[((int64_t)n * c_p + i) * 2 + 0] = a;
            scale_shift[((int64_t)n * c_p + i) * 2 + 1] = b;
        }
    }
}

static int finalize_launch(const double* stats, int N, int c, int c_p, int groups, int64_t spatial, float eps, float* mean_rstd,
                           const float* gamma, const float* beta, float* scale_shift, const NormItems& it, hipStream_t st) {
    const int cpg = c / groups;
    static const int fast = getenv("NNDET_NORM_FINALIZE32") ? atoi(getenv("NNDET_NORM_FINALIZE32")) : 1;
    if (fast && cpg > 0 && 32 % cpg == 0)
        k_norm_finalize32<<<dim3(N, c_p / 32), 256, 0, st>>>(stats, N, c, c_p, groups, spatial, eps, mean_rstd, gamma, beta, scale_shift, it);
    else
        k_norm_finalize<<<N, 256, (size_t)c_p * 16, st>>>(stats, N, c, c_p, groups, spatial, eps, mean_rstd, gamma, beta, scale_shift, it);
    LAUNCH_CHECK();
    return 0;
}

int norm_finalize_run(const double* stats, int N, int c, int c_p, int groups, int64_t spatial, float eps, float* mean_rstd, hipStream_t st) {
    return finalize_launch(stats, N, c, c_p, groups, spatial, eps, mean_rstd, nullptr, nullptr, nullptr, g_norm_uniform, st);
}

extern "C" int nndet_norm_finalize(const double* stats, const float* gamma, const float* beta, int32_t batch, int64_t spatial,
                                   int32_t c, int32_t c_p, int32_t groups, float eps, float* mean_rstd_out, float* scale_shift_out,
                                   void* stream) {
    if (!stats || !gamma || !beta || !mean_rstd_out || !scale_shift_out) return NNDET_EINVAL;
    if (c_p % 32 || c_p > 1024 || c <= 0 || c > c_p || groups <= 0 || c % groups) return NNDET_EINVAL;
    return finalize_launch(stats, batch, c, c_p, groups, spatial, eps, mean_rstd_out, gamma, beta, scale_shift_out, g_norm_uniform,
                           as_stream(stream));
}

// ------------------------------------------------------------------ apply: y = relu?((x - mean) * rstd * gamma + beta)
// Split I/O (round 5, the fused first layer of the two detection-head trunks: one 128 -> 2 x 128 convolution + one GroupNorm over 256
// channels): the channels [0, split) of the normalised output go to `y` and [split, c_p) to `y1`, each a dense [rows][own channel count]
// tensor -- what the two branches' next convolutions read -- and the backward kernels read the incoming gradient from two such tensors.
// y1 / dy1 == NULL: one tensor of c_p channels, as before.
template <typename T>
__global__ __launch_bounds__(256) void k_norm_apply(const T* __restrict__ x, const float* __restrict__ mean_rstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    int64_t spatial, int c, int c_p, int relu, T* __restrict__ y, int RPB,
                                                    const NormItems IT, T* __restrict__ y1 = nullptr, int split = 0) {
    constexpr int E = Vec16<T>::E;
    const int ppr = c_p / E, rpi = 256 / ppr;
    const int n = blockIdx.y;
    const int cp = threadIdx.x % ppr, rr = threadIdx.x / ppr;
    if (rr >= rpi) return;
    int64_t row0 = (int64_t)n * spatial;
    if (IT.n) { spatial = IT.spatial[n]; row0 = IT.row_off[n]; }
    const int64_t r0 = (int64_t)blockIdx.x * RPB;
    if (r0 >= spatial) return;                        // ragged batch: the grid covers the largest item
    float sc[E], sh[E];
    {
        float mu[E], rs[E], ga[E], be[E];
        norm_consts<E>(mean_rstd, gamma, beta, n, c, c_p, cp * E, mu, rs, ga, be);
#pragma unroll
        for (int e = 0; e < E; ++e) {            // (padded channels: rstd = gamma = beta = 0 -> a = b = 0, as before)
            const float a = rs[e] * ga[e];
            sc[e] = a; sh[e] = be[e] - mu[e] * a;
        }
    }
    const int64_t r1 = min(r0 + RPB, spatial);
    const int64_t base = row0 * c_p + cp * E;
    const bool hi = y1 != nullptr && cp * E >= split;
    T* const yo = hi ? y1 : y;
    const int yp = y1 ? (hi ? c_p - split : split) : c_p;                 // row pitch / first channel of this thread's output tensor
    const int64_t ybase = row0 * yp + (hi ? cp * E - split : cp * E);
    for (int64_t r = r0 + rr; r < r1; r += rpi)      // AffinePiece: the SAME code the convolutions run when they apply the norm on load
        *reinterpret_cast<u32x4*>(yo + ybase + r * yp) = AffinePiece<T>::apply(*reinterpret_cast<const u32x4*>(x + base + r * c_p), sc, sh, relu);
}

extern "C" int nndet_norm_apply(int32_t dtype, const void* x, const double* stats, const float* gamma, const float* beta,
                                int32_t batch, int64_t spatial, int32_t c, int32_t c_p, int32_t groups, float eps,
                                int32_t relu, void* y, float* mean_rstd_out, void* stream) {
    if (!x || !stats || !gamma || !beta || !y || !mean_rstd_out) return NNDET_EINVAL;
    if (c_p % 32 || c_p > 1024 || c <= 0 || c > c_p || groups <= 0 || c % groups) return NNDET_EINVAL;
    hipStream_t st = as_stream(stream);
    { const int frc = finalize_launch(stats, batch, c, c_p, groups, spatial, eps, mean_rstd_out, nullptr, nullptr, nullptr, g_norm_uniform, st);
      if (frc) return frc; }
    const int rpb = apply_rows(spatial * batch, c_p, nndet_esize(dtype));
    dim3 grid((unsigned)ceil_div64(spatial, rpb), batch);
    if (dtype == NNDET_BF16)
        k_norm_apply<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, mean_rstd_out, gamma, beta, spatial, c, c_p, relu, (bf16_t*)y, rpb, g_norm_uniform);
    else if (dtype == NNDET_F16)
        k_norm_apply<f16_t><<<grid, 256, 0, st>>>((const f16_t*)x, mean_rstd_out, gamma, beta, spatial, c, c_p, relu, (f16_t*)y, rpb, g_norm_uniform);
    else
        k_norm_apply<float><<<grid, 256, 0, st>>>((const float*)x, mean_rstd_out, gamma, beta, spatial, c, c_p, relu, (float*)y, rpb, g_norm_uniform);
    LAUNCH_CHECK();
    return 0;
}

// per-item voxel counts / first rows of a ragged batch + the totals the launch geometry needs
static int norm_items(const NndetItems* it, NormItems* ni, int64_t* total_rows, int64_t* max_spatial) {
    if (!it || it->n_items < 1 || it->n_items > NNDET_MAX_ITEMS) return NNDET_EINVAL;
    memset(ni, 0, sizeof(*ni));
    ni->n = it->n_items;
    *total_rows = 0; *max_spatial = 0;
    for (int i = 0; i < it->n_items; ++i) {
        const int64_t sp = (int64_t)it->dims[i][0] * it->dims[i][1] * it->dims[i][2];
        if (it->dims[i][0] <= 0 || it->dims[i][1] <= 0 || it->dims[i][2] <= 0 || sp >= (1LL << 31) || it->row_off[i] < 0) return NNDET_EINVAL;
        ni->spatial[i] = (int32_t)sp;
        ni->row_off[i] = it->row_off[i];
        *total_rows += sp;
        if (sp > *max_spatial) *max_spatial = sp;
    }
    return 0;
}

static int norm_apply_items_impl(int32_t dtype, const void* x, const double* stats, const float* gamma, const float* beta,
                                 const NndetItems* items, int32_t c, int32_t c_p, int32_t groups, float eps, int32_t relu,
                                 void* y, void* y1, int32_t split, float* mean_rstd_out, void* stream) {
    if (!x || !stats || !gamma || !beta || !y || !mean_rstd_out) return NNDET_EINVAL;
    if (c_p % 32 || c_p > 1024 || c <= 0 || c > c_p || groups <= 0 || c % groups) return NNDET_EINVAL;
    if (y1 && (split <= 0 || split >= c_p || split % 32)) return NNDET_EINVAL;
    NormItems ni;
    int64_t total = 0, mx = 0;
    const int rc = norm_items(items, &ni, &total, &mx);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    { const int frc = finalize_launch(stats, ni.n, c, c_p, groups, 0, eps, mean_rstd_out, nullptr, nullptr, nullptr, ni, st);
      if (frc) return frc; }
    const int rpb = apply_rows(total, c_p, nndet_esize(dtype));
    dim3 grid((unsigned)ceil_div64(mx, rpb), ni.n);
    if (dtype == NNDET_BF16)
        k_norm_apply<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, mean_rstd_out, gamma, beta, 0, c, c_p, relu, (bf16_t*)y, rpb, ni, (bf16_t*)y1, split);
    else if (dtype == NNDET_F16)
        k_norm_apply<f16_t><<<grid, 256, 0, st>>>((const f16_t*)x, mean_rstd_out, gamma, beta, 0, c, c_p, relu, (f16_t*)y, rpb, ni, (f16_t*)y1, split);
    else
        k_norm_apply<float><<<grid, 256, 0, st>>>((const float*)x, mean_rstd_out, gamma, beta, 0, c, c_p, relu, (float*)y, rpb, ni, (float*)y1, split);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_norm_apply_items(int32_t dtype, const void* x, const double* stats, const float* gamma, const float* beta,
                                      const NndetItems* items, int32_t c, int32_t c_p, int32_t groups, float eps, int32_t relu,
                                      void* y, float* mean_rstd_out, void* stream) {
    return norm_apply_items_impl(dtype, x, stats, gamma, beta, items, c, c_p, groups, eps, relu, y, nullptr, 0, mean_rstd_out, stream);
}

extern "C" int nndet_norm_apply_items_split(int32_t dtype, const void* x, const double* stats, const float* gamma, const float* beta,
                                            const NndetItems* items, int32_t c, int32_t c_p, int32_t groups, float eps, int32_t relu,
                                            void* y_lo, void* y_hi, int32_t split, float* mean_rstd_out, void* stream) {
    if (!y_hi) return NNDET_EINVAL;
    return norm_apply_items_impl(dtype, x, stats, gamma, beta, items, c, c_p, groups, eps, relu, y_lo, y_hi, split, mean_rstd_out, stream);
}

// ------------------------------------------------------------------ y = relu?(x * scale + shift) from a coefficient table
// (materialises a deferred activation for a consumer that cannot apply the norm on load; same arithmetic as k_norm_apply)
template <typename T>
__global__ __launch_bounds__(256) void k_affine_apply(const T* __restrict__ x, const float* __restrict__ ss, int64_t spatial, int c_p,
                                                      int relu, T* __restrict__ y, int RPB) {
    constexpr int E = Vec16<T>::E;
    const int ppr = c_p / E, rpi = 256 / ppr;
    const int n = blockIdx.y;
    const int cp = threadIdx.x % ppr, rr = threadIdx.x / ppr;
    if (rr >= rpi) return;
    float sc[E], sh[E];
    load_affine<E>(ss, n, c_p, cp * E, sc, sh);
    const int64_t r0 = (int64_t)blockIdx.x * RPB;
    const int64_t r1 = min(r0 + RPB, spatial);
    const int64_t base = ((int64_t)n * spatial) * c_p + cp * E;
    for (int64_t r = r0 + rr; r < r1; r += rpi)      // AffinePiece: the SAME code the convolutions run when they apply the norm on load
        *reinterpret_cast<u32x4*>(y + base + r * c_p) = AffinePiece<T>::apply(*reinterpret_cast<const u32x4*>(x + base + r * c_p), sc, sh, relu);
}

extern "C" int nndet_affine_apply(int32_t dtype, const void* x, const float* scale_shift, int32_t batch, int64_t spatial, int32_t c_p,
                                  int32_t relu, void* y, void* stream) {
    if (!x || !scale_shift || !y || c_p % 32 || c_p > 1024 || batch <= 0) return NNDET_EINVAL;
    const int rpb = apply_rows(spatial * batch, c_p, nndet_esize(dtype));
    dim3 grid((unsigned)ceil_div64(spatial, rpb), batch);
    if (dtype == NNDET_BF16)
        k_affine_apply<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((const bf16_t*)x, scale_shift, spatial, c_p, relu, (bf16_t*)y, rpb);
    else if (dtype == NNDET_F16)
        k_affine_apply<f16_t><<<grid, 256, 0, as_stream(stream)>>>((const f16_t*)x, scale_shift, spatial, c_p, relu, (f16_t*)y, rpb);
    else
        k_affine_apply<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)x, scale_shift, spatial, c_p, relu, (float*)y, rpb);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ backward
// pass 1: per (n, channel): A = sum g, B = sum g * xhat, with g = dy * [relu mask]
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void k_norm_bwd_reduce(const T* __restrict__ x, const T* __restrict__ dy,
                                                         const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int64_t spatial, int c, int c_p,
                                                         int N, double* __restrict__ red_ws, int RED_ROWS, int groups,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, const NormItems IT,
                                                         const T* __restrict__ dy1 = nullptr, int split = 0) {
    // RELU is a compile-time parameter since round 5: with the run-time flag, the branch-free prologue below AND the unrolled row loop,
    // hipcc (ROCm 7.2) produced a kernel whose dgamma sums were 1-7 % off -- only inside the training step, next to the weight-gradient
    // kernels of the other stream, never alone (each change alone was fine; tools/r5_race_probe2.py, profiles/round5_norm_reduce_miscompile.txt).
    // tests/test_model_gpu.py::test_norm_backward_inside_the_step_matches_torch_on_the_same_inputs guards the in-step result.
    constexpr int relu = RELU ? 1 : 0;
    constexpr int E = Vec16<T>::E;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem);
    const int ppr = c_p / E, rpi = 256 / ppr;
    const int n = blockIdx.y;
    int64_t row0 = (int64_t)n * spatial;
    unsigned int nblk = gridDim.x;                    // workgroups that work on (and draw a ticket for) image / item n
    if (IT.n) {
        spatial = IT.spatial[n]; row0 = IT.row_off[n];
        nblk = (unsigned int)((spatial + RED_ROWS - 1) / RED_ROWS);
        if (blockIdx.x >= nblk) return;               // ragged batch: the grid covers the largest item
    }
    for (int i = threadIdx.x; i < c_p * 2; i += 256) red[i] = 0.0;
    __syncthreads();
    const int cp = threadIdx.x % ppr, rr = threadIdx.x / ppr;
    if (rr < rpi) {
        float mu[E], rs[E], sc[E], sh[E], sa[E], sb[E];
        {
            float ga[E], be[E];
            norm_consts<E>(mean_rstd, gamma, beta, n, c, c_p, cp * E, mu, rs, ga, be);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                sc[e] = rs[e] * ga[e];
                sh[e] = be[e] - mu[e] * sc[e];
                sa[e] = 0.f; sb[e] = 0.f;
            }
        }
        const int64_t r0 = (int64_t)blockIdx.x * RED_ROWS;
        const int64_t r1 = min(r0 + RED_ROWS, spatial);
        const int64_t base = row0 * c_p + cp * E;
        const bool hi = dy1 != nullptr && cp * E >= split;              // split gradient input (see k_norm_apply)
        const T* const dyi = hi ? dy1 : dy;
        const int gp = dy1 ? (hi ? c_p - split : split) : c_p;
        const int64_t gbase = row0 * gp + (hi ? cp * E - split : cp * E);
        auto row = [&](const float* xv, const float* gv) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float xh = (xv[e] - mu[e]) * rs[e];
                float g = gv[e];
                if (relu && !(fmaf(xv[e], sc[e], sh[e]) > 0.f)) g = 0.f;   // same expression as the forward pass
                sa[e] += g; sb[e] += g * xh;
            }
        };
        // Round 5: RED_U rows (2 * RED_U 16-byte loads) of this thread in flight per iteration. The plain loop had ONE load pair per
        // thread outstanding, i.e. one memory round trip per RED_ROWS / rpi iterations: 128 round trips for the 64-channel layers
        // (0.18-0.29 ms for 315 MB = 1.1-1.7 TB/s inside the step), 0.4-0.9 TB/s for the 128-channel layers and the ragged head
        // batches (profiles/round4_v3_timeline_one_step.txt). Rows are still added in the order r, r + rpi, ...: same sums.
        int64_t r = r0 + rr;
        for (; r + (int64_t)(RED_U - 1) * rpi < r1; r += (int64_t)RED_U * rpi) {
            typename Vec16<T>::raw_t xr[RED_U], gr[RED_U];
#pragma unroll
            for (int u = 0; u < RED_U; ++u) {
                xr[u] = Vec16<T>::ldraw(x + base + (r + (int64_t)u * rpi) * c_p);
                gr[u] = Vec16<T>::ldraw(dyi + gbase + (r + (int64_t)u * rpi) * gp);
            }
#pragma unroll
            for (int u = 0; u < RED_U; ++u) {
                float xv[E], gv[E];
                Vec16<T>::cvt(xr[u], xv);
                Vec16<T>::cvt(gr[u], gv);
                row(xv, gv);
            }
        }
        for (; r < r1; r += rpi) {
            float xv[E], gv[E];
            Vec16<T>::ld(x + base + r * c_p, xv);
            Vec16<T>::ld(dyi + gbase + r * gp, gv);
            row(xv, gv);
        }
        norm_block_sums_to_lds<E>(red, ppr, cp, sa, sb);          // (ppr a power of two: rpi * ppr == 256, every lane is here)
    }
    __syncthreads();
    const int rep = blockIdx.x % NNDET_STATS_REPLICAS;
    double* dst = red_ws + ((int64_t)rep * N + n) * c_p * 2;
    for (int i = threadIdx.x; i < c_p * 2; i += 256) atomicAdd(&dst[i], red[i]);
    // ---- the LAST workgroup of image n finishes the reduction (was a separate 4-workgroup launch, k_norm_bwd_finalize, between the
    // two passes: 84 launches per training step). Ticket counters live behind the replicas in red_ws (zeroed with them). Ordering:
    // everything this workgroup publishes are agent-scope atomic RMWs, performed at the coherence point once vmcnt reaches 0; only
    // then is the ticket drawn. No release fence: it would write back the whole L2 (buffer_wbl2) once per workgroup -- measured
    // +1.2 ms per training step -- and there are no plain stores to publish. The last arriver acquires and reads the replica sums
    // with agent-scope atomic loads.
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned int* ticket = reinterpret_cast<unsigned int*>(red_ws + (int64_t)NNDET_STATS_REPLICAS * N * c_p * 2);
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&ticket[n], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double* ch = red;                               // [c_p][2]
    for (int i = threadIdx.x; i < c_p * 2; i += 256) {
        double v = 0.0;
        const int nrep = nblk < (unsigned)NNDET_STATS_REPLICAS ? (int)nblk : NNDET_STATS_REPLICAS;     // replica = workgroup index % 32: only these were written
        // Round 6: eight replica loads in flight per thread. The plain loop `v += load(r)` was a chain of up to 32 DEPENDENT L2 round trips at
        // the very end of every launch -- the last workgroup of an image, alone on the chip: 18 us for 3 replicas, 45 us for 10, 50+ us for 32
        // (profiles/round6_timeline_one_step.txt: the reductions of the 10^3 / 5^3-voxel layers took 4 x their apply passes). Same order of
        // additions: bit-identical sums.
        const double* src = red_ws + ((int64_t)n * c_p) * 2 + i;
        const int64_t rstride = (int64_t)N * c_p * 2;
        int r = 0;
        for (; r + 8 <= nrep; r += 8) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = __hip_atomic_load(src + (int64_t)(r + u) * rstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = r + u < nrep ? __hip_atomic_load(src + (int64_t)(r + u) * rstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (r + u < nrep) v += t[u];
        }
        ch[i] = v;
    }
    __syncthreads();
    const int cpg = c / groups;
    float* coef = reinterpret_cast<float*>(red_ws + ((int64_t)n * c_p) * 2);   // replica 0, this n: [c_p][2] fp32 (first half)
    for (int i = threadIdx.x; i < c_p; i += 256) {
        float c1 = 0.f, c2 = 0.f;
        if (i < c) {
            atomicAdd(&dbeta[i], (float)ch[i * 2]);
            atomicAdd(&dgamma[i], (float)ch[i * 2 + 1]);
            const int g = i / cpg;
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < cpg; ++k) {
                const double gm = (double)gamma[g * cpg + k];
                s1 += gm * ch[(g * cpg + k) * 2];
                s2 += gm * ch[(g * cpg + k) * 2 + 1];
            }
            const double m = (double)cpg * (double)spatial;
            c1 = (float)(s1 / m); c2 = (float)(s2 / m);
        }
        // every read of this n's replica slices happened before the __syncthreads above; the coefficients are read by the NEXT kernel
        coef[i * 2 + 0] = c1;
        coef[i * 2 + 1] = c2;
    }
}

// (pass 2, dgamma / dbeta and the per-channel group coefficients (s1/m, s2/m) written over replica 0, is the tail of pass 1)

// pass 3: dx = rstd * (g * gamma - s1/m - xhat * s2/m)
template <typename T>
__global__ __launch_bounds__(256) void k_norm_bwd_apply(const T* __restrict__ x, const T* __restrict__ dy,
                                                        const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const double* __restrict__ red_ws,
                                                        int64_t spatial, int c, int c_p, int relu, T* __restrict__ dx, int RPB,
                                                        const NormItems IT, const T* __restrict__ dy1 = nullptr, int split = 0) {
    constexpr int E = Vec16<T>::E;
    const int ppr = c_p / E, rpi = 256 / ppr;
    const int n = blockIdx.y;
    const int cp = threadIdx.x % ppr, rr = threadIdx.x / ppr;
    if (rr >= rpi) return;
    int64_t row0 = (int64_t)n * spatial;
    if (IT.n) { spatial = IT.spatial[n]; row0 = IT.row_off[n]; }
    const int64_t r0 = (int64_t)blockIdx.x * RPB;
    if (r0 >= spatial) return;                        // ragged batch: the grid covers the largest item
    const float* coef = reinterpret_cast<const float*>(red_ws + ((int64_t)n * c_p) * 2);
    float mu[E], rs[E], ga[E], sc[E], sh[E], k1[E], k2[E];
    {
        float be[E];
        float2 kk[E];
#pragma unroll
        for (int e = 0; e < E; ++e) kk[e] = reinterpret_cast<const float2*>(coef)[cp * E + e];   // [c_p][2], zeros beyond c (k_norm_bwd_reduce)
        norm_consts<E>(mean_rstd, gamma, beta, n, c, c_p, cp * E, mu, rs, ga, be);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            sc[e] = rs[e] * ga[e];
            sh[e] = be[e] - mu[e] * sc[e];
            k1[e] = kk[e].x; k2[e] = kk[e].y;
        }
    }
    const int64_t r1 = min(r0 + RPB, spatial);
    const int64_t base = row0 * c_p + cp * E;
    const bool hi = dy1 != nullptr && cp * E >= split;
    const T* const dyi = hi ? dy1 : dy;
    const int gp = dy1 ? (hi ? c_p - split : split) : c_p;
    const int64_t gbase = row0 * gp + (hi ? cp * E - split : cp * E);
    for (int64_t r = r0 + rr; r < r1; r += rpi) {
        float xv[E], gv[E];
        Vec16<T>::ld(x + base + r * c_p, xv);
        Vec16<T>::ld(dyi + gbase + r * gp, gv);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float xh = (xv[e] - mu[e]) * rs[e];
            float g = gv[e];
            if (relu && !(fmaf(xv[e], sc[e], sh[e]) > 0.f)) g = 0.f;
            xv[e] = rs[e] * (g * ga[e] - k1[e] - xh * k2[e]);
        }
        Vec16<T>::st(dx + base + r * c_p, xv);
    }
}

extern "C" int nndet_norm_backward(int32_t dtype, const void* x, const void* dy, const float* mean_rstd, const float* gamma,
                                   const float* beta, int32_t batch, int64_t spatial, int32_t c, int32_t c_p, int32_t groups,
                                   int32_t relu, void* dx, float* dgamma, float* dbeta, double* red_ws, void* stream) {
    if (!x || !dy || !mean_rstd || !gamma || !beta || !dx || !dgamma || !dbeta || !red_ws) return NNDET_EINVAL;
    if (c_p % 32 || c_p > 1024 || c <= 0 || c > c_p || groups <= 0 || c % groups) return NNDET_EINVAL;
    hipStream_t st = as_stream(stream);
    const int rpb = apply_rows(spatial * batch, c_p, nndet_esize(dtype));
    dim3 grid((unsigned)ceil_div64(spatial, rpb), batch);
    const int rr = red_rows(spatial * batch);
    dim3 rgrid((unsigned)ceil_div64(spatial, rr), batch);
    const size_t lds = (size_t)c_p * 16;
    static const int dbg_skip = nndet_timing_experiment("NNDET_NORM_DBG_SKIP_REDUCE");
    if (dbg_skip) {}
    else if (dtype == NNDET_BF16)
        (relu ? k_norm_bwd_reduce<bf16_t, true> : k_norm_bwd_reduce<bf16_t, false>)<<<rgrid, 256, lds, st>>>((const bf16_t*)x, (const bf16_t*)dy, mean_rstd, gamma, beta, spatial, c, c_p, batch, red_ws, rr, groups, dgamma, dbeta, g_norm_uniform, (const bf16_t*)nullptr, 0);
    else if (dtype == NNDET_F16)
        (relu ? k_norm_bwd_reduce<f16_t, true> : k_norm_bwd_reduce<f16_t, false>)<<<rgrid, 256, lds, st>>>((const f16_t*)x, (const f16_t*)dy, mean_rstd, gamma, beta, spatial, c, c_p, batch, red_ws, rr, groups, dgamma, dbeta, g_norm_uniform, (const f16_t*)nullptr, 0);
    else
        (relu ? k_norm_bwd_reduce<float, true> : k_norm_bwd_reduce<float, false>)<<<rgrid, 256, lds, st>>>((const float*)x, (const float*)dy, mean_rstd, gamma, beta, spatial, c, c_p, batch, red_ws, rr, groups, dgamma, dbeta, g_norm_uniform, (const float*)nullptr, 0);
    LAUNCH_CHECK();
    if (dtype == NNDET_BF16)
        k_norm_bwd_apply<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy, mean_rstd, gamma, beta, red_ws, spatial, c, c_p, relu, (bf16_t*)dx, rpb, g_norm_uniform);
    else if (dtype == NNDET_F16)
        k_norm_bwd_apply<f16_t><<<grid, 256, 0, st>>>((const f16_t*)x, (const f16_t*)dy, mean_rstd, gamma, beta, red_ws, spatial, c, c_p, relu, (f16_t*)dx, rpb, g_norm_uniform);
    else
        k_norm_bwd_apply<float><<<grid, 256, 0, st>>>((const float*)x, (const float*)dy, mean_rstd, gamma, beta, red_ws, spatial, c, c_p, relu, (float*)dx, rpb, g_norm_uniform);
    LAUNCH_CHECK();
    return 0;
}

// ---- the sums arrive from somewhere else: the data-gradient kernel that produced `dy` accumulated S1 / S2 per (image, channel) into
// red_ws's replicas in its epilogue (conv_dgs.hip, k_dgs<..., NB = true>). What is left of pass 1 is the tail of k_norm_bwd_reduce's last
// workgroup -- replicas -> dgamma / dbeta and the group coefficients over replica 0 -- as its own small launch, then pass 3 unchanged.
__global__ __launch_bounds__(256) void k_norm_bwd_finalize(double* __restrict__ red_ws, int N, int c, int c_p, int groups, int64_t spatial,
                                                           const float* __restrict__ gamma, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ch = reinterpret_cast<double*>(smem);    // [c_p][2]
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < c_p * 2; i += 256) {
        double v = 0.0;
        static_assert(NNDET_STATS_REPLICAS % 8 == 0, "replica loads come in batches of 8");
        for (int r0 = 0; r0 < NNDET_STATS_REPLICAS; r0 += 8) {       // eight loads in flight, added in replica order (as k_norm_bwd_reduce's tail)
            double l8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) l8[k] = red_ws[(((int64_t)(r0 + k) * N + n) * c_p) * 2 + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) v += l8[k];
        }
        ch[i] = v;
    }
    __syncthreads();
    const int cpg = c / groups;
    float* coef = reinterpret_cast<float*>(red_ws + ((int64_t)n * c_p) * 2);
    for (int i = threadIdx.x; i < c_p; i += 256) {
        float c1 = 0.f, c2 = 0.f;
        if (i < c) {
            atomicAdd(&dbeta[i], (float)ch[i * 2]);
            atomicAdd(&dgamma[i], (float)ch[i * 2 + 1]);
            const int g = i / cpg;
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < cpg; ++k) {
                const double gm = (double)gamma[g * cpg + k];
                s1 += gm * ch[(g * cpg + k) * 2];
                s2 += gm * ch[(g * cpg + k) * 2 + 1];
            }
            const double m = (double)cpg * (double)spatial;
            c1 = (float)(s1 / m); c2 = (float)(s2 / m);
        }
        coef[i * 2 + 0] = c1;
        coef[i * 2 + 1] = c2;
    }
}

extern "C" int nndet_norm_backward_presummed(int32_t dtype, const void* x, const void* dy, const float* mean_rstd, const float* gamma,
                                             const float* beta, int32_t batch, int64_t spatial, int32_t c, int32_t c_p, int32_t groups,
                                             int32_t relu, void* dx, float* dgamma, float* dbeta, double* red_ws, void* stream) {
    if (!x || !dy || !mean_rstd || !gamma || !beta || !dx || !dgamma || !dbeta || !red_ws) return NNDET_EINVAL;
    if (c_p % 32 || c_p > 1024 || c <= 0 || c > c_p || groups <= 0 || c % groups) return NNDET_EINVAL;
    hipStream_t st = as_stream(stream);
    const int rpb = apply_rows(spatial * batch, c_p, nndet_esize(dtype));
    dim3 grid((unsigned)ceil_div64(spatial, rpb), batch);
    k_norm_bwd_finalize<<<batch, 256, (size_t)c_p * 16, st>>>(red_ws, batch, c, c_p, groups, spatial, gamma, dgamma, dbeta);
    LAUNCH_CHECK();
    if (dtype == NNDET_BF16)
        k_norm_bwd_apply<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy, mean_rstd, gamma, beta, red_ws, spatial, c, c_p, relu, (bf16_t*)dx, rpb, g_norm_uniform);
    else if (dtype == NNDET_F16)
        k_norm_bwd_apply<f16_t><<<grid, 256, 0, st>>>((const f16_t*)x, (const f16_t*)dy, mean_rstd, gamma, beta, red_ws, spatial, c, c_p, relu, (f16_t*)dx, rpb, g_norm_uniform);
    else
        k_norm_bwd_apply<float><<<grid, 256, 0, st>>>((const float*)x, (const float*)dy, mean_rstd, gamma, beta, red_ws, spatial, c, c_p, relu, (float*)dx, rpb, g_norm_uniform);
    LAUNCH_CHECK();
    return 0;
}

static int norm_backward_items_impl(int32_t dtype, const void* x, const void* dy, const void* dy1, int32_t split, const float* mean_rstd,
                                    const float* gamma, const float* beta, const NndetItems* items, int32_t c, int32_t c_p, int32_t groups,
                                    int32_t relu, void* dx, float* dgamma, float* dbeta, double* red_ws, void* stream) {
    if (!x || !dy || !mean_rstd || !gamma || !beta || !dx || !dgamma || !dbeta || !red_ws) return NNDET_EINVAL;
    if (c_p % 32 || c_p > 1024 || c <= 0 || c > c_p || groups <= 0 || c % groups) return NNDET_EINVAL;
    if (dy1 && (split <= 0 || split >= c_p || split % 32)) return NNDET_EINVAL;
    NormItems ni;
    int64_t total = 0, mx = 0;
    const int rc = norm_items(items, &ni, &total, &mx);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    const int rpb = apply_rows(total, c_p, nndet_esize(dtype));
    dim3 grid((unsigned)ceil_div64(mx, rpb), ni.n);
    const int rr = red_rows(total);
    dim3 rgrid((unsigned)ceil_div64(mx, rr), ni.n);
    const size_t lds = (size_t)c_p * 16;
    static const int dbg_skip = nndet_timing_experiment("NNDET_NORM_DBG_SKIP_REDUCE");
    if (dbg_skip) {}
    else if (dtype == NNDET_BF16)
        (relu ? k_norm_bwd_reduce<bf16_t, true> : k_norm_bwd_reduce<bf16_t, false>)<<<rgrid, 256, lds, st>>>((const bf16_t*)x, (const bf16_t*)dy, mean_rstd, gamma, beta, 0, c, c_p, ni.n, red_ws, rr, groups, dgamma, dbeta, ni, (const bf16_t*)dy1, split);
    else if (dtype == NNDET_F16)
        (relu ? k_norm_bwd_reduce<f16_t, true> : k_norm_bwd_reduce<f16_t, false>)<<<rgrid, 256, lds, st>>>((const f16_t*)x, (const f16_t*)dy, mean_rstd, gamma, beta, 0, c, c_p, ni.n, red_ws, rr, groups, dgamma, dbeta, ni, (const f16_t*)dy1, split);
    else
        (relu ? k_norm_bwd_reduce<float, true> : k_norm_bwd_reduce<float, false>)<<<rgrid, 256, lds, st>>>((const float*)x, (const float*)dy, mean_rstd, gamma, beta, 0, c, c_p, ni.n, red_ws, rr, groups, dgamma, dbeta, ni, (const float*)dy1, split);
    LAUNCH_CHECK();
    if (dtype == NNDET_BF16)
        k_norm_bwd_apply<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy, mean_rstd, gamma, beta, red_ws, 0, c, c_p, relu, (bf16_t*)dx, rpb, ni, (const bf16_t*)dy1, split);
    else if (dtype == NNDET_F16)
        k_norm_bwd_apply<f16_t><<<grid, 256, 0, st>>>((const f16_t*)x, (const f16_t*)dy, mean_rstd, gamma, beta, red_ws, 0, c, c_p, relu, (f16_t*)dx, rpb, ni, (const f16_t*)dy1, split);
    else
        k_norm_bwd_apply<float><<<grid, 256, 0, st>>>((const float*)x, (const float*)dy, mean_rstd, gamma, beta, red_ws, 0, c, c_p, relu, (float*)dx, rpb, ni, (const float*)dy1, split);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_norm_backward_items(int32_t dtype, const void* x, const void* dy, const float* mean_rstd, const float* gamma,
                                         const float* beta, const NndetItems* items, int32_t c, int32_t c_p, int32_t groups,
                                         int32_t relu, void* dx, float* dgamma, float* dbeta, double* red_ws, void* stream) {
    return norm_backward_items_impl(dtype, x, dy, nullptr, 0, mean_rstd, gamma, beta, items, c, c_p, groups, relu, dx, dgamma, dbeta, red_ws, stream);
}

extern "C" int nndet_norm_backward_items_split(int32_t dtype, const void* x, const void* dy_lo, const void* dy_hi, int32_t split,
                                               const float* mean_rstd, const float* gamma, const float* beta, const NndetItems* items,
                                               int32_t c, int32_t c_p, int32_t groups, int32_t relu, void* dx, float* dgamma, float* dbeta,
                                               double* red_ws, void* stream) {
    if (!dy_hi) return NNDET_EINVAL;
    return norm_backward_items_impl(dtype, x, dy_lo, dy_hi, split, mean_rstd, gamma, beta, items, c, c_p, groups, relu, dx, dgamma, dbeta, red_ws, stream);
}

// ------------------------------------------------------------------ column sums (bias gradient): out[c] += sum_rows x[row][c]
template <typename T>
__global__ __launch_bounds__(256) void k_colsum(const T* __restrict__ x, int64_t rows, int c_p, int c, float* __restrict__ out, int RED_ROWS) {
    constexpr int E = Vec16<T>::E;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    const int ppr = c_p / E, rpi = 256 / ppr;
    for (int i = threadIdx.x; i < c_p; i += 256) red[i] = 0.f;
    __syncthreads();
    const int cp = threadIdx.x % ppr, rr = threadIdx.x / ppr;
    if (rr < rpi) {
        float s[E];
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] = 0.f;
        const int64_t r0 = (int64_t)blockIdx.x * RED_ROWS;
        const int64_t r1 = min(r0 + RED_ROWS, rows);
        int64_t r = r0 + rr;
        for (; r + (int64_t)(RED_U - 1) * rpi < r1; r += (int64_t)RED_U * rpi) {
            typename Vec16<T>::raw_t raw[RED_U];
#pragma unroll
            for (int u = 0; u < RED_U; ++u) raw[u] = Vec16<T>::ldraw(x + (r + (int64_t)u * rpi) * c_p + cp * E);
#pragma unroll
            for (int u = 0; u < RED_U; ++u) {
                float v[E];
                Vec16<T>::cvt(raw[u], v);
#pragma unroll
                for (int e = 0; e < E; ++e) s[e] += v[e];
            }
        }
        for (; r < r1; r += rpi) {
            float v[E];
            Vec16<T>::ld(x + r * c_p + cp * E, v);
#pragma unroll
            for (int e = 0; e < E; ++e) s[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) atomicAdd(&red[cp * E + e], s[e]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += 256) atomicAdd(&out[i], red[i]);
}

int colsum_run(int dtype, const void* x, int64_t rows, int c_p, int c, float* out, hipStream_t st) {
    if (c_p % 32 || c_p > 1024 || rows <= 0) return NNDET_EINVAL;
    const int rr = red_rows(rows);
    const unsigned nb = (unsigned)ceil_div64(rows, rr);
    if (dtype == NNDET_BF16) k_colsum<bf16_t><<<nb, 256, (size_t)c_p * 4, st>>>((const bf16_t*)x, rows, c_p, c, out, rr);
    else if (dtype == NNDET_F16) k_colsum<f16_t><<<nb, 256, (size_t)c_p * 4, st>>>((const f16_t*)x, rows, c_p, c, out, rr);
    else k_colsum<float><<<nb, 256, (size_t)c_p * 4, st>>>((const float*)x, rows, c_p, c, out, rr);
    LAUNCH_CHECK();
    return 0;
}
