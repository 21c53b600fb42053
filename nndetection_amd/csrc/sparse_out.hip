// Sparse backward of the detection head's OUTPUT convolutions (training).
//
// The detection loss touches at most batch_size_per_image * batch sampled anchors (nndet/arch/heads/comb.py:351-405,
// nndet/core/boxes/sampler.py:237-270: <= 42 positives for the regression loss, <= 128 positives + negatives for the classification
// loss at batch 4), so the gradient w.r.t. the head outputs box_deltas [B * 1 186 650, 6] / box_logits [.., C] has <= 170 non-zero
// rows. The reference (and rounds 1-2 here) nevertheless run the dense backward of the two output convolutions over all pyramid
// levels: flatten / cat backward (114 + 19 MB), data gradient 128 <- 162 and 128 <- 27 (3x3x3, 197 + 39 GFLOP) and the weight
// gradients (the same again) -- 0.47 TFLOP and ~1.1 ms of kernel time per step for <= 170 x 27 x 128 x 6 useful products.
//
//   k_ho_scatter : (anchor index, gradient values) entries -> (row of the ragged head buffer, first channel, values x level Scale),
//                  writes them into the (zero-filled) DENSE gradient buffer as well, so that tensor stays a valid gradient for any
//                  consumer that does not know about the sparse form; d(Scale) of the regressor (nndet/arch/layers/scale.py:21-43).
//   k_ho_backward: one workgroup per (entry, tap), one thread per input channel, three passes over the SAME index set: zero the fp32
//                  scratch rows the entries touch, dX32[row + tap] += sum_g v_g W[c0+g][:, tap] (fp32 atomics) with dW[c0+g][:, tap] += v_g
//                  X[row + tap] and dbias[c0+g] += v_g, convert the touched rows into the (zero-filled) dense gradient.
// Replaces, for these two layers only, nndet_head_gather_backward + nndet_conv3d_backward_data_items +
// nndet_conv3d_backward_weight_items. Same mathematics; the values are not rounded to 16 bits on the way (closer to fp32).
#include "common.h"
#include "conv_common.h"

#define HO_MAX_LEVELS 8

struct HoLevels {
    int32_t nlev, A, G, N;
    int64_t anchors_per_image;                 // sum over levels of points * A
    int64_t anchor_off[HO_MAX_LEVELS + 1];     // first anchor of level l inside an image
    int64_t points[HO_MAX_LEVELS];
    int64_t row0[HO_MAX_LEVELS];               // first row of level l in the ragged buffer (level-major: row = row0 + n * points + pos)
    const float* scale[HO_MAX_LEVELS];         // per-level Scale (regressor) or NULL
    float* dscale[HO_MAX_LEVELS];
};

struct HoItems { int32_t n; int32_t dims[NNDET_MAX_ITEMS][3]; int64_t row_off[NNDET_MAX_ITEMS]; };

// grid ceil(K / 64), block 64: one thread per entry
template <typename T>
__global__ void k_ho_scatter(const int64_t* __restrict__ idx, const float* __restrict__ val, int K, const HoLevels Lv,
                             const T* __restrict__ y, int cout_p, T* __restrict__ dy, int32_t* __restrict__ rows,
                             int32_t* __restrict__ c0s, float* __restrict__ vals) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int64_t i = idx[k];
    if (i < 0 || i >= Lv.anchors_per_image * Lv.N) { rows[k] = -1; c0s[k] = 0; return; }
    const int n = (int)(i / Lv.anchors_per_image);
    const int64_t j = i - (int64_t)n * Lv.anchors_per_image;
    int l = 0;
    while (l + 1 < Lv.nlev && j >= Lv.anchor_off[l + 1]) ++l;
    const int64_t jl = j - Lv.anchor_off[l];
    const int64_t pos = jl / Lv.A;
    const int a = (int)(jl - pos * Lv.A);
    const int64_t row = Lv.row0[l] + (int64_t)n * Lv.points[l] + pos;
    const int c0 = a * Lv.G;
    const float sc = Lv.scale[l] ? *Lv.scale[l] : 1.f;
    float ds = 0.f;
    for (int g = 0; g < Lv.G; ++g) {
        const float v = val[(int64_t)k * Lv.G + g];
        if (Lv.scale[l]) ds += v * Elem<T>::ld(y[row * cout_p + c0 + g]);        // out = scale * y  =>  d(scale) += g * y
        const float vs = v * sc;
        vals[(int64_t)k * Lv.G + g] = vs;
        dy[row * cout_p + c0 + g] = Elem<T>::st(vs);     // (two sampled anchors never share (row, channel): distinct anchors)
    }
    if (Lv.dscale[l]) atomicAdd(Lv.dscale[l], ds);
    rows[k] = (int32_t)row;
    c0s[k] = c0;
}

// grid (K, 27), block 128 (thread = input channel, looped): one workgroup per (entry, tap) -- K x 27 ~ 1 000 ... 3 500 small workgroups
// instead of K long ones (the per-entry loop over 27 taps ran 55 - 75 us on 42 ... 127 workgroups, on the critical chain between the
// loss and the head trunks' backward pass). MODE 0: zero the fp32 scratch rows this (entry, tap) touches; MODE 1: accumulate;
// MODE 2: convert the touched rows to the activation type (rows touched by several entries are written several times with the same
// value). Only touched rows of the scratch are ever read, so it needs no dense fill and no dense conversion pass.
template <typename T, int MODE>
__global__ __launch_bounds__(128) void k_ho_backward(const int32_t* __restrict__ rows, const int32_t* __restrict__ c0s,
                                                     const float* __restrict__ vals, int G, const HoItems It, const T* __restrict__ x,
                                                     int cin, int cin_p, const float* __restrict__ w, int cout, float* __restrict__ dx32,
                                                     T* __restrict__ dx, float* __restrict__ dw, float* __restrict__ dbias) {
    const int k = blockIdx.x, t = blockIdx.y;
    const int row = rows[k];
    if (row < 0) return;                                                     // uniform
    int it = 0;
    while (it + 1 < It.n && row >= It.row_off[it + 1]) ++it;
    const int D = It.dims[it][0], H = It.dims[it][1], W = It.dims[it][2];
    const int pos = row - (int)It.row_off[it];
    const int pd = pos / (H * W), ph = (pos / W) % H, pw = pos % W;
    const int qd = pd + t / 9 - 1, qh = ph + (t / 3) % 3 - 1, qw = pw + t % 3 - 1;
    const bool inside = (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W;
    const int c0 = c0s[k];
    if (!inside) return;                                                     // zero padding (uniform)
    const int64_t qrow = It.row_off[it] + ((int64_t)qd * H + qh) * W + qw;
    if constexpr (MODE == 0) {
        for (int ci = threadIdx.x; ci < cin_p; ci += blockDim.x) dx32[qrow * cin_p + ci] = 0.f;
    } else if constexpr (MODE == 2) {
        for (int ci = threadIdx.x; ci < cin_p; ci += blockDim.x) dx[qrow * cin_p + ci] = Elem<T>::st(dx32[qrow * cin_p + ci]);
    } else {
        float v[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) v[g] = (g < G && c0 + g < cout) ? vals[(int64_t)k * G + g] : 0.f;
        for (int ci = threadIdx.x; ci < cin; ci += blockDim.x) {
            float acc = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g < G && c0 + g < cout) {
                    const int64_t wi = ((int64_t)(c0 + g) * cin + ci) * 27 + t;
                    acc = fmaf(v[g], w[wi], acc);                            // y[p][c] = sum W[c][ci][t] x[p + t - 1][ci]
                }
            }
            atomicAdd(dx32 + qrow * cin_p + ci, acc);
        }
    }
}

// Weight and bias gradient of the output convolution from the K entries, GATHERED (round 6): one workgroup per (output channel c, tap t),
// thread = input channel: dW[c][ci][t] += sum over the entries k that cover c (c0_k <= c < c0_k + G) of v_k[c - c0_k] * x[row_k + t - 1][ci],
// in entry order, ONE writer per element. The first form scattered from k_ho_backward<1> with 6-8 fp32 atomics per thread into dW (870 k
// atomics for the regressor's 42 entries): 47-50 us per output convolution on the chain at the start of the backward pass whatever the
// number of entries, and a summation order that changed from run to run. Here every workgroup looks at the K (row, c0) pairs once
// (flags in LDS), then touches only its ~K G / cout matching entries.
template <typename T, int NU>
__global__ __launch_bounds__(128) void k_ho_wgrad(const int32_t* __restrict__ rows, const int32_t* __restrict__ c0s, const float* __restrict__ vals,
                                                  int K, int G, const HoItems It, const T* __restrict__ x, int cin, int cin_p, int cout,
                                                  float* __restrict__ dw, float* __restrict__ dbias) {
    constexpr int EB = 16 / NU;            // entries per batch: 16 row loads in flight per thread
    __shared__ int64_t it_off[NNDET_MAX_ITEMS];    // the item table in LDS: the slots decode their entries in PARALLEL with per-lane item indices
    __shared__ int it_dim[NNDET_MAX_ITEMS][3];     // (a per-lane index into the by-value kernel argument would send the table to scratch; decoding the
                                                   //  covering entries one after the other cost 4 integer divisions each: 72 us when one anchor class
                                                   //  holds most of the sampled anchors)
    __shared__ int64_t q_s[256];           // slot i of the chunk: row of the tap's voxel of entry k0 + i, -2 = zero padding, -1 = does not cover c
    __shared__ float v_s[256];
    __shared__ int64_t cq_s[256];          // the covering entries compacted in entry order
    __shared__ float cv_s[256];
    __shared__ int cn_s;
    const int c = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    if (tid < It.n) { it_off[tid] = It.row_off[tid]; it_dim[tid][0] = It.dims[tid][0]; it_dim[tid][1] = It.dims[tid][1]; it_dim[tid][2] = It.dims[tid][2]; }
    const int nit = It.n;
    float acc[NU];                         // ci = tid + 128 u (cin <= 128 NU)
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = 0.f;
    float bsum = 0.f;
    for (int k0 = 0; k0 < K; k0 += 256) {
        __syncthreads();
        for (int i = tid; i < 256; i += 128) {
            const int k = k0 + i;
            int64_t q = -1;
            float v = 0.f;
            if (k < K) {
                const int row = rows[k], g = c - c0s[k];
                if (row >= 0 && g >= 0 && g < G) {
                    v = vals[(int64_t)k * G + g];
                    int it = 0;
                    while (it + 1 < nit && row >= it_off[it + 1]) ++it;
                    const int D = it_dim[it][0], H = it_dim[it][1], W = it_dim[it][2];
                    const int pos = row - (int)it_off[it];
                    const int pd = pos / (H * W), ph = (pos / W) % H, pw = pos % W;
                    const int qd = pd + t / 9 - 1, qh = ph + (t / 3) % 3 - 1, qw = pw + t % 3 - 1;
                    q = ((unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W)
                            ? it_off[it] + ((int64_t)qd * H + qh) * W + qw : -2;
                }
            }
            q_s[i] = q; v_s[i] = v;
        }
        __syncthreads();
        if (tid < 64) {                    // wave 0: ballot + prefix count per group of 64 slots -> entry order kept
            int base = 0;
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const int64_t q = q_s[grp * 64 + tid];
                const float v = v_s[grp * 64 + tid];
                const unsigned long long m = __ballot(q != -1);
                if (q != -1) { const int at = base + __popcll(m & ((1ULL << tid) - 1ULL)); cq_s[at] = q; cv_s[at] = v; }
                base += __popcll(m);
            }
            if (tid == 0) cn_s = base;
        }
        __syncthreads();
        const int n = cn_s;
        for (int i = 0; i < n; i += EB) {
            float v4[EB], x4[EB][NU];
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int64_t q = i + e < n ? cq_s[i + e] : -2;
                v4[e] = i + e < n ? cv_s[i + e] : 0.f;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    // UNCONDITIONAL load of a clamped address (a conditional load makes hipcc branch and wait per load: one round trip each);
                    // entries that do not count carry v = 0 or are zeroed here
                    const int ci = min(tid + 128 * u, cin - 1);
                    const float xv = Elem<T>::ld(x[(q >= 0 ? q : 0) * cin_p + ci]);
                    x4[e][u] = (q >= 0 && tid + 128 * u < cin) ? xv : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                bsum += v4[e];
#pragma unroll
                for (int u = 0; u < NU; ++u) acc[u] = fmaf(v4[e], x4[e][u], acc[u]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int ci = tid + 128 * u;
        if (ci < cin && acc[u] != 0.f) dw[((int64_t)c * cin + ci) * 27 + t] += acc[u];
    }
    if (t == 13 && tid == 0 && dbias && bsum != 0.f) dbias[c] += bsum;      // (the centre tap is always inside: every covering entry counts once)
}

// Forward of an output convolution at K sampled anchors only (training: the regression loss reads the box deltas of the <= 42 sampled
// positives and nothing else, nndet/arch/heads/comb.py:383-401): out[k][g] = scale_l * (sum_{t, ci} W[c0+g][ci][t] x[row + t - 1][ci] +
// b[c0+g]). One workgroup per entry, thread = input channel, LDS reduction. Also emits (row, c0) and the UNSCALED values (for d(Scale)).
template <typename T>
__global__ __launch_bounds__(512) void k_ho_forward(const int64_t* __restrict__ idx, int K, const HoLevels Lv, const HoItems It,
                                                    const T* __restrict__ x, int cin, int cin_p, const float* __restrict__ w,
                                                    const float* __restrict__ bias, int cout, float* __restrict__ out,
                                                    float* __restrict__ raw, int32_t* __restrict__ rows, int32_t* __restrict__ c0s, int32_t* __restrict__ lvls) {
    // Round 6: 512 threads = 4 tap groups x 128 input channels, the 7 taps of a group unrolled with every load in flight together (clamped
    // addresses, invalid taps multiply by zero). The first form walked the 27 taps one after the other with 7 dependent loads each: 27 L2
    // round trips = 55 us on the chain between the forward and the backward pass for 42 workgroups of work.
    __shared__ float red[8][8];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int64_t i = idx[k];
    const int G = Lv.G;
    if (i < 0 || i >= Lv.anchors_per_image * Lv.N) {                         // unused slot (uniform)
        if (tid < G) { out[(int64_t)k * G + tid] = 0.f; raw[(int64_t)k * G + tid] = 0.f; }
        if (tid == 0) { rows[k] = -1; c0s[k] = 0; lvls[k] = -1; }
        return;
    }
    const int n = (int)(i / Lv.anchors_per_image);
    const int64_t j = i - (int64_t)n * Lv.anchors_per_image;
    int l = 0;
    while (l + 1 < Lv.nlev && j >= Lv.anchor_off[l + 1]) ++l;
    const int64_t jl = j - Lv.anchor_off[l];
    const int64_t pos = jl / Lv.A;
    const int a = (int)(jl - pos * Lv.A);
    const int64_t row = Lv.row0[l] + (int64_t)n * Lv.points[l] + pos;
    const int c0 = a * G;
    int it = 0;
    while (it + 1 < It.n && row >= It.row_off[it + 1]) ++it;
    const int D = It.dims[it][0], H = It.dims[it][1], W = It.dims[it][2];
    const int p = (int)(row - It.row_off[it]);
    const int pd = p / (H * W), ph = (p / W) % H, pw = p % W;
    const int tg = tid >> 7;
    float acc[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) acc[g] = 0.f;
    for (int ci = tid & 127; ci < cin; ci += 128) {
        float xv[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int t = tg + 4 * u;
            const int tt = t < 27 ? t : 0;
            const int qd = pd + tt / 9 - 1, qh = ph + (tt / 3) % 3 - 1, qw = pw + tt % 3 - 1;
            const bool ok = t < 27 && (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W;
            const int64_t qrow = ok ? It.row_off[it] + ((int64_t)qd * H + qh) * W + qw : row;
            const float v = Elem<T>::ld(x[qrow * cin_p + ci]);
            xv[u] = ok ? v : 0.f;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const bool gv = g < G && c0 + g < cout;
            const float* wg = w + ((int64_t)(gv ? c0 + g : 0) * cin + ci) * 27;
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int t = tg + 4 * u;
                const float wv = wg[t < 27 ? t : 0];
                acc[g] = fmaf(gv ? wv : 0.f, xv[u], acc[g]);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float s = wave_sum_f32(acc[g]);
        if ((tid & 63) == 0) red[tid >> 6][g] = s;
    }
    __syncthreads();
    if (tid < G) {
        float v = 0.f;
#pragma unroll
        for (int wv_ = 0; wv_ < 8; ++wv_) v += red[wv_][tid];
        if (bias && c0 + tid < cout) v += bias[c0 + tid];
        raw[(int64_t)k * G + tid] = v;
        out[(int64_t)k * G + tid] = Lv.scale[l] ? v * *Lv.scale[l] : v;
    }
    if (tid == 0) { rows[k] = (int32_t)row; c0s[k] = c0; lvls[k] = l; }
}

static int ho_fill(const NndetHeadLevels* levels, int32_t N, int32_t A, int32_t G, const int64_t* level_row0_host, HoLevels* Lv) {
    if (!levels || levels->nlev <= 0 || levels->nlev > HO_MAX_LEVELS || N <= 0 || A <= 0 || G <= 0 || G > 8 || !level_row0_host) return NNDET_EINVAL;
    memset(Lv, 0, sizeof(*Lv));
    Lv->nlev = levels->nlev; Lv->A = A; Lv->G = G; Lv->N = N;
    int64_t off = 0;
    for (int l = 0; l < levels->nlev; ++l) {
        Lv->anchor_off[l] = off; Lv->points[l] = levels->points[l]; Lv->row0[l] = level_row0_host[l];
        Lv->scale[l] = reinterpret_cast<const float*>(levels->scale[l]); Lv->dscale[l] = reinterpret_cast<float*>(levels->dscale[l]);
        off += levels->points[l] * A;
    }
    Lv->anchor_off[levels->nlev] = off;
    Lv->anchors_per_image = off;
    return 0;
}

static void ho_items(const NndetItems* items, HoItems* It) {
    memset(It, 0, sizeof(*It));
    It->n = items->n_items;
    for (int i = 0; i < items->n_items; ++i) {
        for (int a = 0; a < 3; ++a) It->dims[i][a] = items->dims[i][a];
        It->row_off[i] = items->row_off[i];
    }
}

extern "C" int nndet_conv_out_sparse_forward(const NndetConv* c, const NndetItems* items, const NndetHeadLevels* levels, int32_t N,
                                             int32_t A, int32_t G, const int64_t* level_row0_host, const int64_t* idx, int32_t K,
                                             const void* x, const float* w_f32, const float* bias, float* out, float* raw_out,
                                             int32_t* rows_out, int32_t* c0_out, int32_t* level_out, void* stream) {
    if (!c || !items || items->n_items < 1 || items->n_items > NNDET_MAX_ITEMS || K < 0) return NNDET_EINVAL;
    if (c->transposed || c->cin_p % 32 || c->cout_p % 32 || A * G > c->cout_p) return NNDET_EINVAL;
    for (int i = 0; i < 3; ++i) if (c->k[i] != 3 || c->s[i] != 1 || c->p[i] != 1) return NNDET_EINVAL;
    HoLevels Lv;
    int rc = ho_fill(levels, N, A, G, level_row0_host, &Lv);
    if (rc) return rc;
    if (K == 0) return 0;
    if (!idx || !x || !w_f32 || !out || !raw_out || !rows_out || !c0_out || !level_out) return NNDET_EINVAL;
    HoItems It;
    ho_items(items, &It);
    hipStream_t st = as_stream(stream);
    if (c->dtype == NNDET_BF16) k_ho_forward<bf16_t><<<K, 512, 0, st>>>(idx, K, Lv, It, (const bf16_t*)x, c->cin, c->cin_p, w_f32, bias, c->cout, out, raw_out, rows_out, c0_out, level_out);
    else if (c->dtype == NNDET_F16) k_ho_forward<f16_t><<<K, 512, 0, st>>>(idx, K, Lv, It, (const f16_t*)x, c->cin, c->cin_p, w_f32, bias, c->cout, out, raw_out, rows_out, c0_out, level_out);
    else if (c->dtype == NNDET_F32) k_ho_forward<float><<<K, 512, 0, st>>>(idx, K, Lv, It, (const float*)x, c->cin, c->cin_p, w_f32, bias, c->cout, out, raw_out, rows_out, c0_out, level_out);
    else return NNDET_EINVAL;
    LAUNCH_CHECK();
    return 0;
}

// Backward of the Scale layer at the K sampled anchors of the sparse regressor forward (k_ho_forward produced out = scale_l * raw):
// vals[k][g] = grad[k][g] * scale_l (the gradient of the conv output, the input of k_ho_backward) and dscale[l] = sum over the
// anchors of level l of grad . raw. One small block: K <= a few hundred. Replaces 13 element-wise / index launches in autograd.
__global__ __launch_bounds__(256) void k_ho_scale_bwd(const float* __restrict__ g, const float* __restrict__ raw, const int32_t* __restrict__ lvl,
                                                      int K, int G, HoLevels Lv, float* __restrict__ vals) {
    __shared__ float ds[HO_MAX_LEVELS];
    if (threadIdx.x < HO_MAX_LEVELS) ds[threadIdx.x] = 0.f;
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const int l = lvl[k];
        const float sc = (l >= 0 && Lv.scale[l]) ? *Lv.scale[l] : (l >= 0 ? 1.f : 0.f);
        float d = 0.f;
        for (int j = 0; j < G; ++j) {
            const float gv = g[(int64_t)k * G + j];
            vals[(int64_t)k * G + j] = gv * sc;
            d = fmaf(gv, raw[(int64_t)k * G + j], d);
        }
        if (l >= 0) atomicAdd(&ds[l], d);
    }
    __syncthreads();
    if ((int)threadIdx.x < Lv.nlev && Lv.dscale[threadIdx.x]) *Lv.dscale[threadIdx.x] = ds[threadIdx.x];
}

extern "C" int nndet_conv_out_sparse_scale_backward(const NndetHeadLevels* levels, const float* grad, const float* raw, const int32_t* level,
                                                    int32_t K, int32_t G, float* vals_out, void* stream) {
    if (!levels || levels->nlev <= 0 || levels->nlev > HO_MAX_LEVELS || K < 0 || G <= 0 || G > 8) return NNDET_EINVAL;
    if (K == 0) return 0;
    if (!grad || !raw || !level || !vals_out) return NNDET_EINVAL;
    HoLevels Lv;
    memset(&Lv, 0, sizeof(Lv));
    Lv.nlev = levels->nlev; Lv.G = G;
    for (int l = 0; l < levels->nlev; ++l) {
        Lv.scale[l] = reinterpret_cast<const float*>(levels->scale[l]); Lv.dscale[l] = reinterpret_cast<float*>(levels->dscale[l]);
    }
    k_ho_scale_bwd<<<1, 256, 0, as_stream(stream)>>>(grad, raw, level, K, G, Lv, vals_out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_head_out_sparse_scatter(int32_t dtype, const NndetHeadLevels* levels, int32_t N, int32_t A, int32_t G,
                                             const int64_t* level_row0_host, const int64_t* idx, const float* val, int32_t K,
                                             const void* y, int32_t cout_p, void* dy_zeroed, int32_t* rows_out, int32_t* c0_out,
                                             float* vals_out, void* stream) {
    if (!levels || levels->nlev <= 0 || levels->nlev > HO_MAX_LEVELS || N <= 0 || A <= 0 || G <= 0 || G > 8 || K < 0 || !level_row0_host)
        return NNDET_EINVAL;
    if (K == 0) return 0;
    if (!idx || !val || !y || !dy_zeroed || !rows_out || !c0_out || !vals_out || cout_p % 32 || A * G > cout_p) return NNDET_EINVAL;
    HoLevels Lv;
    memset(&Lv, 0, sizeof(Lv));
    Lv.nlev = levels->nlev; Lv.A = A; Lv.G = G; Lv.N = N;
    int64_t off = 0;
    for (int l = 0; l < levels->nlev; ++l) {
        Lv.anchor_off[l] = off; Lv.points[l] = levels->points[l]; Lv.row0[l] = level_row0_host[l];
        Lv.scale[l] = reinterpret_cast<const float*>(levels->scale[l]); Lv.dscale[l] = reinterpret_cast<float*>(levels->dscale[l]);
        off += levels->points[l] * A;
    }
    Lv.anchor_off[levels->nlev] = off;
    Lv.anchors_per_image = off;
    hipStream_t st = as_stream(stream);
    const int nb = ceil_div(K, 64);
    if (dtype == NNDET_BF16) k_ho_scatter<bf16_t><<<nb, 64, 0, st>>>(idx, val, K, Lv, (const bf16_t*)y, cout_p, (bf16_t*)dy_zeroed, rows_out, c0_out, vals_out);
    else if (dtype == NNDET_F16) k_ho_scatter<f16_t><<<nb, 64, 0, st>>>(idx, val, K, Lv, (const f16_t*)y, cout_p, (f16_t*)dy_zeroed, rows_out, c0_out, vals_out);
    else if (dtype == NNDET_F32) k_ho_scatter<float><<<nb, 64, 0, st>>>(idx, val, K, Lv, (const float*)y, cout_p, (float*)dy_zeroed, rows_out, c0_out, vals_out);
    else return NNDET_EINVAL;
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_conv_out_sparse_backward(const NndetConv* c, const NndetItems* items, const int32_t* rows, const int32_t* c0,
                                              const float* vals, int32_t K, int32_t G, const void* x, const float* w_f32,
                                              float* dx32_scratch, void* dx_zeroed, float* dw, float* dbias, void* stream) {
    if (!c || !items || items->n_items < 1 || items->n_items > NNDET_MAX_ITEMS || K < 0 || G <= 0 || G > 8) return NNDET_EINVAL;
    if (c->transposed || c->cin_p % 32 || c->cout_p % 32) return NNDET_EINVAL;
    for (int i = 0; i < 3; ++i) if (c->k[i] != 3 || c->s[i] != 1 || c->p[i] != 1) return NNDET_EINVAL;
    if (!x || !w_f32 || !dx32_scratch || !dx_zeroed || !dw || c->cin > 512) return NNDET_EINVAL;
    if (K == 0) return 0;
    if (!rows || !c0 || !vals) return NNDET_EINVAL;
    hipStream_t st = as_stream(stream);
    HoItems It;
    ho_items(items, &It);
    const dim3 grid(K, 27);
#define HO_BWD(T_, MODE_) k_ho_backward<T_, MODE_><<<grid, 128, 0, st>>>(rows, c0, vals, G, It, (const T_*)x, c->cin, c->cin_p, w_f32, c->cout, \
                                                                         dx32_scratch, (T_*)dx_zeroed, dw, dbias)
#define HO_ALL(T_) do { HO_BWD(T_, 0); HO_BWD(T_, 1); HO_BWD(T_, 2); \
                        if (c->cin <= 128) k_ho_wgrad<T_, 1><<<dim3(c->cout, 27), 128, 0, st>>>(rows, c0, vals, K, G, It, (const T_*)x, c->cin, c->cin_p, c->cout, dw, dbias); \
                        else if (c->cin <= 256) k_ho_wgrad<T_, 2><<<dim3(c->cout, 27), 128, 0, st>>>(rows, c0, vals, K, G, It, (const T_*)x, c->cin, c->cin_p, c->cout, dw, dbias); \
                        else k_ho_wgrad<T_, 4><<<dim3(c->cout, 27), 128, 0, st>>>(rows, c0, vals, K, G, It, (const T_*)x, c->cin, c->cin_p, c->cout, dw, dbias); } while (0)
    if (c->dtype == NNDET_BF16) HO_ALL(bf16_t);
    else if (c->dtype == NNDET_F16) HO_ALL(f16_t);
    else if (c->dtype == NNDET_F32) HO_ALL(float);
    else return NNDET_EINVAL;
#undef HO_ALL
#undef HO_BWD
    LAUNCH_CHECK();
    return 0;
}
