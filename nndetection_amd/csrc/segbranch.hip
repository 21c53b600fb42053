// The segmentation branch of a TRAINING step as one 32 -> 1 convolution fused with the loss (gfx950).
//
// Reference chain (nndet/arch/decoder/base.py:243-270 `out.P0` = Conv3d 3x3x3 32 -> 32 + bias, no norm / activation;
// nndet/arch/heads/segmenter.py:184-206,273-289 DiCESegmenterFgBg: Conv3d 1x1x1 32 -> 2 + bias, softmax, CE + SoftDice):
//     y = conv3(x; W) + b                      [32 channels, full resolution: 629 MB in 16 bits at 160x160x96, batch 4]
//     (l0, l1) = w_head y + b_head             [2 logits per voxel]
//     loss = f(softmax(l), target)             [fg / bg: depends on z = l1 - l0 only]
// Both layers are linear and nothing but the loss reads y or the logits when no prediction is asked for, so
//     z[p] = c0 + sum_{t, cin} wc[t][cin] x[p + t - 1][cin],   wc[t][cin] = sum_c (w1 - w0)[c] W[c][cin][t],
//     c0 = (w1 - w0) . b + (b1 - b0)
// is ONE 3x3x3 convolution with a single output channel: 27 x 32 MACs per voxel instead of 27 x 32 x 32 + 64, the 32-channel map y
// never exists (0.54 TFLOP, one 629 MB write and one 629 MB read per step disappear). The backward side of the same algebra has
// been in place since the rank-1 gradient (arch/conv.py: _rank1_backward); the head's weight gradient follows from the one-channel
// correlation E the producer convolution's weight gradient needs anyway (arch/segmenter.py: _SegBranchFn).
//
//   k_segbranch_fwd: per 4 x 8 x 8 output tile: u[t][q] = wc[t] . x[q] for the 600 halo voxels q as 38 x 2 MFMAs 16x16x32 (rows = taps,
//                    columns = voxels; x goes from memory straight into the B operand), u through LDS (65 KB), then z[p] = c0 + sum_t
//                    u[t][p + t - 1] (27 conflict-free ds_read_b32 + adds per output) and the four loss sums. Bound by the read of x.
//   k_segbranch_bwd: d1 = dL/dz from (z, target, the Jacobian of the scalar tail) -> 16-bit d1 + sum(d1); streaming.
#include "common.h"

#define SB_TD 4
#define SB_TH 8
#define SB_TW 8
#define SB_HD (SB_TD + 2)
#define SB_HH (SB_TH + 2)
#define SB_HW (SB_TW + 2)
#define SB_NHV (SB_HD * SB_HH * SB_HW)                // 600 halo voxels
#define SB_NGRP ((SB_NHV + 15) / 16)                  // 38 groups of 16 voxels (one MFMA column block each)
#define SB_UP (SB_NGRP * 16)                          // pitch of one tap row of u (608 floats)
#define SB_LDS (32 * SB_UP * 4)                       // u[tap (27 used of 32)][halo voxel] fp32 = 77 824 bytes -> 2 workgroups per CU
#define SB_REPL 16                                    // replicas of the loss sums (fp64 atomics of ~2000 workgroups)

__device__ __forceinline__ float sb_softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

// With ONE output channel the convolution is cheapest as "per-voxel tap products, then a shifted sum":
//     u[t][q] = sum_cin wc[t][cin] x[q][cin]      for every voxel q of the tile's halo  -- a 32(taps, 27 used) x 32 x 16-voxel MFMA pair
//     z[p]    = c0 + sum_t u[t][p + t - 1]                                               -- 27 LDS reads + adds per output
// Every x value is read from memory exactly once per tile, straight into the MFMA B operand (no LDS staging of x: nothing re-uses
// it), 76 MFMAs per 256 outputs; the kernel is bound by the 629 MB read of x (halo re-reads come from L2).
// TWO: a second 32-channel input x2 with its own composed weights wq2, accumulated into the same tap products (K = 64): the decoder's
// lateral convolution absorbed as well (z = conv3(a0; wc . W_lat) + conv3(up; wc) + c0, see nndet_segbranch_forward2).
// UP: a third term that is already reduced to one value per voxel -- the absorbed top-down step (arch/segmenter.py: NNDET_SEG_UP):
// zup [N, D/2, H/2, W/2, 32] holds, in channel (pd * 2 + ph) * 2 + pw of the half-resolution voxel m, the contribution to the output
// voxel 2 m + (pd, ph, pw); cb[27] is the bias of the voxel's border class (cd * 3 + ch) * 3 + cw (0 first plane, 1 inside, 2 last).
template <typename T, bool TWO, bool UP = false>
__global__ __launch_bounds__(256, 2) void k_segbranch_fwd(const T* __restrict__ x, const uint32_t* __restrict__ wq, const T* __restrict__ x2,
                                                          const uint32_t* __restrict__ wq2, const float* __restrict__ c0p,
                                                          const uint8_t* __restrict__ target, int N, int D, int H, int W,
                                                          float* __restrict__ z, double* __restrict__ sums,
                                                          const T* __restrict__ zup = nullptr, const float* __restrict__ cbp = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* u = reinterpret_cast<float*>(smem);
    __shared__ double red[4];
    __shared__ float scb[27];
    if constexpr (UP) { if (threadIdx.x < 27) scb[threadIdx.x] = cbp[threadIdx.x]; }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    if (tid < 4) red[tid] = 0.0;
    const int ntd = (D + SB_TD - 1) / SB_TD, nth = (H + SB_TH - 1) / SB_TH, ntw = (W + SB_TW - 1) / SB_TW;
    const int per_img = ntd * nth * ntw, ntiles = per_img * N;
    const float c0 = *c0p;
    const int img_bytes = D * H * W * 64;                              // < 2^31 (host check)
    // A operand (weights): lane (li, q) holds tap row li (+ 16 for the second row tile), channels 8 q .. 8 q + 7; taps >= 27 are zero
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(wq + li * 16 + q * 4);
    const u32x4 a1 = (16 + li < 27) ? *reinterpret_cast<const u32x4*>(wq + (16 + li) * 16 + q * 4) : zero4;
    u32x4 a20 = zero4, a21 = zero4;
    if constexpr (TWO) {
        a20 = *reinterpret_cast<const u32x4*>(wq2 + li * 16 + q * 4);
        a21 = (16 + li < 27) ? *reinterpret_cast<const u32x4*>(wq2 + (16 + li) * 16 + q * 4) : zero4;
    }
    constexpr int NIT = (SB_NGRP + 3) / 4;                             // groups per wave (10)
    // this thread's output voxel and the halo index of its tap (0, 0, 0)
    const int ow_l = tid & 7, oh_l = (tid >> 3) & 7, od_l = tid >> 6;
    const int ubase = (od_l * SB_HH + oh_l) * SB_HW + ow_l;
    float ce = 0.f, tp = 0.f, fp = 0.f, fn = 0.f;
    // B operands of one tile: 16 bytes of voxel (group g = wave + 4 k, column li), channels 8 q ..; out-of-volume voxels read zeros
    // (the conv padding). The loads of tile i + 1 are issued BEFORE tile i is processed: a tile is ~1 us of work against ~2 us of
    // memory latency, and only two workgroups share a CU.
    auto load_tile = [&](int tile, u32x4* bv, float& zu) {
        const int n = tile / per_img;
        int tt = tile - n * per_img;
        const int tw_i = tt % ntw; tt /= ntw;
        const int th_i = tt % nth;
        const int td_i = tt / nth;
        const int d0 = td_i * SB_TD, h0 = th_i * SB_TH, w0 = tw_i * SB_TW;
        if constexpr (UP) {                                             // this thread's output voxel of that tile: its value of the third term
            const int od = d0 + od_l, oh = h0 + oh_l, ow = w0 + ow_l;
            zu = 0.f;
            if (od < D && oh < H && ow < W) {
                const int64_t m = (((int64_t)n * (D >> 1) + (od >> 1)) * (H >> 1) + (oh >> 1)) * (W >> 1) + (ow >> 1);
                zu = Elem<T>::ld(zup[m * 32 + (((od & 1) * 2 + (oh & 1)) * 2 + (ow & 1))]);
            }
        }
        const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(x)) + (int64_t)n * img_bytes,
                                                           0, img_bytes, 0x00020000);
        const auto xrs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(TWO ? x2 : x)) + (int64_t)n * img_bytes,
                                                            0, img_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int g = wv + 4 * k;
            const int hv = g * 16 + li;
            const int hd = hv / (SB_HH * SB_HW), r = hv - hd * (SB_HH * SB_HW);
            const int hh = r / SB_HW, hw = r - hh * SB_HW;
            const int gd = d0 - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
            const bool ok = g < SB_NGRP && hv < SB_NHV && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            const int off = ok ? ((gd * H + gh) * W + gw) * 64 + q * 16 : (int)0x80000000;
            bv[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
            if constexpr (TWO) bv[NIT + k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs2, off, 0, 0));
        }
    };
    constexpr int NB = TWO ? 2 * NIT : NIT;
    // Tile order: workgroup ids are dealt round-robin to the 8 XCDs (one L2 each). XCD k walks over the k-th eighth of the tiles, its
    // gridDim / 8 workgroups side by side over consecutive tiles, so the halo overlap of neighbouring tiles (2.3 x the tile itself) is
    // re-read from THAT L2 instead of from HBM (measured without: 0.29 ms = the 1.47 GB of tile + halo reads at the HBM rate).
    const int nx = gridDim.x >> 3;                                     // workgroups per XCD (the host launches a multiple of 8)
    const int t8 = (ntiles + 7) >> 3;                                  // tiles per XCD
    const int xcd = blockIdx.x & 7, lw = blockIdx.x >> 3;
    const int t_begin = xcd * t8 + lw, t_end = min((xcd + 1) * t8, ntiles);
    u32x4 bv[NB], bn[NB];
    float zu = 0.f, zun = 0.f;
    if (t_begin < t_end) load_tile(t_begin, bv, zu);
    for (int tile = t_begin; tile < t_end; tile += nx) {
        const int n = tile / per_img;
        int tt = tile - n * per_img;
        const int tw_i = tt % ntw; tt /= ntw;
        const int th_i = tt % nth;
        const int td_i = tt / nth;
        const int d0 = td_i * SB_TD, h0 = th_i * SB_TH, w0 = tw_i * SB_TW;
        const int next = tile + nx;
        if (next < t_end) load_tile(next, bn, zun);
        // (LDS-only barriers: __syncthreads() would also wait for the loads of the NEXT tile that were just issued)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the previous tile's reads of u are done
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int g = wv + 4 * k;
            if (g < SB_NGRP) {                                         // wave-uniform
                const f32x4 zf = {0.f, 0.f, 0.f, 0.f};
                f32x4 d0v = H16<T>::mma(a0, bv[k], zf);                 // taps 4 q .. 4 q + 3 of voxel (g, li)
                f32x4 d1v = H16<T>::mma(a1, bv[k], zf);                 // taps 16 + 4 q ..
                if constexpr (TWO) { d0v = H16<T>::mma(a20, bv[NIT + k], d0v); d1v = H16<T>::mma(a21, bv[NIT + k], d1v); }
                float* up = u + g * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) up[(4 * q + r) * SB_UP] = d0v[r];
#pragma unroll
                for (int r = 0; r < 4; ++r) up[(16 + 4 * q + r) * SB_UP] = d1v[r];      // (rows 27 .. 31: zero weights, never read)
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- z = c0 + the 27 shifted taps; three accumulation chains
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float* ur = u + ((kd * 3 + kh) * 3) * SB_UP + ubase + (kd * SB_HH + kh) * SB_HW;
                s0 += ur[0]; s1 += ur[SB_UP + 1]; s2 += ur[2 * SB_UP + 2];
            }
        const int od = d0 + od_l, oh = h0 + oh_l, ow = w0 + ow_l;
        if (od < D && oh < H && ow < W) {
            const int64_t idx = (((int64_t)n * D + od) * H + oh) * W + ow;
            float zv = (s0 + s1) + (s2 + c0);
            if constexpr (UP) {
                const int cd = od == 0 ? 0 : (od == D - 1 ? 2 : 1), ch = oh == 0 ? 0 : (oh == H - 1 ? 2 : 1), cw = ow == 0 ? 0 : (ow == W - 1 ? 2 : 1);
                zv += zu + scb[(cd * 3 + ch) * 3 + cw];
            }
            z[idx] = zv;
            const bool t = target[idx] > 0;
            const float p1 = 1.f / (1.f + expf(-zv));
            ce += t ? sb_softplus(-zv) : sb_softplus(zv);               // -log softmax(l)[t]
            if (t) { tp += p1; fn += 1.f - p1; } else { fp += p1; }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) bv[k] = bn[k];
        zu = zun;
    }
    double dsum[4] = {(double)ce, (double)tp, (double)fp, (double)fn};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double sv = wave_sum_f64(dsum[k]);
        if (lane == 0) atomicAdd(&red[k], sv);
    }
    __syncthreads();
    if (tid < 4) atomicAdd(&sums[(blockIdx.x % SB_REPL) * 4 + tid], red[tid]);
}

// d1 = dL/dz per voxel (the same expression as k_seghead_bwd), stored in the activation type for the stem kernels that consume it
template <typename T>
__global__ __launch_bounds__(256) void k_segbranch_bwd(const float* __restrict__ z, const uint8_t* __restrict__ target, int64_t nvox,
                                                       const float* __restrict__ coeffs, T* __restrict__ d1_out, double* __restrict__ dsum) {
    __shared__ double red;
    if (threadIdx.x == 0) red = 0.0;
    __syncthreads();
    const float g_ce = coeffs[0], g_tp = coeffs[1], g_fp = coeffs[2], g_fn = coeffs[3];
    float acc = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * 256) {
        const float zv = z[v];
        const bool t = target[v] > 0;
        const float p1 = 1.f / (1.f + expf(-zv));
        const float dp = p1 * (1.f - p1);
        float d1 = g_ce * (p1 - (t ? 1.f : 0.f));
        d1 += dp * (t ? (g_tp - g_fn) : g_fp);
        const T r = Elem<T>::st(d1);
        d1_out[v] = r;
        acc += Elem<T>::ld(r);                                          // the sum of what the consumers read
    }
    const double s = wave_sum_f64((double)acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(&red, s);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&dsum[blockIdx.x % SB_REPL], red);
}

// NNDET_SEG_UP, backward side: the space-to-depth copy of d1 that the composed half-resolution convolution takes as its output
// gradient -- dzs[n][m][(pd * 2 + ph) * 2 + pw] = d1[n][2 m + (pd, ph, pw)], channels 8 .. 31 zero -- and the sum of d1 per border class
// (27 sums; csum [SB_REPL][27], zeroed by the caller), from which the host forms S[t] = sum of d1 over the voxels whose tap t stays
// inside the volume. One thread per half-resolution voxel; D, H, W even.
template <typename T>
__global__ __launch_bounds__(256) void k_segbranch_s2d(const T* __restrict__ d1, int N, int D, int H, int W, T* __restrict__ dzs,
                                                       double* __restrict__ csum) {
    __shared__ double bins[27];
    if (threadIdx.x < 27) bins[threadIdx.x] = 0.0;
    __syncthreads();
    const int D2 = D >> 1, H2 = H >> 1, W2 = W >> 1;
    const int64_t total = (int64_t)N * D2 * H2 * W2;
    float inner = 0.f;
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < total; m += (int64_t)gridDim.x * 256) {
        const int mw = (int)(m % W2);
        int64_t r = m / W2;
        const int mh = (int)(r % H2); r /= H2;
        const int md = (int)(r % D2);
        const int n = (int)(r / D2);
        uint32_t pk[4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int od = 2 * md + a, oh = 2 * mh + b;
                const int64_t v = (((int64_t)n * D + od) * H + oh) * W + 2 * mw;
                const uint32_t two = *reinterpret_cast<const uint32_t*>(d1 + v);        // (pw = 0, 1): 4-byte aligned, W even
                pk[a * 2 + b] = two;
                const float v0 = H16<T>::lo(two), v1 = H16<T>::hi(two);
                const int cd = od == 0 ? 0 : (od == D - 1 ? 2 : 1), ch = oh == 0 ? 0 : (oh == H - 1 ? 2 : 1);
                const int c0 = mw == 0 ? 0 : 1, c1 = mw == W2 - 1 ? 2 : 1;               // ow = 2 mw: first or inside; 2 mw + 1: inside or last
                if (cd == 1 && ch == 1 && c0 == 1) inner += v0; else atomicAdd(&bins[(cd * 3 + ch) * 3 + c0], (double)v0);
                if (cd == 1 && ch == 1 && c1 == 1) inner += v1; else atomicAdd(&bins[(cd * 3 + ch) * 3 + c1], (double)v1);
            }
        u32x4* dst = reinterpret_cast<u32x4*>(dzs + m * 32);
        dst[0] = u32x4{pk[0], pk[1], pk[2], pk[3]};
        dst[1] = u32x4{0u, 0u, 0u, 0u}; dst[2] = u32x4{0u, 0u, 0u, 0u}; dst[3] = u32x4{0u, 0u, 0u, 0u};
    }
    const double sv = wave_sum_f64((double)inner);
    if ((threadIdx.x & 63) == 0) atomicAdd(&bins[13], sv);
    __syncthreads();
    if (threadIdx.x < 27 && bins[threadIdx.x] != 0.0) atomicAdd(&csum[(blockIdx.x % SB_REPL) * 27 + threadIdx.x], bins[threadIdx.x]);
}

extern "C" int nndet_segbranch_s2d(int32_t dtype, const void* d1, int32_t N, int32_t D, int32_t H, int32_t W, void* dzs_out, double* csum_out,
                                   void* stream) {
    if (!d1 || !dzs_out || !csum_out || N <= 0 || D < 2 || H < 2 || W < 2 || (D | H | W) & 1 || !nndet_is16(dtype)) return NNDET_EINVAL;
    const int64_t total = (int64_t)N * (D / 2) * (H / 2) * (W / 2);
    int64_t nb = ceil_div64(total, 256 * 2);
    if (nb > 4096) nb = 4096;
    hipStream_t st = as_stream(stream);
    if (dtype == NNDET_BF16) k_segbranch_s2d<bf16_t><<<(unsigned)nb, 256, 0, st>>>((const bf16_t*)d1, N, D, H, W, (bf16_t*)dzs_out, csum_out);
    else k_segbranch_s2d<f16_t><<<(unsigned)nb, 256, 0, st>>>((const f16_t*)d1, N, D, H, W, (f16_t*)dzs_out, csum_out);
    LAUNCH_CHECK();
    return 0;
}

// All parameter gradients of the branch from the one-channel correlations (tiny tensors, one workgroup): with Ec[c][t] = E[c][26 - t]
// (E = what the stem weight-gradient kernel returns for (d1, tensor): sum_p tensor[p][c] d1[p + t - 1]),
//   Ef[c][t]       = Ec_x[c][t] + sum_k W_lat[c][k] Ec_a[k][t]           (correlation of d1 with the convolution's full input)
//   dW_out[o][c][t] = wd[o] Ef[c][t],  db_out[o] = wd[o] S,  S = sum(d1)
//   Gy[o]          = sum_{c,t} W_out[o][c][t] Ef[c][t] + b_out[o] S       ->  dW_head = (-Gy, +Gy), db_head = (-S, +S)
//   dW_lat[c][k]   = sum_t wc[c][t] Ec_a[k][t],  wc[c][t] = sum_o wd[o] W_out[o][c][t]
// Replaces ~15 einsum / flip / mul launches of the autograd thread in front of the head trunks' backward chain.
__global__ __launch_bounds__(256) void k_segbranch_params(const float* __restrict__ w_out, const float* __restrict__ b_out,
                                                          const float* __restrict__ w_lat, const float* __restrict__ wd,
                                                          const float* __restrict__ e_x, const float* __restrict__ e_a,
                                                          const double* __restrict__ dsum, int nrep, float* __restrict__ dw_out,
                                                          float* __restrict__ db_out, float* __restrict__ dw_lat, float* __restrict__ dw_head,
                                                          float* __restrict__ db_head) {
    __shared__ float ef[32 * 27], eca[32 * 27], wc[32 * 27], swd[32], red[256];
    __shared__ float S;
    const int tid = threadIdx.x;
    if (tid < 32) swd[tid] = wd[tid];
    if (tid == 0) { double a = 0.0; for (int i = 0; i < nrep; ++i) a += dsum[i]; S = (float)a; }
    for (int i = tid; i < 864; i += 256) eca[i] = e_a ? e_a[(i / 27) * 27 + 26 - i % 27] : 0.f;
    __syncthreads();
    for (int i = tid; i < 864; i += 256) {
        const int c = i / 27, t = i % 27;
        float v = e_x[c * 27 + 26 - t];
        if (w_lat) for (int k = 0; k < 32; ++k) v = fmaf(w_lat[c * 32 + k], eca[k * 27 + t], v);
        ef[i] = v;
        float w = 0.f;
        for (int o = 0; o < 32; ++o) w = fmaf(swd[o], w_out[(o * 32 + c) * 27 + t], w);
        wc[i] = w;
    }
    __syncthreads();
    for (int i = tid; i < 32 * 864; i += 256) dw_out[i] = swd[i / 864] * ef[i % 864];
    if (tid < 32 && db_out) db_out[tid] = swd[tid] * S;
    if (dw_lat)
        for (int i = tid; i < 1024; i += 256) {
            const int c = i >> 5, k = i & 31;
            float v = 0.f;
            for (int t = 0; t < 27; ++t) v = fmaf(wc[c * 27 + t], eca[k * 27 + t], v);
            dw_lat[i] = v;
        }
    // Gy[o]: 8 threads per output channel o, 108 products each, reduced through LDS
    {
        const int o = tid >> 3, part = tid & 7;
        float v = 0.f;
        for (int i = part; i < 864; i += 8) v = fmaf(w_out[o * 864 + i], ef[i], v);
        red[tid] = v;
    }
    __syncthreads();
    if (tid < 32) {
        float g = 0.f;
        for (int j = 0; j < 8; ++j) g += red[tid * 8 + j];
        if (b_out) g = fmaf(b_out[tid], S, g);
        dw_head[tid] = -g; dw_head[32 + tid] = g;
    }
    if (tid == 0 && db_head) { db_head[0] = -S; db_head[1] = S; }
}

// ------------------------------------------------------------------------------------------------ parameter-only part of the absorbed branch
// Everything the fused branch with the absorbed lateral and top-down step needs from the PARAMETERS (arch/segmenter.py: _compose_up_branch),
// in one workgroup instead of ~25 torch launches (einsum / matmul / flip / cast / add ...: each a 4-6 us kernel on the branch's critical path,
// ~0.13 ms of chip time per step wherever they ran; round 6):
//   wd[c]        = w_head[1][c] - w_head[0][c]
//   wc[t][i]     = sum_c wd[c] W_out[c][i][t]                                  the composed 3x3x3 kernel, tap-major [27][32]
//   c0           = sum_c wd[c] b_out[c] + (b_head[1] - b_head[0])
//   wca[t][k]    = sum_i wc[t][i] W_lat[i][k]        -> wqa (16 bit, [27][32]) and wfa[k][26 - t] (the flipped kernel of the data gradient)
//   bsum[k]      = b_up[k] + b_lat[k];   cb[cls] = sum_t (sum_k wc[t][k] bsum[k]) K3[t][cls]
//   A[t*8+par][i] = sum_k wc[t][k] W_up[i][k][par];   Wc[pi][i][delta] = sum_r T[pi*27+delta][r] A[r][i]       (T, K3: the 0/1 tables of _up_tables)
// All in fp32; A lives in LDS (216 x I floats).
template <typename T16>
__global__ __launch_bounds__(1024) void k_segbranch_compose(const float* __restrict__ w_out, const float* __restrict__ b_out,
                                                            const float* __restrict__ w_head, const float* __restrict__ b_head,
                                                            const float* __restrict__ w_lat, const float* __restrict__ w_up,
                                                            const float* __restrict__ b_up, const float* __restrict__ b_lat,
                                                            const float* __restrict__ K3, int I,
                                                            float* __restrict__ wd_o, float* __restrict__ wc_o, float* __restrict__ c0_o,
                                                            T16* __restrict__ wqa_o, float* __restrict__ wfa_o, float* __restrict__ bsum_o,
                                                            float* __restrict__ Wc_o, float* __restrict__ cb_o) {
    // LDS: phase 1 holds W_out [32][32][27] (110 592 B); phase 2 re-uses the space for A [216][I] and W_up [I][32][8] (I <= 64: 120 832 B).
    // Every global tensor is copied in with coalesced 16-byte loads first: the sums below would otherwise be chains of dependent L2 round trips
    // (the first version of this kernel took 0.45 ms that way).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* R = reinterpret_cast<float*>(smem);
    __shared__ float wd[32], wc[27 * 32], bs[32], Bt[27], wl[32 * 32], k3[27 * 27];
    const int tid = threadIdx.x, NT = blockDim.x;
    for (int o = tid; o < 32 * 32 * 27 / 4; o += NT) reinterpret_cast<float4*>(R)[o] = reinterpret_cast<const float4*>(w_out)[o];
    for (int o = tid; o < 32 * 32; o += NT) wl[o] = w_lat[o];
    for (int o = tid; o < 27 * 27; o += NT) k3[o] = K3[o];
    if (tid < 32) {
        wd[tid] = w_head[32 + tid] - w_head[tid];
        wd_o[tid] = wd[tid];
        const float b = (b_up ? b_up[tid] : 0.f) + (b_lat ? b_lat[tid] : 0.f);
        bs[tid] = b; bsum_o[tid] = b;
    }
    __syncthreads();
    for (int o = tid; o < 27 * 32; o += NT) {
        const int t = o >> 5, i = o & 31;
        float a = 0.f;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) a = fmaf(wd[c], R[(c * 32 + i) * 27 + t], a);
        wc[o] = a; wc_o[o] = a;
    }
    if (tid == 0) {
        float a = 0.f;
        if (b_out) for (int c = 0; c < 32; ++c) a = fmaf(wd[c], b_out[c], a);
        if (b_head) a += b_head[1] - b_head[0];
        c0_o[0] = a;
    }
    __syncthreads();                                       // wc complete, W_out no longer needed
    float* A = R;                                          // [216][I]
    float* wu = R + 216 * I;                               // [I][32][8]
    for (int o = tid; o < I * 256 / 4; o += NT) reinterpret_cast<float4*>(wu)[o] = reinterpret_cast<const float4*>(w_up)[o];
    for (int o = tid; o < 27 * 32; o += NT) {
        const int t = o >> 5, k = o & 31;
        float a = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) a = fmaf(wc[t * 32 + i], wl[i * 32 + k], a);
        wqa_o[o] = Elem<T16>::st(a);
        wfa_o[k * 27 + (26 - t)] = a;
    }
    if (tid < 27) {
        float a = 0.f;
        for (int k = 0; k < 32; ++k) a = fmaf(wc[tid * 32 + k], bs[k], a);
        Bt[tid] = a;
    }
    __syncthreads();
    if (tid < 27) {
        float a = 0.f;
        for (int t = 0; t < 27; ++t) a = fmaf(Bt[t], k3[t * 27 + tid], a);
        cb_o[tid] = a;
    }
    for (int o = tid; o < 216 * I; o += NT) {             // A[(t, par)][i] = sum_k wc[t][k] W_up[i][k][par]
        const int r = o / I, i = o - r * I;
        const int t = r >> 3, par = r & 7;
        float a = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) a = fmaf(wc[t * 32 + k], wu[(i * 32 + k) * 8 + par], a);
        A[o] = a;
    }
    __syncthreads();
    // Wc[pi][i][delta] = sum over the (tap, kernel position) pairs that the selection table T of arch/segmenter.py:_up_tables holds a 1 for.
    // Per axis, an output voxel of parity p reads at the half-resolution offset e - 1 (e = 0, 1, 2) through (tap, kernel position):
    //   p = 0: e = 0: (0, 1) | e = 1: (1, 0), (2, 1) | e = 2: none;     p = 1: e = 0: none | e = 1: (0, 0), (1, 1) | e = 2: (2, 0)
    // (the 3D table is the product of the three axes: at most 8 terms per output; tests compare with the dense table product)
    for (int o = tid; o < 216 * I; o += NT) {
        const int pd = o / I, i = o - pd * I;
        const int pi = pd / 27, delta = pd - pi * 27;
        // per axis: n = number of (tap, kernel position) pairs, packed as bits: tap0 | pos0 << 2 | tap1 << 3 | pos1 << 5   (scalars only: arrays
        // indexed in loops would live in scratch memory)
        auto axis = [](int p, int e, int& n) -> int {
            if (p == 0) {
                if (e == 0) { n = 1; return 0 | (1 << 2); }
                if (e == 1) { n = 2; return 1 | (0 << 2) | (2 << 3) | (1 << 5); }
            } else {
                if (e == 1) { n = 2; return 0 | (0 << 2) | (1 << 3) | (1 << 5); }
                if (e == 2) { n = 1; return 2 | (0 << 2); }
            }
            n = 0; return 0;
        };
        int n0, n1, n2;
        const int c0_ = axis(pi >> 2, delta / 9, n0), c1_ = axis((pi >> 1) & 1, (delta / 3) % 3, n1), c2_ = axis(pi & 1, delta % 3, n2);
        float a = 0.f;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int z = 0; z < 2; ++z) {
                    if (x < n0 && y < n1 && z < n2) {
                        const int t0 = (c0_ >> (3 * x)) & 3, p0 = (c0_ >> (3 * x + 2)) & 1;
                        const int t1 = (c1_ >> (3 * y)) & 3, p1 = (c1_ >> (3 * y + 2)) & 1;
                        const int t2 = (c2_ >> (3 * z)) & 3, p2 = (c2_ >> (3 * z + 2)) & 1;
                        a += A[(((t0 * 3 + t1) * 3 + t2) * 8 + (p0 * 2 + p1) * 2 + p2) * I + i];
                    }
                }
        Wc_o[(pi * I + i) * 27 + delta] = a;
    }
}

extern "C" int nndet_segbranch_compose_up(int32_t dtype, const float* w_out, const float* b_out, const float* w_head, const float* b_head,
                                          const float* w_lat, const float* w_up, const float* b_up, const float* b_lat, const float* t_table,
                                          const float* k3_table, int32_t cin1, float* wd, float* wc, float* c0, void* wqa, float* wfa,
                                          float* bsum, float* wc_up, float* cb, void* stream) {
    if (!w_out || !w_head || !w_lat || !w_up || !t_table || !k3_table || !wd || !wc || !c0 || !wqa || !wfa || !bsum || !wc_up || !cb) return NNDET_EINVAL;
    if (!nndet_is16(dtype) || cin1 < 1 || cin1 > 64 || cin1 % 4) return NNDET_EINVAL;   // (LDS: W_out, then A [216][cin1] + W_up [cin1][256])
    (void)t_table;                                                                      // (the kernel carries the table's structure; kept in the signature: the tests' reference)
    const size_t lds_a = ((size_t)216 * cin1 + (size_t)cin1 * 256) * sizeof(float), lds_w = (size_t)32 * 32 * 27 * sizeof(float);
    const size_t lds = lds_a > lds_w ? lds_a : lds_w;
    static NndetDevOnce at;
    if (at.need()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_segbranch_compose<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_segbranch_compose<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880));
        at.done();
    }
    if (dtype == NNDET_BF16)
        k_segbranch_compose<bf16_t><<<1, 1024, lds, as_stream(stream)>>>(w_out, b_out, w_head, b_head, w_lat, w_up, b_up, b_lat, k3_table, cin1, wd, wc, c0,
                                                                         (bf16_t*)wqa, wfa, bsum, wc_up, cb);
    else
        k_segbranch_compose<f16_t><<<1, 1024, lds, as_stream(stream)>>>(w_out, b_out, w_head, b_head, w_lat, w_up, b_up, b_lat, k3_table, cin1, wd, wc, c0,
                                                                        (f16_t*)wqa, wfa, bsum, wc_up, cb);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_segbranch_param_grads(const float* w_out, const float* b_out, const float* w_lat, const float* wd, const float* e_x,
                                           const float* e_a, const double* dsum, int32_t n_dsum, float* dw_out, float* db_out,
                                           float* dw_lat, float* dw_head, float* db_head, void* stream) {
    if (!w_out || !wd || !e_x || !dsum || n_dsum <= 0 || !dw_out || !dw_head) return NNDET_EINVAL;
    if ((w_lat == nullptr) != (e_a == nullptr) || (w_lat == nullptr) != (dw_lat == nullptr)) return NNDET_EINVAL;
    k_segbranch_params<<<1, 256, 0, as_stream(stream)>>>(w_out, b_out, w_lat, wd, e_x, e_a, dsum, n_dsum, dw_out, db_out, dw_lat, dw_head, db_head);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_segbranch_replicas(void) { return SB_REPL; }

static int segbranch_launch(int32_t dtype, const void* x, const void* w_packed, const void* x2, const void* w2_packed, int32_t N, int32_t D,
                            int32_t H, int32_t W, int32_t c_p, const float* c0, const uint8_t* target, float* z_out, double* sums_out,
                            void* stream, const void* zup = nullptr, const float* cb = nullptr) {
    if (!x || !w_packed || !c0 || !target || !z_out || !sums_out || N <= 0 || D <= 0 || H <= 0 || W <= 0 || c_p != 32) return NNDET_EINVAL;
    if ((x2 == nullptr) != (w2_packed == nullptr)) return NNDET_EINVAL;
    if ((zup == nullptr) != (cb == nullptr) || (zup && (x2 || ((D | H | W) & 1)))) return NNDET_EINVAL;
    if (!nndet_is16(dtype)) return NNDET_EINVAL;                       // the fp32 path keeps the separate layers
    if ((int64_t)D * H * W * 64 >= (1LL << 31)) return NNDET_EINVAL;    // 32-bit buffer offsets per image
    const int64_t ntiles = (int64_t)ceil_div(D, SB_TD) * ceil_div(H, SB_TH) * ceil_div(W, SB_TW) * N;
    const unsigned nb = (unsigned)(ntiles < 2048 ? (ntiles + 7) / 8 * 8 : 2048);      // a multiple of 8: one share per XCD
    static NndetDevOnce attr_done;
    if (attr_done.need()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_segbranch_fwd<bf16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_segbranch_fwd<f16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_segbranch_fwd<bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_segbranch_fwd<f16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_segbranch_fwd<bf16_t, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_segbranch_fwd<f16_t, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS));
        attr_done.done();
    }
    hipStream_t st = as_stream(stream);
#define SB_GO(T_, TWO_) k_segbranch_fwd<T_, TWO_><<<nb, 256, SB_LDS, st>>>((const T_*)x, (const uint32_t*)w_packed, (const T_*)x2, \
                                                                           (const uint32_t*)w2_packed, c0, target, N, D, H, W, z_out, sums_out)
#define SB_GO3(T_) k_segbranch_fwd<T_, false, true><<<nb, 256, SB_LDS, st>>>((const T_*)x, (const uint32_t*)w_packed, nullptr, nullptr, c0, target, \
                                                                           N, D, H, W, z_out, sums_out, (const T_*)zup, cb)
    if (zup) { if (dtype == NNDET_BF16) SB_GO3(bf16_t); else SB_GO3(f16_t); }
    else if (dtype == NNDET_BF16) { if (x2) SB_GO(bf16_t, true); else SB_GO(bf16_t, false); }
    else { if (x2) SB_GO(f16_t, true); else SB_GO(f16_t, false); }
#undef SB_GO3
#undef SB_GO
    LAUNCH_CHECK();
    return 0;
}

extern "C" int nndet_segbranch_forward(int32_t dtype, const void* x, int32_t N, int32_t D, int32_t H, int32_t W, int32_t c_p,
                                       const void* w_packed, const float* c0, const uint8_t* target, float* z_out, double* sums_out,
                                       void* stream) {
    return segbranch_launch(dtype, x, w_packed, nullptr, nullptr, N, D, H, W, c_p, c0, target, z_out, sums_out, stream);
}

extern "C" int nndet_segbranch_forward2(int32_t dtype, const void* x, const void* w_packed, const void* x2, const void* w2_packed, int32_t N,
                                        int32_t D, int32_t H, int32_t W, int32_t c_p, const float* c0, const uint8_t* target, float* z_out,
                                        double* sums_out, void* stream) {
    if (!x2 || !w2_packed) return NNDET_EINVAL;
    return segbranch_launch(dtype, x, w_packed, x2, w2_packed, N, D, H, W, c_p, c0, target, z_out, sums_out, stream);
}

extern "C" int nndet_segbranch_forward_up(int32_t dtype, const void* x, const void* w_packed, const void* zup, const float* cb, int32_t N,
                                          int32_t D, int32_t H, int32_t W, int32_t c_p, const float* c0, const uint8_t* target, float* z_out,
                                          double* sums_out, void* stream) {
    if (!zup || !cb) return NNDET_EINVAL;
    return segbranch_launch(dtype, x, w_packed, nullptr, nullptr, N, D, H, W, c_p, c0, target, z_out, sums_out, stream, zup, cb);
}

extern "C" int nndet_segbranch_backward(int32_t dtype, const float* z, const uint8_t* target, int64_t nvox, const float* coeffs,
                                        void* d1_out, double* dsum_out, void* stream) {
    if (!z || !target || !coeffs || !d1_out || !dsum_out || nvox <= 0 || !nndet_is16(dtype)) return NNDET_EINVAL;
    int64_t nb = ceil_div64(nvox, 256 * 8);
    if (nb > 4096) nb = 4096;
    hipStream_t st = as_stream(stream);
    if (dtype == NNDET_BF16) k_segbranch_bwd<bf16_t><<<(unsigned)nb, 256, 0, st>>>(z, target, nvox, coeffs, (bf16_t*)d1_out, dsum_out);
    else k_segbranch_bwd<f16_t><<<(unsigned)nb, 256, 0, st>>>(z, target, nvox, coeffs, (f16_t*)d1_out, dsum_out);
    LAUNCH_CHECK();
    return 0;
}
