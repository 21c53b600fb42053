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
//   k_segbranch_fwd: tile of 4 x 8 x 16 outputs per pass, halo 6 x 10 x 18 voxels x 64 B staged in LDS (16-byte parts XOR-swizzled,
//                    odd row pitch: conflict-free ds_read_b128), two W-adjacent outputs per thread, v_dot2c_f32_{bf16,f16} with the
//                    composed weights as SCALAR operands (s_load, 16 dwords per tap); writes z (fp32) and the four loss sums.
//                    HBM-bound by design: reads x once (+ halo re-reads from L2), writes 4 B per voxel.
//   k_segbranch_bwd: d1 = dL/dz from (z, target, the Jacobian of the scalar tail) -> 16-bit d1 + sum(d1); streaming.
#include "common.h"

template <typename T> struct Dot2;
template <> struct Dot2<bf16_t> {
    __device__ static __forceinline__ float f(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
    }
};
template <> struct Dot2<f16_t> {
    __device__ static __forceinline__ float f(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
    }
};

#define SB_TD 4
#define SB_TH 8
#define SB_TW 16
#define SB_HD (SB_TD + 2)
#define SB_HH (SB_TH + 2)
#define SB_HW (SB_TW + 2)
#define SB_PITCH 19                                   // voxels per halo row in LDS (odd: rows start on different bank quarters)
#define SB_LDS (SB_HD * SB_HH * SB_PITCH * 64)
#define SB_REPL 16                                    // replicas of the loss sums (fp64 atomics of ~2000 workgroups)

__device__ __forceinline__ float sb_softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

template <typename T>
__global__ __launch_bounds__(256, 2) void k_segbranch_fwd(const T* __restrict__ x, const uint32_t* __restrict__ wq, const float* __restrict__ c0p,
                                                          const uint8_t* __restrict__ target, int N, int D, int H, int W,
                                                          float* __restrict__ z, double* __restrict__ sums) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    if (tid < 4) red[tid] = 0.0;
    const int ntd = (D + SB_TD - 1) / SB_TD, nth = (H + SB_TH - 1) / SB_TH, ntw = (W + SB_TW - 1) / SB_TW;
    const int per_img = ntd * nth * ntw, ntiles = per_img * N;
    const float c0 = *c0p;
    const int img_bytes = D * H * W * 64;                              // < 2^31 (host check)
    const int wl = tid & 7, hl = (tid >> 3) & 7, dl = tid >> 6;
    // LDS addresses of this thread's four input columns (halo w index 2 wl + j), 16-byte part q: voxel * 64 + ((q ^ swz) << 4)
    int addr[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int hw = 2 * wl + j;
        const int vox = (dl * SB_HH + hl) * SB_PITCH + hw;
#pragma unroll
        for (int q = 0; q < 4; ++q) addr[j][q] = vox * 64 + ((q ^ ((hw >> 2) & 3)) << 4);
    }
    float ce = 0.f, tp = 0.f, fp = 0.f, fn = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / per_img;
        int tt = tile - n * per_img;
        const int tw_i = tt % ntw; tt /= ntw;
        const int th_i = tt % nth;
        const int td_i = tt / nth;
        const int d0 = td_i * SB_TD, h0 = th_i * SB_TH, w0 = tw_i * SB_TW;
        const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(x)) + (int64_t)n * img_bytes,
                                                           0, img_bytes, 0x00020000);
        __syncthreads();                                               // the previous tile's reads are done
        // ---- stage the halo: 1080 voxels x 4 parts of 16 bytes; out-of-volume pieces read zeros (offset beyond num_records)
        constexpr int NPIECE = SB_HD * SB_HH * SB_HW * 4, NIT = (NPIECE + 255) / 256;
        u32x4 v[NIT];
        int dst[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int id = tid + k * 256;
            const int hv = id >> 2, q = id & 3;
            const int hd = hv / (SB_HH * SB_HW), r = hv - hd * (SB_HH * SB_HW);
            const int hh = r / SB_HW, hw = r - hh * SB_HW;
            const int gd = d0 - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
            const bool ok = id < NPIECE && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            const int off = ok ? ((gd * H + gh) * W + gw) * 64 + q * 16 : (int)0x80000000;
            v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
            dst[k] = ((hd * SB_HH + hh) * SB_PITCH + hw) * 64 + ((q ^ ((hw >> 2) & 3)) << 4);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if ((k + 1) * 256 <= NPIECE || tid + k * 256 < NPIECE) *reinterpret_cast<u32x4*>(smem + dst[k]) = v[k];   // (only the last round is partial)
        __syncthreads();
        // ---- 27 taps x 32 channels for the two outputs (w, w + 1) of this thread; two accumulation chains per output.
        // A REAL loop over the 9 (kd, kh) rows (fully unrolled the compiler hoists all 144 LDS reads and all 432 weight dwords to the
        // top and spills): per row 16 ds_read_b128 (4 input columns), 3 x 16 weight dwords through the SCALAR cache (uniform address
        // -> s_load_dwordx16; the constant address space keeps them scalar whatever the alias analysis thinks of the stores to z),
        // 96 v_dot2c. The second workgroup of the CU covers the latency at the top of each row.
        float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
        typedef const uint32_t __attribute__((address_space(4))) * cw_t;
        const cw_t wc = (cw_t) reinterpret_cast<uint64_t>(wq);
#pragma unroll 1
        for (int row = 0; row < 9; ++row) {
            const int rowoff = (((row / 3) * SB_HH + row % 3) * SB_PITCH) * 64;
            u32x4 xv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) xv[j][q] = *reinterpret_cast<const u32x4*>(smem + addr[j][q] + rowoff);
            uint32_t wt[3][16];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int i = 0; i < 16; ++i) wt[kw][i] = wc[(row * 3 + kw) * 16 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j <= 2) {                                           // output 0 (at halo column 2 wl + 1): tap kw = j
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        a0 = Dot2<T>::f(xv[j][q][0], wt[j][q * 4 + 0], a0); a1 = Dot2<T>::f(xv[j][q][1], wt[j][q * 4 + 1], a1);
                        a0 = Dot2<T>::f(xv[j][q][2], wt[j][q * 4 + 2], a0); a1 = Dot2<T>::f(xv[j][q][3], wt[j][q * 4 + 3], a1);
                    }
                }
                if (j >= 1) {                                           // output 1 (at halo column 2 wl + 2): tap kw = j - 1
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        b0 = Dot2<T>::f(xv[j][q][0], wt[j - 1][q * 4 + 0], b0); b1 = Dot2<T>::f(xv[j][q][1], wt[j - 1][q * 4 + 1], b1);
                        b0 = Dot2<T>::f(xv[j][q][2], wt[j - 1][q * 4 + 2], b0); b1 = Dot2<T>::f(xv[j][q][3], wt[j - 1][q * 4 + 3], b1);
                    }
                }
            }
        }
        // ---- epilogue: z and the loss sums of the two voxels
        const int od = d0 + dl, oh = h0 + hl, ow = w0 + 2 * wl;
        if (od < D && oh < H) {
            const int64_t base = (((int64_t)n * D + od) * H + oh) * W + ow;
            const float zz[2] = {a0 + a1 + c0, b0 + b1 + c0};
#pragma unroll
            for (int o = 0; o < 2; ++o)
                if (ow + o < W) {
                    const float zv = zz[o];
                    z[base + o] = zv;
                    const bool t = target[base + o] > 0;
                    const float p1 = 1.f / (1.f + expf(-zv));
                    ce += t ? sb_softplus(-zv) : sb_softplus(zv);       // -log softmax(l)[t]
                    if (t) { tp += p1; fn += 1.f - p1; } else { fp += p1; }
                }
        }
    }
    double dsum[4] = {(double)ce, (double)tp, (double)fp, (double)fn};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double s = wave_sum_f64(dsum[k]);
        if ((tid & 63) == 0) atomicAdd(&red[k], s);
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

extern "C" int nndet_segbranch_replicas(void) { return SB_REPL; }

extern "C" int nndet_segbranch_forward(int32_t dtype, const void* x, int32_t N, int32_t D, int32_t H, int32_t W, int32_t c_p,
                                       const void* w_packed, const float* c0, const uint8_t* target, float* z_out, double* sums_out,
                                       void* stream) {
    if (!x || !w_packed || !c0 || !target || !z_out || !sums_out || N <= 0 || D <= 0 || H <= 0 || W <= 0 || c_p != 32) return NNDET_EINVAL;
    if (!nndet_is16(dtype)) return NNDET_EINVAL;                       // the fp32 path keeps the two separate layers
    if ((int64_t)D * H * W * 64 >= (1LL << 31)) return NNDET_EINVAL;    // 32-bit buffer offsets per image
    const int64_t ntiles = (int64_t)ceil_div(D, SB_TD) * ceil_div(H, SB_TH) * ceil_div(W, SB_TW) * N;
    const unsigned nb = (unsigned)(ntiles < 2048 ? ntiles : 2048);
    static int attr_done = 0;
    if (!attr_done) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_segbranch_fwd<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_segbranch_fwd<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS));
        attr_done = 1;
    }
    hipStream_t st = as_stream(stream);
    if (dtype == NNDET_BF16)
        k_segbranch_fwd<bf16_t><<<nb, 256, SB_LDS, st>>>((const bf16_t*)x, (const uint32_t*)w_packed, c0, target, N, D, H, W, z_out, sums_out);
    else
        k_segbranch_fwd<f16_t><<<nb, 256, SB_LDS, st>>>((const f16_t*)x, (const uint32_t*)w_packed, c0, target, N, D, H, W, z_out, sums_out);
    LAUNCH_CHECK();
    return 0;
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
