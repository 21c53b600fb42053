// Stem convolution (Cin == 1): forward and weight gradient. The contraction is only k^3 = 27 deep, so this
// layer is HBM-bound (reads 1 channel, writes 32) and runs on the vector ALUs; no data gradient is needed
// (the image does not require grad). Replaces encoder.stages.0.convs.0.0.conv of the reference
// (nndet/arch/encoder/modular.py:79-108 with in_channels = 1).
#include "common.h"
#include "conv_common.h"

struct StemArgs {
    const void* x; const float* w; const float* bias; void* y; float* dw; const void* dy;
    int32_t N, I[3], O[3], Cy, cout;
    int32_t k[3], s[3], p[3];
    int64_t total;   // N * O0 * O1 * O2
};

template <typename T> __device__ __forceinline__ void store_row32(T* p, const float* v) {     // 16-bit types
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint4 u;
        u.x = H16<T>::pack2(v[8 * i + 0], v[8 * i + 1]);
        u.y = H16<T>::pack2(v[8 * i + 2], v[8 * i + 3]);
        u.z = H16<T>::pack2(v[8 * i + 4], v[8 * i + 5]);
        u.w = H16<T>::pack2(v[8 * i + 6], v[8 * i + 7]);
        reinterpret_cast<uint4*>(p)[i] = u;
    }
}
template <> __device__ __forceinline__ void store_row32<float>(float* p, const float* v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) reinterpret_cast<float4*>(p)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

// grid (ceil(total/256), Cy/32); thread = one output voxel x 32 output channels
template <typename T>
__global__ __launch_bounds__(256) void k_stem_fwd(const StemArgs A) {
    __shared__ float ws[27 * 32];
    __shared__ float bs[32];
    const int taps = A.k[0] * A.k[1] * A.k[2];
    const int c0 = blockIdx.y * 32;
    for (int i = threadIdx.x; i < taps * 32; i += 256) {
        const int t = i >> 5, c = i & 31;
        ws[i] = (c0 + c < A.cout) ? A.w[(int64_t)(c0 + c) * taps + t] : 0.f;
    }
    if (threadIdx.x < 32) bs[threadIdx.x] = (A.bias && c0 + (int)threadIdx.x < A.cout) ? A.bias[c0 + threadIdx.x] : 0.f;
    __syncthreads();
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= A.total) return;
    int64_t t2 = v;
    const int ow = (int)(t2 % A.O[2]); t2 /= A.O[2];
    const int oh = (int)(t2 % A.O[1]); t2 /= A.O[1];
    const int od = (int)(t2 % A.O[0]);
    const int n = (int)(t2 / A.O[0]);
    const T* xn = reinterpret_cast<const T*>(A.x) + (int64_t)n * A.I[0] * A.I[1] * A.I[2];
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = bs[c];
    int t = 0;
    for (int kd = 0; kd < A.k[0]; ++kd)
        for (int kh = 0; kh < A.k[1]; ++kh)
            for (int kw = 0; kw < A.k[2]; ++kw, ++t) {
                const int id = od * A.s[0] - A.p[0] + kd, ih = oh * A.s[1] - A.p[1] + kh, iw = ow * A.s[2] - A.p[2] + kw;
                float xv = 0.f;
                if ((unsigned)id < (unsigned)A.I[0] && (unsigned)ih < (unsigned)A.I[1] && (unsigned)iw < (unsigned)A.I[2])
                    xv = Elem<T>::ld(xn[((int64_t)id * A.I[1] + ih) * A.I[2] + iw]);
                const float4* w4 = reinterpret_cast<const float4*>(ws + t * 32);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float4 w = w4[c];
                    acc[4 * c + 0] = fmaf(xv, w.x, acc[4 * c + 0]);
                    acc[4 * c + 1] = fmaf(xv, w.y, acc[4 * c + 1]);
                    acc[4 * c + 2] = fmaf(xv, w.z, acc[4 * c + 2]);
                    acc[4 * c + 3] = fmaf(xv, w.w, acc[4 * c + 3]);
                }
            }
    store_row32<T>(reinterpret_cast<T*>(A.y) + v * A.Cy + c0, acc);
}

// persistent grid; block 256: thread = (tap, group of 4 output channels); chunk of 256 voxels staged in LDS.
// Staging uses UNCONDITIONAL clamped loads (27 input taps + the voxel's 32-channel dY row as 16-byte vectors) so that all
// loads of a chunk are in flight together; dys rows are padded to 36 floats: conflict-free float4 writes and reads.
#define STEM_DYS_STRIDE 36
template <typename T> __device__ __forceinline__ void load_row32(const T* p, float* v) {      // 16-bit types
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 u = reinterpret_cast<const uint4*>(p)[i];
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[8 * i + 2 * k] = H16<T>::lo(w[k]); v[8 * i + 2 * k + 1] = H16<T>::hi(w[k]); }
    }
}
template <> __device__ __forceinline__ void load_row32<float>(const float* p, float* v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float4 f = reinterpret_cast<const float4*>(p)[i]; v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
}

template <typename T>
__global__ __launch_bounds__(256) void k_stem_wgrad(const StemArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_stem[];
    float* xs = reinterpret_cast<float*>(smem_stem);                 // [27][257]: taps land in different banks
    float* dys = xs + 27 * 257 + 1;                                   // [256][36] fp32 (16-byte aligned: 27*257+1 = 6940 floats)
    const int taps = A.k[0] * A.k[1] * A.k[2];
    const int c0 = blockIdx.y * 32;
    const int slot = threadIdx.x >> 3, cg = threadIdx.x & 7;          // slot = static (kd, kh, kw) in 3^3
    const int skd = slot / 9, skh = (slot / 3) % 3, skw = slot % 3;
    const bool active = slot < 27 && skd < A.k[0] && skh < A.k[1] && skw < A.k[2];
    const int tap = (skd * A.k[1] + skh) * A.k[2] + skw;              // index into the weight's flattened kernel volume
    // (parity path: the 256 products of a chunk are summed in fp32, the chunks of this workgroup in fp64 -- one fp32 chain over
    // the ~150 chunks x 256 voxels a workgroup walks at 160x160x96 x batch 4 was 2e-3 of the largest element off the reference,
    // tests/test_parity_full_gpu.py::test_luna160_b4_fp32_vs_reference_golden)
    double A0 = 0.0, A1 = 0.0, A2 = 0.0, A3 = 0.0;
    const int64_t nchunks = (A.total + 255) / 256;
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const int64_t v = ch * 256 + threadIdx.x;
        const bool valid = v < A.total;
        int64_t t2 = valid ? v : 0;
        const int ow = (int)(t2 % A.O[2]); t2 /= A.O[2];
        const int oh = (int)(t2 % A.O[1]); t2 /= A.O[1];
        const int od = (int)(t2 % A.O[0]);
        const int n = (int)(t2 / A.O[0]);
        const T* xn = reinterpret_cast<const T*>(A.x) + (int64_t)n * A.I[0] * A.I[1] * A.I[2];
        float xv[27];     // static slots (kd, kh, kw) in 3^3; kernels smaller than 3 leave slots inactive (zero)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int id = od * A.s[0] - A.p[0] + kd, ih = oh * A.s[1] - A.p[1] + kh, iw = ow * A.s[2] - A.p[2] + kw;
                    const bool ok = valid && kd < A.k[0] && kh < A.k[1] && kw < A.k[2] &&
                                    (unsigned)id < (unsigned)A.I[0] && (unsigned)ih < (unsigned)A.I[1] && (unsigned)iw < (unsigned)A.I[2];
                    const float raw = Elem<T>::ld(xn[ok ? ((int64_t)id * A.I[1] + ih) * A.I[2] + iw : 0]);
                    xv[kd * 9 + kh * 3 + kw] = ok ? raw : 0.f;
                }
        float dv[32];
        load_row32<T>(reinterpret_cast<const T*>(A.dy) + (valid ? v : 0) * A.Cy + c0, dv);
        __syncthreads();                        // previous chunk's compute is done with the LDS buffers
#pragma unroll
        for (int k = 0; k < 27; ++k) xs[k * 257 + threadIdx.x] = xv[k];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            *reinterpret_cast<float4*>(dys + threadIdx.x * STEM_DYS_STRIDE + 4 * k) =
                valid ? make_float4(dv[4 * k], dv[4 * k + 1], dv[4 * k + 2], dv[4 * k + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        if (active) {
            const float* xr = xs + slot * 257;
            const float* dr = dys + cg * 4;
#pragma unroll 8
            for (int i = 0; i < 256; ++i) {
                const float x1 = xr[i];
                const float4 d = *reinterpret_cast<const float4*>(dr + i * STEM_DYS_STRIDE);
                a0 = fmaf(x1, d.x, a0); a1 = fmaf(x1, d.y, a1); a2 = fmaf(x1, d.z, a2); a3 = fmaf(x1, d.w, a3);
            }
            A0 += (double)a0; A1 += (double)a1; A2 += (double)a2; A3 += (double)a3;
        }
    }
    if (active) {
        const int c = c0 + cg * 4;
        if (c + 0 < A.cout) atomicAdd(A.dw + (int64_t)(c + 0) * taps + tap, (float)A0);
        if (c + 1 < A.cout) atomicAdd(A.dw + (int64_t)(c + 1) * taps + tap, (float)A1);
        if (c + 2 < A.cout) atomicAdd(A.dw + (int64_t)(c + 2) * taps + tap, (float)A2);
        if (c + 3 < A.cout) atomicAdd(A.dw + (int64_t)(c + 3) * taps + tap, (float)A3);
    }
}

// ------------------------------------------------------------------------------------------------ bf16, 3x3x3, stride 1
// MFMA formulation of the stem weight gradient: dW[r][tap] = sum_p dY[p][r] * X[p + tap] is a GEMM with M = 32 output
// channels, N = 27 (padded to 32) taps and K = lattice points once the input patch is expanded ("im2col") into a
// [point][32 taps] tile -- which is cheap here because the input has ONE channel: the halo of a 4 x 8 x 8 point tile is
// 600 bf16 values. Both operands are then read with the LDS transpose read exactly like the P / Q tiles of k_wgrad3.
// Per tile of 256 points: 16 KB of dY (4 buffer loads per thread), 1.2 KB of X, 27 ds_read_u16 + 4 ds_write_b128 per thread
// for the expansion, 8 MFMAs per wave. The VALU kernel above needed 256 x (ds_read_b32 + ds_read_b128 + 4 FMA) per thread
// and chunk and was LDS-bound at 0.9 ms (profiles/round1_v6_kernel_stats_by_grid.txt); this one is HBM-bound on dY.
typedef short st_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) st_s16x4* st_lds_s16x4_ptr;
__device__ __forceinline__ u32x4 stem_trfrag(const char* p) {   // 8 points (rows p, p + 4 voxels) of channel li, see WF<bf16_t>
    const st_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_lds_s16x4_ptr)(p));
    const st_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_lds_s16x4_ptr)(p + 4 * 64));
    const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
    return u32x4{ua.x, ua.y, ub.x, ub.y};
}

template <typename T>      // bf16_t / f16_t: raw 16-bit moves, the MFMA of the type
__global__ __launch_bounds__(256, 2) void k_stem_wgrad3(const StemArgs A, int nt0, int nt1, int nt2, int total_tiles) {
    constexpr int PROW = 8 * 64 + 32;                  // bytes per row of 8 points (64 B per point + bank padding)
    constexpr int TILE = 32 * PROW;                    // 17408
    __shared__ __attribute__((aligned(16))) char dyt[TILE];
    __shared__ __attribute__((aligned(16))) char imt[TILE];
    __shared__ __attribute__((aligned(16))) uint16_t xh[608];
    __shared__ float red[32 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int c0 = blockIdx.y * 32;
    for (int i = tid; i < 1024; i += 256) red[i] = 0.f;

    // dY staging geometry: piece s of a thread = point (pd = s, ph = tid >> 5, pw = (tid >> 2) & 7), 16-byte part tid & 3
    const int d_ph = tid >> 5, d_pw = (tid >> 2) & 7;
    const int d_dst0 = d_ph * PROW + d_pw * 64 + (tid & 3) * 16;
    const int d_rel = (d_ph * A.O[2] + d_pw) * A.Cy * 2 + (tid & 3) * 16;
    const int d_slab = A.O[1] * A.O[2] * A.Cy * 2;
    const int dy_img = A.O[0] * d_slab, x_img = A.I[0] * A.I[1] * A.I[2] * 2;
    // X halo: values tid, tid + 256, tid + 512 of the 6 x 10 x 10 halo
    int x_rel[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int i = tid + s * 256;
        const int hd = i / 100, hh = (i / 10) % 10, hw = i % 10;
        x_rel[s] = (hd << 16) | (hh << 8) | hw;
    }
    // expansion: thread = point tid (pd = tid >> 6, ph = (tid >> 3) & 7, pw = tid & 7)
    const int e_base = ((tid >> 6) * 10 + ((tid >> 3) & 7)) * 10 + (tid & 7);
    char* const e_dst = imt + (tid >> 3) * PROW + (tid & 7) * 64;
    // fragments: wave wv takes the contraction steps 2 wv, 2 wv + 1 (rows of 8 points ks * 4 + q)
    const int f_lane = q * PROW + (li >> 2) * 64 + (li & 3) * 8;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tiles_per_n = nt0 * nt1 * nt2;
    u32x4 vd[4];
    uint16_t vx[3];
    auto issue = [&](int tile) {
        const int n = tile / tiles_per_n;
        int tt = tile - n * tiles_per_n;
        const int tw_i = tt % nt2; tt /= nt2;
        const int th_i = tt % nt1;
        const int td_i = tt / nt1;
        const int l0d = td_i * 4, l0h = th_i * 8, l0w = tw_i * 8;
        const auto drs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(A.dy)) + (int64_t)n * dy_img + c0 * 2, 0, dy_img - c0 * 2, 0x00020000);
        const auto xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(A.x)) + (int64_t)n * x_img, 0, x_img, 0x00020000);
        const int d_org = ((l0d * A.O[1] + l0h) * A.O[2] + l0w) * A.Cy * 2;
        const bool okhw = (l0h + d_ph < A.O[1]) && (l0w + d_pw < A.O[2]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            vd[s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                drs, (okhw && l0d + s < A.O[0]) ? d_rel + s * d_slab : (int)0x80000000, d_org, 0));
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int id = l0d - 1 + (x_rel[s] >> 16), ih = l0h - 1 + ((x_rel[s] >> 8) & 255), iw = l0w - 1 + (x_rel[s] & 255);
            const bool ok = (tid + s * 256 < 600) && (unsigned)id < (unsigned)A.I[0] && (unsigned)ih < (unsigned)A.I[1] && (unsigned)iw < (unsigned)A.I[2];
            vx[s] = __builtin_amdgcn_raw_buffer_load_b16(xrs, ok ? ((id * A.I[1] + ih) * A.I[2] + iw) * 2 : (int)0x80000000, 0, 0);
        }
    };
    int tile = blockIdx.x;
    if (tile < total_tiles) issue(tile);
    for (; tile < total_tiles; tile += gridDim.x) {
        __syncthreads();                                   // the MFMA phase of the previous tile is done with the LDS tiles
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<u32x4*>(dyt + d_dst0 + s * 8 * PROW) = vd[s];
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (tid + s * 256 < 600) xh[tid + s * 256] = vx[s];
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < total_tiles) issue(next);               // in flight during the expansion + MFMA phase
        uint32_t pk[16];
#pragma unroll
        for (int t = 0; t < 32; t += 2) {
            uint32_t lo = 0, hi = 0;
            if (t < 27) lo = xh[e_base + (t / 9) * 100 + ((t / 3) % 3) * 10 + t % 3];
            if (t + 1 < 27) hi = xh[e_base + ((t + 1) / 9) * 100 + (((t + 1) / 3) % 3) * 10 + (t + 1) % 3];
            pk[t >> 1] = lo | (hi << 16);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(e_dst + k * 16) = u32x4{pk[4 * k], pk[4 * k + 1], pk[4 * k + 2], pk[4 * k + 3]};
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ks = wv * 2 + kk;
            u32x4 pf[2], qf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                pf[i] = stem_trfrag(dyt + f_lane + ks * 4 * PROW + i * 32);
                qf[i] = stem_trfrag(imt + f_lane + ks * 4 * PROW + i * 32);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = H16<T>::mma(pf[i], qf[j], acc[i][j]);
        }
    }
    // workgroup reduction in LDS, then one atomic per (channel, tap)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) atomicAdd(&red[(i * 16 + q * 4 + rr) * 32 + j * 16 + li], acc[i][j][rr]);
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) {
        const int r = i >> 5, t = i & 31;
        if (t < 27 && c0 + r < A.cout) atomicAdd(A.dw + (int64_t)(c0 + r) * 27 + t, red[i]);
    }
}

// ------------------------------------------------------------------------------------------------ forward on MFMA
// Same expansion as k_stem_wgrad3: y[p][r] = sum_tap X[p + tap] * W[r][tap] with the [point][32 taps] tile as the B operand
// (one plain ds_read_b128 per 16 points and K = 32 step) and the 32 x 27 weights as A fragments that live in registers for the
// whole kernel. The C layout gives every lane 4 consecutive channels of one voxel (8-byte stores) and the InstanceNorm
// statistics of the rounded outputs are reduced in the epilogue like in k_ig3 (the separate 629 MB statistics pass of the
// VALU kernel disappears). grid (S, Cout_p / 32, N): a workgroup walks tiles of ONE image so its statistics flush once.
__device__ __forceinline__ float stem_dpp_row_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

// Hand-over point between two phases of ONE wave that exchange data through LDS (lane i writes what lane j reads): the LDS queue of a wave
// is in order, so no s_barrier is needed -- only the compiler has to keep the accesses on their side (and the data has to have landed).
__device__ __forceinline__ void stem_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// MODE 0: y = conv(x) (+ bias), stored rounded to T, optional statistics of the rounded values (the stand-alone convolution).
// MODE 1 / 2 = the two passes of the FUSED stem block conv -> InstanceNorm -> ReLU (nndet_stem_block_forward): the convolution is so
// cheap (27 MACs per output, one input channel) that computing it twice beats writing the pre-norm tensor and reading it back:
//   MODE 1: statistics of the (unrounded fp32) conv outputs only -- reads 2 bytes per voxel, writes nothing;
//   MODE 2: recompute, y * scale + shift (norm coefficients per image and channel, `coef` = mean_rstd [N][Cy][2]), ReLU, store.
// The pre-norm activation (629 MB at 160x160x96x32 bf16, batch 4) never exists in HBM.
template <typename T, int MODE = 0>
__global__ __launch_bounds__(256, 2) void k_stem_fwd3(const StemArgs A, int nt0, int nt1, int nt2, double* __restrict__ stats,
                                                      const float* __restrict__ coef = nullptr, const float* __restrict__ gamma = nullptr,
                                                      const float* __restrict__ beta = nullptr, int relu = 0) {
    constexpr int PROW = 8 * 64 + 32;
    __shared__ __attribute__((aligned(16))) char imt[32 * PROW];
    __shared__ __attribute__((aligned(16))) uint16_t xh_all[4 * 304];
    __shared__ double red[64];
    // Round 5, the recipe of k_stem_bwd3 (profiles/round5_stem_bwd_pmc.txt: these kernels are bound by VALU issue, and their 16-byte expansion
    // stores replayed bank conflicts for 47 % of the LDS cycles): NO workgroup barrier in the tile loop -- wave wv owns d-plane wv of every
    // tile (its own 3 x 10 x 10 halo, its own 64 rows of the expanded tile, the 64 points its MFMAs consume); slot h (8 bytes = 4 taps) of the
    // point in column c lives at slot h ^ (c >> 1); tiles inside the volume / with their halo inside (wave-uniform tests) skip the per-lane
    // bounds arithmetic and address their outputs by a scalar tile base + a lane constant.
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint16_t* const xh = xh_all + wv * 304;
    const int li = lane & 15, q = lane >> 4;
    const int c0 = blockIdx.y * 32, n = blockIdx.z;
    if (tid < 64) red[tid] = 0.0;
    // A fragments: W[c0 + i*16 + li][taps 8q .. 8q+7] rounded to bf16 (taps >= 27 are zero)
    u32x4 af[2];
    float bia[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = c0 + i * 16 + li;
        float wv8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = q * 8 + j;
            wv8[j] = (t < 27 && ch < A.cout) ? A.w[(int64_t)ch * 27 + t] : 0.f;
        }
        af[i] = u32x4{H16<T>::pack2(wv8[0], wv8[1]), H16<T>::pack2(wv8[2], wv8[3]), H16<T>::pack2(wv8[4], wv8[5]), H16<T>::pack2(wv8[6], wv8[7])};
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int cb = c0 + i * 16 + q * 4 + rr;
            bia[i][rr] = (A.bias && cb < A.cout) ? A.bias[cb] : 0.f;
        }
    }
    float nsc[2][4], nsh[2][4];                          // MODE 2: the same two expressions k_norm_apply evaluates per channel
    if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int cb = c0 + i * 16 + q * 4 + rr;
                float a_ = 0.f, b_ = 0.f;
                if (cb < A.cout) {
                    const float mean = coef[((int64_t)n * A.Cy + cb) * 2], rstd = coef[((int64_t)n * A.Cy + cb) * 2 + 1];
                    a_ = rstd * gamma[cb];
                    b_ = beta[cb] - mean * a_;
                }
                nsc[i][rr] = a_; nsh[i][rr] = b_;
            }
    }
    int x_rel[5], x_off[5];                                     // this wave's halo: planes wv .. wv + 2 of the tile's 6 x 10 x 10
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int i = lane + s * 64;
        const int rd = wv + i / 100, rh = (i / 10) % 10, rw = i % 10;
        x_rel[s] = i < 300 ? ((rd << 16) | (rh << 8) | rw) : 0x7f7f7f7f;       // (no such halo voxel: fails every bounds check)
        x_off[s] = ((rd * A.I[1] + rh) * A.I[2] + rw) * 2;                    // byte offset relative to the halo's first voxel
    }
    const int e_base = ((tid >> 3) & 7) * 10 + (tid & 7);                     // (relative to this wave's halo planes)
    char* const e_dst = imt + (tid >> 3) * PROW + (tid & 7) * 64;
    const int e_m = (tid & 7) >> 1;
    // MFMA phase: lane = (K chunk / channel quad q, point li = row li >> 3, column li & 7 of the wave's 16-point group j)
    const int r_m = (li >> 1) & 3;
    const int r_pt = (wv * 8 + (li >> 3)) * PROW + (li & 7) * 64;
    const int r_bfa = r_pt + (((2 * q) ^ r_m) * 8), r_bfb = r_pt + (((2 * q + 1) ^ r_m) * 8);
    int y_rel[4];                                               // output byte offset of point (j, li) relative to the tile's first voxel, channel quad q
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = (wv * 4 + j) * 16 + li;
        y_rel[j] = ((((p >> 6) * A.O[1] + ((p >> 3) & 7)) * A.O[2] + (p & 7)) * A.Cy + q * 4) * 2;
    }
    const int x_img = A.I[0] * A.I[1] * A.I[2] * 2;
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A.x)) + (int64_t)n * x_img, 0, x_img, 0x00020000);
    const int y_img = A.O[0] * A.O[1] * A.O[2] * A.Cy * 2;
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(A.y) + (int64_t)n * y_img + c0 * 2, 0, y_img - c0 * 2, 0x00020000);
    const int tiles_per_n = nt0 * nt1 * nt2;
    float ssum[2][4], ssq[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { ssum[i][rr] = 0.f; ssq[i][rr] = 0.f; }
    uint16_t vx[5];
    auto issue = [&](int tile, int& l0d, int& l0h, int& l0w) {
        int tt = tile;
        const int tw_i = tt % nt2; tt /= nt2;
        const int th_i = tt % nt1;
        const int td_i = tt / nt1;
        l0d = td_i * 4; l0h = th_i * 8; l0w = tw_i * 8;
        const int x_org = (((l0d - 1) * A.I[1] + (l0h - 1)) * A.I[2] + (l0w - 1)) * 2;       // (scalar; may be negative: added to the lane offset)
        const bool inner = l0d >= 1 && l0h >= 1 && l0w >= 1 && l0d + 5 <= A.I[0] && l0h + 9 <= A.I[1] && l0w + 9 <= A.I[2];
        if (inner) {
#pragma unroll
            for (int s = 0; s < 5; ++s)
                vx[s] = __builtin_amdgcn_raw_buffer_load_b16(xrs, (s < 4 || lane < 300 - 256) ? x_org + x_off[s] : (int)0x80000000, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int id = l0d - 1 + (x_rel[s] >> 16), ih = l0h - 1 + ((x_rel[s] >> 8) & 255), iw = l0w - 1 + (x_rel[s] & 255);
                const bool ok = (unsigned)id < (unsigned)A.I[0] && (unsigned)ih < (unsigned)A.I[1] && (unsigned)iw < (unsigned)A.I[2];
                vx[s] = __builtin_amdgcn_raw_buffer_load_b16(xrs, ok ? x_org + x_off[s] : (int)0x80000000, 0, 0);
            }
        }
    };
    int tile = xcd_compact(blockIdx.x, gridDim.x, gridDim.x);     // neighbouring tiles (shared halo rows) on one XCD
    int l0d = 0, l0h = 0, l0w = 0, n0d, n0h, n0w;
    if (tile < tiles_per_n) issue(tile, l0d, l0h, l0w);
    for (; tile < tiles_per_n; tile += gridDim.x) {
        // (this wave's MFMA phase of the previous tile has issued its LDS reads: the stores below queue behind them)
#pragma unroll
        for (int s = 0; s < 5; ++s)
            if (lane + s * 64 < 300) xh[lane + s * 64] = vx[s];
        stem_wave_lds_sync();
        const int next = tile + gridDim.x;
        if (next < tiles_per_n) issue(next, n0d, n0h, n0w);
        uint32_t pk[16];
#pragma unroll
        for (int t = 0; t < 32; t += 2) {
            uint32_t lo = 0, hi = 0;
            if (t < 27) lo = xh[e_base + (t / 9) * 100 + ((t / 3) % 3) * 10 + t % 3];
            if (t + 1 < 27) hi = xh[e_base + ((t + 1) / 9) * 100 + (((t + 1) / 3) % 3) * 10 + (t + 1) % 3];
            pk[t >> 1] = lo | (hi << 16);
        }
#pragma unroll
        for (int h = 0; h < 8; ++h) *reinterpret_cast<uint2*>(e_dst + ((h ^ e_m) * 8)) = uint2{pk[2 * h], pk[2 * h + 1]};
        stem_wave_lds_sync();
        u32x4 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                      // 16 points = rows 2 j, 2 j + 1 of the wave's plane
            const uint2 ba = *reinterpret_cast<const uint2*>(imt + j * 2 * PROW + r_bfa);
            const uint2 bb = *reinterpret_cast<const uint2*>(imt + j * 2 * PROW + r_bfb);
            bf[j] = u32x4{ba.x, ba.y, bb.x, bb.y};
        }
        stem_wave_lds_sync();                              // (the next tile's stores stay behind these reads)
        const bool full = l0d + 4 <= A.O[0] && l0h + 8 <= A.O[1] && l0w + 8 <= A.O[2];         // wave-uniform
        const int y_org = ((l0d * A.O[1] + l0h) * A.O[2] + l0w) * A.Cy * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool valid = true;
            int vo = y_rel[j];
            if (!full) {
                const int p = (wv * 4 + j) * 16 + li;
                valid = (l0d + (p >> 6) < A.O[0]) && (l0h + ((p >> 3) & 7) < A.O[1]) && (l0w + (p & 7) < A.O[2]);
                vo = valid ? vo : (int)0x80000000;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 c = f32x4{bia[i][0], bia[i][1], bia[i][2], bia[i][3]};
                c = H16<T>::mma(af[i], bf[j], c);
                typedef unsigned int v2u_t __attribute__((__vector_size__(8)));
                if constexpr (MODE == 1) {               // statistics of the fp32 outputs, nothing stored
                    if (valid) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) { ssum[i][rr] += c[rr]; ssq[i][rr] += c[rr] * c[rr]; }
                    }
                    continue;
                }
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        float v_ = fmaf(c[rr], nsc[i][rr], nsh[i][rr]);
                        c[rr] = relu ? fmaxf(v_, 0.f) : v_;
                    }
                }
                v2u_t o;
                o[0] = H16<T>::pack2(c[0], c[1]); o[1] = H16<T>::pack2(c[2], c[3]);
                __builtin_amdgcn_raw_buffer_store_b64(o, yrs, vo + i * 32, y_org, 0);
                if (MODE == 0 && stats && valid) {
                    const float r0 = H16<T>::lo(o[0]), r1 = H16<T>::hi(o[0]);
                    const float r2 = H16<T>::lo(o[1]), r3 = H16<T>::hi(o[1]);
                    ssum[i][0] += r0; ssum[i][1] += r1; ssum[i][2] += r2; ssum[i][3] += r3;
                    ssq[i][0] += r0 * r0; ssq[i][1] += r1 * r1; ssq[i][2] += r2 * r2; ssq[i][3] += r3 * r3;
                }
            }
        }
        l0d = n0d; l0h = n0h; l0w = n0w;
    }
    if (stats) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float s = stem_dpp_row_sum(ssum[i][rr]), s2 = stem_dpp_row_sum(ssq[i][rr]);
                if (li == 0) {
                    atomicAdd(&red[(i * 16 + q * 4 + rr) * 2 + 0], (double)s);
                    atomicAdd(&red[(i * 16 + q * 4 + rr) * 2 + 1], (double)s2);
                }
            }
        __syncthreads();
        if (tid < 64) {
            const int rep = blockIdx.x % NNDET_STATS_REPLICAS;
            atomicAdd(stats + (((int64_t)rep * A.N + n) * A.Cy + c0 + (tid >> 1)) * 2 + (tid & 1), red[tid]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ fused stem block, backward
// conv(1 -> C) -> InstanceNorm -> ReLU with the gradient dA w.r.t. the block OUTPUT as the only large input. The input image needs no
// gradient, so all the block has to deliver is dW [C][27], dgamma [C], dbeta [C]. With g = dA * [z > 0], xh = (y - mean) * rstd:
//     dL/dy = rstd * gamma * (g - S1 / M - xh * S2 / M),   S1 = sum_p g, S2 = sum_p g * xh  (per image and channel, M voxels)
//     dW[c][t] = sum_n sum_p dL/dy[n,p,c] * x[n, p + t]
//              = sum_n rstd * gamma * (A_n[c][t] - S1 / M * B_n[t] - S2 / M * C_n[c][t])
// with A_n = sum_p g (x) x_shifted, B_n = sum_p x_shifted, C_n = sum_p xh (x) x_shifted: three correlations that do NOT need S1 / S2
// in advance. So ONE pass over dA computes S1, S2, A, B, C together -- the pre-norm activation y is recomputed from the one-channel
// image in LDS (one more 32 x 32 x 32 MFMA per 16 points) -- and a 1-block kernel combines them. This replaces k_norm_bwd_reduce
// (reads y, dA), k_norm_bwd_apply (reads y, dA, writes dy) and k_stem_wgrad3 (reads dy): 6 passes over a 629 MB tensor become 1, at
// the very end of the backward pass where nothing else is left to overlap with (round 3 timeline: 1.25 ms -> one 0.3 ms launch).
struct StemBwdArgs {
    const void* x; const void* dy; const float* w; const float* coef; const float* gamma; const float* beta;
    float* accA; float* accC; float* accB; double* accS;     // [N][Cy][32], [N][Cy][32], [N][32], [N][Cy][2]   (zeroed)
    int32_t N, I[3], O[3], Cy, cout, relu;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void k_stem_bwd3(const StemBwdArgs A, int nt0, int nt1, int nt2) {
    constexpr int PROW = 8 * 64 + 32;                  // bytes per row of 8 points (64 B per point + bank padding)
    constexpr int TILE = 32 * PROW;
    __shared__ __attribute__((aligned(16))) char dyt[TILE];     // dA tile, rewritten in place with g = dA * mask
    __shared__ __attribute__((aligned(16))) char imt[TILE];     // [point][32 taps] expansion of the image halo
    __shared__ __attribute__((aligned(16))) char xnt[TILE];     // xh = normalised pre-activation, [point][32 channels]
    // Exactly 3 tiles = 51 KB of LDS (round 5; 60.8 KB before): at the end of the backward pass this kernel runs beside k_wgrad3d of the
    // weight-gradient stream, whose persistent workgroups hold 108 KB on EVERY CU -- with 60.8 KB no workgroup of this kernel fitted next
    // to one of those (160 KB per CU), the two kernels shared the chip CU by CU instead of wave by wave. The small buffers alias tiles
    // that are dead while they live: the image halo `xh` (tile start -> expansion) sits in xnt (written by the recompute phase, read
    // until the next tile's first barrier), the final reduction buffers in dyt / imt behind the tile loop.
    //
    // NO workgroup barrier inside the tile loop (round 5; 4 per tile before): wave wv owns d-plane wv of every tile -- 64 points = rows
    // wv * 8 .. wv * 8 + 7 of the three LDS tiles. It stages its own plane of dA and its own 3 x 10 x 10 halo of the image, expands, recomputes
    // and contracts exactly these points (the contraction steps 2 wv, 2 wv + 1 ARE these 64 points), so every LDS dependency is wave-local
    // and ordered by the in-order LDS queue of the wave; the four waves drift apart freely and hide each other's load / LDS latencies.
    //
    // Bank swizzle (round 5): a point's 64 bytes are eight 8-byte slots (4 channels / 4 taps each); slot h of the point in column c of a row
    // lives at slot h ^ (c >> 1). Unswizzled, the 8-byte accesses of the recompute phase (columns c and c + 4 are 256 B apart = the same
    // bank) and the 16-byte stores of the expansion were 2- to 4-way conflicts: SQ_LDS_BANK_CONFLICT was 51 % of SQ_LDS_IDX_ACTIVE, the LDS
    // 66 % busy (profiles/round5_stem_bwd_pmc.txt). Every access below goes through the same map; the expansion stores and the B-fragment
    // loads of the recompute MFMA are 8-byte accesses now (same LDS cycles as the 16-byte forms).
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint16_t* const xh = reinterpret_cast<uint16_t*>(xnt + wv * 8 * PROW);     // [300], inside this wave's own rows of xnt
    float* const red = reinterpret_cast<float*>(dyt);            // [2 * 32 * 32 + 32]
    double* const reds = reinterpret_cast<double*>(imt);         // [64]
    static_assert((2 * 32 * 32 + 32) * 4 <= TILE && 300 * 2 <= 8 * PROW, "aliased buffers fit");
    const int li = lane & 15, q = lane >> 4;
    const int c0 = blockIdx.y * 32, n = blockIdx.z;
    // A fragments of the forward convolution (identical to k_stem_fwd3: the recomputed y is bit-identical to the forward pass's)
    u32x4 af[2];
    float n_sc[2][4], n_sh[2][4], n_rs[2][4], n_mrs[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = c0 + i * 16 + li;
        float wv8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = q * 8 + j;
            wv8[j] = (t < 27 && ch < A.cout) ? A.w[(int64_t)ch * 27 + t] : 0.f;
        }
        af[i] = u32x4{H16<T>::pack2(wv8[0], wv8[1]), H16<T>::pack2(wv8[2], wv8[3]), H16<T>::pack2(wv8[4], wv8[5]), H16<T>::pack2(wv8[6], wv8[7])};
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int cb = c0 + i * 16 + q * 4 + rr;
            float a_ = 0.f, b_ = 0.f, rs = 0.f, mrs = 0.f;
            if (cb < A.cout) {
                const float mean = A.coef[((int64_t)n * A.Cy + cb) * 2], rstd = A.coef[((int64_t)n * A.Cy + cb) * 2 + 1];
                a_ = rstd * A.gamma[cb];
                b_ = A.beta[cb] - mean * a_;
                rs = rstd; mrs = mean * rstd;
            }
            n_sc[i][rr] = a_; n_sh[i][rr] = b_; n_rs[i][rr] = rs; n_mrs[i][rr] = mrs;
        }
    }
    // dA staging geometry: piece k of a lane = point (pd = wv, ph = 2 k + (lane >> 5), pw = (lane >> 2) & 7), 16-byte part lane & 3
    const int d_ph = lane >> 5, d_pw = (lane >> 2) & 7;
    const int d_m = d_pw >> 1;                                   // swizzle of this lane's point: the two 8-byte halves of its 16-byte part go to slots (2 k) ^ m, (2 k + 1) ^ m
    const int d_pt = (wv * 8 + d_ph) * PROW + d_pw * 64;
    const int d_dst0 = d_pt + (((2 * (lane & 3)) ^ d_m) * 8), d_dst1 = d_pt + (((2 * (lane & 3) + 1) ^ d_m) * 8);
    const int d_slab = A.O[1] * A.O[2] * A.Cy * 2;
    const int d_rel = wv * d_slab + (d_ph * A.O[2] + d_pw) * A.Cy * 2 + (lane & 3) * 16;
    const int d_step = 2 * A.O[2] * A.Cy * 2;                   // two h-rows further
    const int dy_img = A.O[0] * d_slab, x_img = A.I[0] * A.I[1] * A.I[2] * 2;
    const auto drs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(A.dy)) + (int64_t)n * dy_img + c0 * 2, 0, dy_img - c0 * 2, 0x00020000);
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(A.x)) + (int64_t)n * x_img, 0, x_img, 0x00020000);
    int x_rel[5], x_off[5];                                     // this wave's halo: planes wv .. wv + 2 of the tile's 6 x 10 x 10
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int i = lane + s * 64;
        const int rd = wv + i / 100, rh = (i / 10) % 10, rw = i % 10;
        x_rel[s] = i < 300 ? ((rd << 16) | (rh << 8) | rw) : 0x7f7f7f7f;       // (no such halo voxel: fails every bounds check below)
        x_off[s] = ((rd * A.I[1] + rh) * A.I[2] + rw) * 2;                    // byte offset relative to the halo's first voxel
    }
    const int e_pd = tid >> 6, e_ph = (tid >> 3) & 7, e_pw = tid & 7;       // expansion: thread = point tid (plane e_pd = wv)
    const int e_base = e_ph * 10 + e_pw;                                    // (relative to this wave's halo planes)
    char* const e_dst = imt + (tid >> 3) * PROW + (tid & 7) * 64;
    const int e_m = (tid & 7) >> 1;
    // transposed fragment reads (see k_stem_wgrad3): lane = (point column li >> 2 [+ 4 for the second read], slot (li & 3) + 4 i)
    int f_a[2], f_b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int fc = li >> 2, fh = (li & 3) + 4 * i;
        f_a[i] = (wv * 8 + q) * PROW + fc * 64 + ((fh ^ (fc >> 1)) * 8);              // (this wave's rows: offsets inside the plane are immediates)
        f_b[i] = (wv * 8 + q) * PROW + (fc + 4) * 64 + ((fh ^ ((fc + 4) >> 1)) * 8);
    }
    auto trfrag = [&](const char* tile_rows, int i) -> u32x4 {      // stem_trfrag through the swizzle
        const st_s16x4 a_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_lds_s16x4_ptr)(tile_rows + f_a[i]));
        const st_s16x4 b_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_lds_s16x4_ptr)(tile_rows + f_b[i]));
        const uint2 ua = __builtin_bit_cast(uint2, a_), ub = __builtin_bit_cast(uint2, b_);
        return u32x4{ua.x, ua.y, ub.x, ub.y};
    };
    // recompute phase: lane = (K chunk / channel quad q, point li = row li >> 3, column li & 7 of the wave's 16-point group)
    const int r_m = (li >> 1) & 3;
    const int r_pt = (wv * 8 + (li >> 3)) * PROW + (li & 7) * 64;
    const int r_bfa = r_pt + (((2 * q) ^ r_m) * 8), r_bfb = r_pt + (((2 * q + 1) ^ r_m) * 8);   // the two tap halves of chunk q (these two reads keep a
    // 2-way conflict between the lanes q and q ^ 1 of a half wave: reading the halves in a lane-dependent order avoids it for 16 selects per plane -- dearer)
    const int r_ch[2] = {r_pt + ((q ^ r_m) * 8), r_pt + (((q + 4) ^ r_m) * 8)};                                     // channel quad q of block i

    f32x4 accA[2][2], accC[2][2], accB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        accB[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) { accA[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accC[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    float s1[2][4], s2[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { s1[i][rr] = 0.f; s2[i][rr] = 0.f; }
    const u32x4 ones = u32x4{H16<T>::ONE2, H16<T>::ONE2, H16<T>::ONE2, H16<T>::ONE2};
    const float relu_floor = A.relu ? 0.f : -__builtin_inff();
    const int tiles_per_n = nt0 * nt1 * nt2;
    u32x4 vd[4];
    uint16_t vx[5];
    auto issue = [&](int tile, int& l0d, int& l0h, int& l0w) {
        int tt = tile;
        const int tw_i = tt % nt2; tt /= nt2;
        const int th_i = tt % nt1;
        const int td_i = tt / nt1;
        l0d = td_i * 4; l0h = th_i * 8; l0w = tw_i * 8;
        const int d_org = ((l0d * A.O[1] + l0h) * A.O[2] + l0w) * A.Cy * 2;
        // The kernel is bound by VALU issue (2 waves per SIMD issue 90 % of the cycles, profiles/round5_stem_bwd_pmc.txt): tiles that lie
        // completely inside the volume / whose halo does (wave-uniform tests) skip the per-lane bounds arithmetic.
        const bool full = l0d + 4 <= A.O[0] && l0h + 8 <= A.O[1] && l0w + 8 <= A.O[2];
        if (full) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                vd[s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, d_rel + s * d_step, d_org, 0));
        } else {
            const bool okdw = (l0d + wv < A.O[0]) && (l0w + d_pw < A.O[2]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                vd[s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    drs, (okdw && l0h + 2 * s + d_ph < A.O[1]) ? d_rel + s * d_step : (int)0x80000000, d_org, 0));
        }
        const int x_org = (((l0d - 1) * A.I[1] + (l0h - 1)) * A.I[2] + (l0w - 1)) * 2;       // (scalar; may be negative: added to the lane offset below)
        const bool inner = l0d >= 1 && l0h >= 1 && l0w >= 1 && l0d + 5 <= A.I[0] && l0h + 9 <= A.I[1] && l0w + 9 <= A.I[2];
        if (inner) {
#pragma unroll
            for (int s = 0; s < 5; ++s)
                vx[s] = __builtin_amdgcn_raw_buffer_load_b16(xrs, (s < 4 || lane < 300 - 256) ? x_org + x_off[s] : (int)0x80000000, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int id = l0d - 1 + (x_rel[s] >> 16), ih = l0h - 1 + ((x_rel[s] >> 8) & 255), iw = l0w - 1 + (x_rel[s] & 255);
                const bool ok = (unsigned)id < (unsigned)A.I[0] && (unsigned)ih < (unsigned)A.I[1] && (unsigned)iw < (unsigned)A.I[2];
                vx[s] = __builtin_amdgcn_raw_buffer_load_b16(xrs, ok ? x_org + x_off[s] : (int)0x80000000, 0, 0);
            }
        }
    };
    int tile = xcd_compact(blockIdx.x, gridDim.x, gridDim.x);
    int l0d = 0, l0h = 0, l0w = 0, n0d = 0, n0h = 0, n0w = 0;
    if (tile < tiles_per_n) issue(tile, l0d, l0h, l0w);
    for (; tile < tiles_per_n; tile += gridDim.x) {
        // (this wave's MFMA phase of the previous tile has issued its LDS reads: the writes below queue behind them)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            *reinterpret_cast<uint2*>(dyt + d_dst0 + s * 2 * PROW) = uint2{vd[s][0], vd[s][1]};
            *reinterpret_cast<uint2*>(dyt + d_dst1 + s * 2 * PROW) = uint2{vd[s][2], vd[s][3]};
        }
#pragma unroll
        for (int s = 0; s < 5; ++s)
            if (lane + s * 64 < 300) xh[lane + s * 64] = vx[s];
        stem_wave_lds_sync();
        const int next = tile + gridDim.x;
        if (next < tiles_per_n) issue(next, n0d, n0h, n0w);    // in flight during the rest of this tile
        {   // expansion; a point outside the volume gets a zero row (it must not contribute to B = sum of the shifted image)
            uint32_t pk[16];
#pragma unroll
            for (int t = 0; t < 32; t += 2) {
                uint32_t lo = 0, hi = 0;
                if (t < 27) lo = xh[e_base + (t / 9) * 100 + ((t / 3) % 3) * 10 + t % 3];
                if (t + 1 < 27) hi = xh[e_base + ((t + 1) / 9) * 100 + (((t + 1) / 3) % 3) * 10 + (t + 1) % 3];
                pk[t >> 1] = lo | (hi << 16);
            }
            if (!(l0d + 4 <= A.O[0] && l0h + 8 <= A.O[1] && l0w + 8 <= A.O[2])) {      // (wave-uniform: a tile that sticks out of the volume)
                const bool pv = (l0d + e_pd < A.O[0]) && (l0h + e_ph < A.O[1]) && (l0w + e_pw < A.O[2]);
#pragma unroll
                for (int t = 0; t < 16; ++t) pk[t] = pv ? pk[t] : 0u;
            }
#pragma unroll
            for (int h = 0; h < 8; ++h) *reinterpret_cast<uint2*>(e_dst + ((h ^ e_m) * 8)) = uint2{pk[2 * h], pk[2 * h + 1]};
        }
        stem_wave_lds_sync();
        {   // recompute y, mask the incoming gradient, normalise: g -> dyt (in place), xh -> xnt; S1 / S2 in registers
            u32x4 bf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint2 ba = *reinterpret_cast<const uint2*>(imt + j * 2 * PROW + r_bfa);
                const uint2 bb = *reinterpret_cast<const uint2*>(imt + j * 2 * PROW + r_bfb);
                bf[j] = u32x4{ba.x, ba.y, bb.x, bb.y};      // taps q * 8 .. q * 8 + 7 in order
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // (a point outside the volume needs no special case: its dA was staged as zero -> g = 0, and its row of the expanded image is
                //  zero -> y = 0, a finite xh that meets a zero row in the correlation C)
                constexpr int PR2 = 2 * PROW;
                const int prow = j * PR2;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
                    c = H16<T>::mma(af[i], bf[j], c);
                    const uint2 dv = *reinterpret_cast<const uint2*>(dyt + prow + r_ch[i]);
                    const float d4[4] = {H16<T>::lo(dv.x), H16<T>::hi(dv.x), H16<T>::lo(dv.y), H16<T>::hi(dv.y)};
                    float g4[4], x4[4];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float z = fmaf(c[rr], n_sc[i][rr], n_sh[i][rr]);          // the forward pass's expression: same ReLU decisions
                        const float xn = fmaf(c[rr], n_rs[i][rr], -n_mrs[i][rr]);
                        const float g = z > relu_floor ? d4[rr] : 0.f;                   // (relu_floor = -inf without ReLU)
                        g4[rr] = g; x4[rr] = xn;
                        s1[i][rr] += g; s2[i][rr] = fmaf(g, xn, s2[i][rr]);
                    }
                    uint2 go, xo;
                    go.x = H16<T>::pack2(g4[0], g4[1]); go.y = H16<T>::pack2(g4[2], g4[3]);
                    xo.x = H16<T>::pack2(x4[0], x4[1]); xo.y = H16<T>::pack2(x4[2], x4[3]);
                    *reinterpret_cast<uint2*>(dyt + prow + r_ch[i]) = go;
                    *reinterpret_cast<uint2*>(xnt + prow + r_ch[i]) = xo;
                }
            }
        }
        stem_wave_lds_sync();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {                   // wave wv takes the contraction steps 2 wv, 2 wv + 1 (32 points each) = its own plane
            const int ks = kk;                                 // (relative to the wave's rows, see f_a / f_b)
            u32x4 pf[2], xf[2], qf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                pf[i] = trfrag(dyt + ks * 4 * PROW, i);
                xf[i] = trfrag(xnt + ks * 4 * PROW, i);
                qf[i] = trfrag(imt + ks * 4 * PROW, i);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                accB[j] = H16<T>::mma(ones, qf[j], accB[j]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    accA[i][j] = H16<T>::mma(pf[i], qf[j], accA[i][j]);
                    accC[i][j] = H16<T>::mma(xf[i], qf[j], accC[i][j]);
                }
            }
        }
        stem_wave_lds_sync();                              // (the next tile's stores stay behind this tile's fragment reads)
        l0d = n0d; l0h = n0h; l0w = n0w;
    }
    // workgroup reduction in LDS (the tiles are dead now), then one atomic per value into the per-image accumulators
    __syncthreads();
    for (int i = tid; i < 2 * 32 * 32 + 32; i += 256) red[i] = 0.f;
    if (tid < 64) reds[tid] = 0.0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                atomicAdd(&red[(i * 16 + q * 4 + rr) * 32 + j * 16 + li], accA[i][j][rr]);
                atomicAdd(&red[1024 + (i * 16 + q * 4 + rr) * 32 + j * 16 + li], accC[i][j][rr]);
            }
    if (q == 0) {                                          // every row of the ones-product is the same: row 0 = lanes q == 0, element 0
        atomicAdd(&red[2048 + li], accB[0][0]);
        atomicAdd(&red[2048 + 16 + li], accB[1][0]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const float a_ = stem_dpp_row_sum(s1[i][rr]), b_ = stem_dpp_row_sum(s2[i][rr]);
            if (li == 0) {
                atomicAdd(&reds[(i * 16 + q * 4 + rr) * 2 + 0], (double)a_);
                atomicAdd(&reds[(i * 16 + q * 4 + rr) * 2 + 1], (double)b_);
            }
        }
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) {
        atomicAdd(A.accA + ((int64_t)n * A.Cy + c0) * 32 + i, red[i]);
        atomicAdd(A.accC + ((int64_t)n * A.Cy + c0) * 32 + i, red[1024 + i]);
    }
    if (tid < 32 && c0 == 0) atomicAdd(A.accB + (int64_t)n * 32 + tid, red[2048 + tid]);
    if (tid < 64) atomicAdd(A.accS + ((int64_t)n * A.Cy + c0) * 2 + tid, reds[tid]);
}

// dW, dgamma, dbeta from the per-image sums (see k_stem_bwd3). One block; thread = (channel, tap) pair.
__global__ void k_stem_bwd_finish(const StemBwdArgs A, float* __restrict__ dw, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const double M = (double)A.O[0] * A.O[1] * A.O[2];
    for (int i = threadIdx.x; i < A.cout * 27; i += blockDim.x) {
        const int c = i / 27, t = i - c * 27;
        double acc = 0.0;
        for (int n = 0; n < A.N; ++n) {
            const double rstd = A.coef[((int64_t)n * A.Cy + c) * 2 + 1];
            const double S1 = A.accS[((int64_t)n * A.Cy + c) * 2], S2 = A.accS[((int64_t)n * A.Cy + c) * 2 + 1];
            const double a_ = A.accA[((int64_t)n * A.Cy + c) * 32 + t], c_ = A.accC[((int64_t)n * A.Cy + c) * 32 + t];
            const double b_ = A.accB[(int64_t)n * 32 + t];
            acc += rstd * (double)A.gamma[c] * (a_ - S1 / M * b_ - S2 / M * c_);
        }
        dw[i] = (float)acc;
    }
    for (int c = threadIdx.x; c < A.cout; c += blockDim.x) {
        double g = 0.0, b = 0.0;
        for (int n = 0; n < A.N; ++n) { b += A.accS[((int64_t)n * A.Cy + c) * 2]; g += A.accS[((int64_t)n * A.Cy + c) * 2 + 1]; }
        dgamma[c] = (float)g; dbeta[c] = (float)b;
    }
}

static int stem_args(const NndetConv* c, StemArgs* a) {
    if (c->cin_p != 1 || c->cin != 1 || c->transposed || c->cout_p % 32) return NNDET_EINVAL;
    if (c->k[0] * c->k[1] * c->k[2] > 27) return NNDET_EINVAL;
    memset(a, 0, sizeof(*a));
    a->N = c->batch; a->Cy = c->cout_p; a->cout = c->cout;
    const int in_sp[3] = {c->in_d, c->in_h, c->in_w}, out_sp[3] = {c->out_d, c->out_h, c->out_w};
    for (int i = 0; i < 3; ++i) {
        a->I[i] = in_sp[i]; a->O[i] = out_sp[i]; a->k[i] = c->k[i]; a->s[i] = c->s[i]; a->p[i] = c->p[i];
        if (out_sp[i] != (in_sp[i] + 2 * c->p[i] - c->k[i]) / c->s[i] + 1) return NNDET_EINVAL;
    }
    a->total = (int64_t)c->batch * out_sp[0] * out_sp[1] * out_sp[2];
    return 0;
}

int stem_forward(const NndetConv* c, const void* x, const float* w_f32, const float* bias, void* y, double* stats, int* stats_done,
                 hipStream_t st) {
    StemArgs a;
    int rc = stem_args(c, &a);
    if (rc) return rc;
    a.x = x; a.w = w_f32; a.bias = bias; a.y = y;
    *stats_done = 0;
    const int64_t yb = (int64_t)a.O[0] * a.O[1] * a.O[2] * a.Cy * 2;
    if (nndet_is16(c->dtype) && c->k[0] == 3 && c->k[1] == 3 && c->k[2] == 3 && c->s[0] == 1 && c->s[1] == 1 && c->s[2] == 1 &&
        c->p[0] == 1 && c->p[1] == 1 && c->p[2] == 1 && yb < (1LL << 31) && !getenv("NNDET_STEM_VALU")) {
        const int nt0 = ceil_div(a.O[0], 4), nt1 = ceil_div(a.O[1], 8), nt2 = ceil_div(a.O[2], 8);
        const int64_t tiles = (int64_t)nt0 * nt1 * nt2;
        if (tiles < (1LL << 30)) {
            int S = 512 / (a.N * (c->cout_p / 32));
            if (S < 8) S = 8;
            S = (S / 8) * 8;
            if (S > tiles) S = (int)tiles;
            if (c->dtype == NNDET_F16) k_stem_fwd3<f16_t><<<dim3(S, c->cout_p / 32, a.N), 256, 0, st>>>(a, nt0, nt1, nt2, stats);
            else k_stem_fwd3<bf16_t><<<dim3(S, c->cout_p / 32, a.N), 256, 0, st>>>(a, nt0, nt1, nt2, stats);
            LAUNCH_CHECK();
            *stats_done = 1;
            return 0;
        }
    }
    dim3 grid((unsigned)ceil_div64(a.total, 256), c->cout_p / 32);
    if (c->dtype == NNDET_BF16) k_stem_fwd<bf16_t><<<grid, 256, 0, st>>>(a);
    else if (c->dtype == NNDET_F16) k_stem_fwd<f16_t><<<grid, 256, 0, st>>>(a);
    else k_stem_fwd<float><<<grid, 256, 0, st>>>(a);
    LAUNCH_CHECK();
    return 0;
}

int stem_wgrad(const NndetConv* c, const void* x, const void* dy, float* dw, hipStream_t st) {
    StemArgs a;
    int rc = stem_args(c, &a);
    if (rc) return rc;
    a.x = x; a.dy = dy; a.dw = dw;
    const int64_t dyb = (int64_t)a.O[0] * a.O[1] * a.O[2] * a.Cy * 2;
    if (nndet_is16(c->dtype) && c->k[0] == 3 && c->k[1] == 3 && c->k[2] == 3 && c->s[0] == 1 && c->s[1] == 1 && c->s[2] == 1 &&
        c->p[0] == 1 && c->p[1] == 1 && c->p[2] == 1 && dyb < (1LL << 31) && !getenv("NNDET_STEM_VALU")) {
        const int nt0 = ceil_div(a.O[0], 4), nt1 = ceil_div(a.O[1], 8), nt2 = ceil_div(a.O[2], 8);
        const int64_t tiles = (int64_t)a.N * nt0 * nt1 * nt2;
        if (tiles < (1LL << 31)) {
            dim3 g3((unsigned)(tiles < 512 ? tiles : 512), c->cout_p / 32);
            if (c->dtype == NNDET_F16) k_stem_wgrad3<f16_t><<<g3, 256, 0, st>>>(a, nt0, nt1, nt2, (int)tiles);
            else k_stem_wgrad3<bf16_t><<<g3, 256, 0, st>>>(a, nt0, nt1, nt2, (int)tiles);
            LAUNCH_CHECK();
            return 0;
        }
    }
    int64_t nchunks = ceil_div64(a.total, 256);
    dim3 grid((unsigned)(nchunks < 1024 ? nchunks : 1024), c->cout_p / 32);
    const size_t lds = (27 * 257 + 1 + 256 * STEM_DYS_STRIDE) * sizeof(float);   // 64.6 KB
    static NndetDevOnce attr;
    if (attr.need()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_wgrad<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_wgrad<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_wgrad<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr.done();
    }
    if (c->dtype == NNDET_BF16) k_stem_wgrad<bf16_t><<<grid, 256, lds, st>>>(a);
    else if (c->dtype == NNDET_F16) k_stem_wgrad<f16_t><<<grid, 256, lds, st>>>(a);
    else k_stem_wgrad<float><<<grid, 256, lds, st>>>(a);
    LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------ fused stem block (C ABI)
static bool stem_block_ok(const NndetConv* c) {
    if (!c || !nndet_is16(c->dtype) || c->cin_p != 1 || c->cin != 1 || c->transposed || c->cout_p % 32 || c->in_affine) return false;
    for (int i = 0; i < 3; ++i) if (c->k[i] != 3 || c->s[i] != 1 || c->p[i] != 1) return false;
    const int64_t yb = (int64_t)c->out_d * c->out_h * c->out_w * c->cout_p * 2;
    return yb < (1LL << 31) && (int64_t)ceil_div(c->out_d, 4) * ceil_div(c->out_h, 8) * ceil_div(c->out_w, 8) < (1LL << 30);
}

extern "C" int32_t nndet_stem_block_supported(const NndetConv* c) { return stem_block_ok(c) ? 1 : 0; }

static int stem_block_grid(const StemArgs& a, int cout_p, int* S, int* nt) {
    nt[0] = ceil_div(a.O[0], 4); nt[1] = ceil_div(a.O[1], 8); nt[2] = ceil_div(a.O[2], 8);
    const int64_t tiles = (int64_t)nt[0] * nt[1] * nt[2];
    int s_ = 512 / (a.N * (cout_p / 32));
    if (s_ < 8) s_ = 8;
    s_ = (s_ / 8) * 8;
    if (s_ > tiles) s_ = (int)tiles;
    *S = s_;
    return 0;
}

extern "C" int nndet_stem_block_forward(const NndetConv* c, const void* x, const float* w_f32, const float* gamma, const float* beta,
                                        float eps, int32_t relu, void* out, double* stats, float* mean_rstd_out, void* stream) {
    if (!stem_block_ok(c) || !x || !w_f32 || !gamma || !beta || !out || !stats || !mean_rstd_out) return NNDET_EINVAL;
    hipStream_t st = as_stream(stream);
    StemArgs a;
    int rc = stem_args(c, &a);
    if (rc) return rc;
    a.x = x; a.w = w_f32; a.bias = nullptr; a.y = nullptr;
    int S, nt[3];
    stem_block_grid(a, c->cout_p, &S, nt);
    {   // NNDET_STEM_FWD_WGS: total workgroups of the two forward passes. Default 1024 = 4 per CU (20 KB of LDS, ~100 VGPRs): the waves are
        // independent of each other since round 5, more of them hide more latency (statistics pass alone: 104 us with 512, 84 with 1024, 79 with 2048)
        const char* e = getenv("NNDET_STEM_FWD_WGS");
        const int want = (e && atoi(e) >= 8) ? atoi(e) : 1024;
        {
            int s_ = want / (a.N * (c->cout_p / 32));
            s_ = s_ < 8 ? 8 : (s_ / 8) * 8;
            const int64_t tiles = (int64_t)nt[0] * nt[1] * nt[2];
            S = s_ > tiles ? (int)tiles : s_;
        }
    }
    const dim3 grid(S, c->cout_p / 32, a.N);
    if (c->dtype == NNDET_F16) k_stem_fwd3<f16_t, 1><<<grid, 256, 0, st>>>(a, nt[0], nt[1], nt[2], stats);
    else k_stem_fwd3<bf16_t, 1><<<grid, 256, 0, st>>>(a, nt[0], nt[1], nt[2], stats);
    LAUNCH_CHECK();
    const int64_t spatial = (int64_t)a.O[0] * a.O[1] * a.O[2];
    rc = norm_finalize_run(stats, a.N, c->cout, c->cout_p, c->cout, spatial, eps, mean_rstd_out, st);     // InstanceNorm: one group per channel
    if (rc) return rc;
    a.y = out;
    if (c->dtype == NNDET_F16) k_stem_fwd3<f16_t, 2><<<grid, 256, 0, st>>>(a, nt[0], nt[1], nt[2], nullptr, mean_rstd_out, gamma, beta, relu);
    else k_stem_fwd3<bf16_t, 2><<<grid, 256, 0, st>>>(a, nt[0], nt[1], nt[2], nullptr, mean_rstd_out, gamma, beta, relu);
    LAUNCH_CHECK();
    return 0;
}

extern "C" size_t nndet_stem_block_backward_workspace_bytes(const NndetConv* c) {
    if (!stem_block_ok(c)) return 0;
    return (size_t)c->batch * ((size_t)c->cout_p * 32 * 4 * 2 + 32 * 4 + (size_t)c->cout_p * 2 * 8) + 256;
}

extern "C" int nndet_stem_block_backward(const NndetConv* c, const void* x, const void* d_out, const float* w_f32, const float* mean_rstd,
                                         const float* gamma, const float* beta, int32_t relu, float* dw, float* dgamma, float* dbeta,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!stem_block_ok(c) || !x || !d_out || !w_f32 || !mean_rstd || !gamma || !beta || !dw || !dgamma || !dbeta || !workspace)
        return NNDET_EINVAL;
    const size_t need = nndet_stem_block_backward_workspace_bytes(c);
    if (workspace_bytes < need) return NNDET_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    StemArgs a;
    int rc = stem_args(c, &a);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(workspace, 0, need, st));
    StemBwdArgs b;
    memset(&b, 0, sizeof(b));
    b.x = x; b.dy = d_out; b.w = w_f32; b.coef = mean_rstd; b.gamma = gamma; b.beta = beta;
    char* wsp = reinterpret_cast<char*>(workspace);
    const size_t nA = (size_t)c->batch * c->cout_p * 32;
    b.accS = reinterpret_cast<double*>(wsp);                                         // 8-byte aligned part first
    b.accA = reinterpret_cast<float*>(wsp + (size_t)c->batch * c->cout_p * 2 * 8);
    b.accC = b.accA + nA;
    b.accB = b.accC + nA;
    b.N = a.N; b.Cy = a.Cy; b.cout = a.cout; b.relu = relu;
    for (int i = 0; i < 3; ++i) { b.I[i] = a.I[i]; b.O[i] = a.O[i]; }
    int S, nt[3];
    stem_block_grid(a, c->cout_p, &S, nt);
    {   // NNDET_STEM_BWD_WGS: total workgroups of the backward kernel (default 512 = 2 per CU; 51 KB of LDS each allow 3)
        const char* e = getenv("NNDET_STEM_BWD_WGS");
        if (e && atoi(e) >= 8) {
            int s_ = atoi(e) / (a.N * (c->cout_p / 32));
            s_ = s_ < 8 ? 8 : (s_ / 8) * 8;
            const int64_t tiles = (int64_t)nt[0] * nt[1] * nt[2];
            S = s_ > tiles ? (int)tiles : s_;
        }
    }
    const dim3 grid(S, c->cout_p / 32, a.N);
    if (c->dtype == NNDET_F16) k_stem_bwd3<f16_t><<<grid, 256, 0, st>>>(b, nt[0], nt[1], nt[2]);
    else k_stem_bwd3<bf16_t><<<grid, 256, 0, st>>>(b, nt[0], nt[1], nt[2]);
    LAUNCH_CHECK();
    k_stem_bwd_finish<<<1, 1024, 0, st>>>(b, dw, dgamma, dbeta);
    LAUNCH_CHECK();
    return 0;
}
