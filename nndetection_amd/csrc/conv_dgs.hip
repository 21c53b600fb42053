// Data gradient of the STRIDED 3x3x3 / pad 1 convolutions (encoder stage transitions, stride 2 per axis or (2, 2, 1)) for gfx950.
//
// k_igemm computes a strided data gradient by output-parity classes (no zero-tap work): class c = (cd, ch, cw) owns the outputs
// o = s * i + c and only the taps t with (c + p - t) % s == 0, reading dY at i + (c + p - t) / s. It runs ONE class per
// workgroup: the up-to-8 classes of a tile each stage the SAME dY halo into LDS, and with 1-8 taps per class the kernel is
// almost pure staging: 19 200 workgroups x 93 KB = 1.8 GB through the load -> ds_write path for 136 GFLOP (32 <- 64 channels at
// 160^3: 0.49 ms, 279 TFLOP/s).
// Here a workgroup stages the halo of its lattice tile ONCE, for all channel chunks, and then walks over the classes:
// per class zero 32-64 accumulators, run that class' taps x chunks out of LDS, store the class' outputs (stride-s scatter).
//   lattice tile 4 x 8 x 8 points (x classes = up to 8 x 16 x 16 outputs), one d-plane per wave, 4 point tiles of 16 per wave;
//   halo = tile + (hi - lo) per axis (stride 2: deltas {0, 1} -> +1; stride 1: {-1, 0, 1} -> +2), chunk-major [chunk][voxel][64 B]
//   with the row-parity XOR swizzle of k_igemm (conflict-free ds_read_b128 for unit-stride points);
//   MFMA operands as everywhere: A = packed mode-1 weights [tap][cin_p rows][cout_p k] from global memory (one fragment per
//   row tile, tap and chunk, prefetched one (tap, chunk) step ahead), B = dY fragments from LDS.
// Optional residual (dx += ...: the fused gradient accumulation of nndet_conv3d_backward_data_acc), read at the element it writes.
#include "common.h"
#include "conv_common.h"
#include <type_traits>

template <typename T> struct DgMma {     // the 16-bit storage types (bf16_t, f16_t)
    static constexpr int KC = 32, EPL = 8;
    typedef f32x4 acc_t;
    __device__ static __forceinline__ acc_t zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ static __forceinline__ f32x4 f32(const acc_t& c) { return c; }
    __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) { c = H16<T>::mma(a, b, c); }
    __device__ static __forceinline__ void store4(T* p, float a, float b, float c, float d) {
        uint2 v; v.x = H16<T>::pack2(a, b); v.y = H16<T>::pack2(c, d);
        *reinterpret_cast<uint2*>(p) = v;
    }
    __device__ static __forceinline__ void load4(const T* p, float* v) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        v[0] = H16<T>::lo(u.x); v[1] = H16<T>::hi(u.x);
        v[2] = H16<T>::lo(u.y); v[3] = H16<T>::hi(u.y);
    }
};
template <> struct DgMma<float> {          // the parity path accumulates in float64 (conv_igemm.hip: Mma<float>)
    static constexpr int KC = 16, EPL = 4;
    typedef f64x4_t acc_t;
    __device__ static __forceinline__ acc_t zero() { return f64x4_t{0.0, 0.0, 0.0, 0.0}; }
    __device__ static __forceinline__ f32x4 f32(const acc_t& c) { return f64acc_rows_to_f32(c); }
    __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, acc_t& c) {
        const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[0], (double)fb[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[1], (double)fb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[2], (double)fb[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[3], (double)fb[3], c, 0, 0, 0);
    }
    __device__ static __forceinline__ void store4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }
    __device__ static __forceinline__ void load4(const float* p, float* v) {
        const float4 f = *reinterpret_cast<const float4*>(p); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    }
};

struct DgsClass { int32_t c[3]; int32_t L[3]; int32_t tap0, ntap; };
struct DgsTap { int32_t toff; int32_t flip; int32_t wt; int32_t pad; };   // LDS byte offset of the delta inside a chunk image, swizzle flip, weight tap
struct DgsArgs {
    const void* dy; const void* w; const void* res; void* dx;
    int32_t N, K, R;            // dY channels (= cout_p, the contraction), dx channels (= cin_p, the rows)
    int32_t O[3], I[3];         // dY spatial dims (conv output), dx spatial dims (conv input)
    int32_t s[3], lo[3];        // stride, smallest delta per axis
    int32_t H[3], nt[3];        // halo dims, lattice tiles per axis
    uint32_t mHW, mHH, mHV4;    // magic multipliers for / H[2], / H[1], / (halo voxels * 4)
    int32_t chb;                // bytes of one chunk image in LDS (halo voxels * 64)
    int32_t ncls;
    DgsClass cls[8];
    DgsTap taps[27];
    // Norm-backward reduction in the epilogue (round 5, k_dgs<..., NB = true>): dx is the COMPLETE gradient w.r.t. the normalised (+ReLU)
    // output of a conv -> norm -> ReLU block (this launch adds the last contribution, nndet_conv3d_backward_data_acc), so the sums
    // k_norm_bwd_reduce would read it back for -- S1 = sum g, S2 = sum g * xhat per (image, channel), g = dx * [ReLU mask] -- are
    // accumulated here from the values being stored and the block's pre-norm tensor `ny` (one extra read instead of two).
    const void* ny;             // pre-norm tensor of that block, same shape / layout as dx
    const float* nmr;           // [N][R][2] mean, rstd (k_norm_finalize)
    const float* ngamma; const float* nbeta;   // [ncout]
    double* nred;               // [replicas][N][R][2] (k_norm_bwd_reduce's replica layout, zeroed)
    int32_t nrelu, ncout;
};

#define DGS_TD 4
#define DGS_TH 8
#define DGS_TW 8
// NB: also accumulate the norm-backward sums of the block that produced this convolution's input (DgsArgs::ny ...): 16-bit types, 32 rows
// per workgroup. Costs 32 + 16 + 16 registers (pre-norm values in flight with the residual, mask constants, partial sums): 2 workgroups
// per CU instead of 3. Measured at full resolution (batch 2, alone): 265 -> 405 us for this kernel against the 220 us reduction pass
// it replaces (profiles/round5_ab_norm_red_fuse.txt; variants with the constants in LDS, loads behind the MFMA loop or one class at
// a time were 417-518 us).
template <typename T, int MT, int MAXP, int G, bool NB = false>
__global__ __launch_bounds__(256, (MT == 2 && sizeof(T) == 2) ? (NB ? 2 : 3) : 1) void k_dgs(const DgsArgs A) {
    using M = DgMma<T>;
    constexpr int KC = M::KC, EPL = M::EPL, NT = 4;
    static_assert(!NB || (sizeof(T) == 2 && MT == 2), "the norm-backward epilogue: 16-bit types, 32 rows per workgroup");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float nb_c[NB ? 32 : 1][4];          // per channel of this workgroup's rows: mean, rstd, scale, shift
    __shared__ double nb_s[NB ? 32 : 1][2];         // S1, S2 of this workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int n = blockIdx.z;
    int tt = xcd_compact(blockIdx.x, gridDim.x, 768);
    const int tw_i = tt % A.nt[2]; tt /= A.nt[2];
    const int th_i = tt % A.nt[1];
    const int td_i = tt / A.nt[1];
    const int l0d = td_i * DGS_TD, l0h = th_i * DGS_TH, l0w = tw_i * DGS_TW;
    const int row0 = blockIdx.y * (MT * 16);
    const int HH = A.H[1], HW = A.H[2];
    const int HV4 = A.H[0] * HH * HW * 4;
    const int nchunk = A.K / KC;

    // ---- stage the dY halo of the lattice tile, every chunk (the only staging of this workgroup)
    {
        const T* dyn = reinterpret_cast<const T*>(A.dy) + (int64_t)n * A.O[0] * A.O[1] * A.O[2] * A.K;
        const int total = HV4 * nchunk;
        const int i0d = l0d + A.lo[0], i0h = l0h + A.lo[1], i0w = l0w + A.lo[2];
#pragma unroll
        for (int s0 = 0; s0 < MAXP; s0 += 8) {
            if (s0 * 256 >= total) break;      // uniform
            u32x4 v[8];
            int dst[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int p = tid + (s0 + b) * 256;
                const int kc = (int)__umulhi((unsigned)p, A.mHV4);
                const int pi = p - kc * HV4;
                const int hv = pi >> 2, part = pi & 3;
                const int t2 = A.mHW ? (int)__umulhi((unsigned)hv, A.mHW) : hv;
                const int hw = hv - t2 * HW;
                const int hd = A.mHH ? (int)__umulhi((unsigned)t2, A.mHH) : t2;
                const int hh = t2 - hd * HH;
                const int id = i0d + hd, ih = i0h + hh, iw = i0w + hw;
                const bool ok = p < total && (unsigned)id < (unsigned)A.O[0] && (unsigned)ih < (unsigned)A.O[1] && (unsigned)iw < (unsigned)A.O[2];
                const int64_t o = ok ? ((int64_t)(id * A.O[1] + ih) * A.O[2] + iw) * A.K + kc * KC + part * EPL : 0;
                v[b] = *reinterpret_cast<const u32x4*>(dyn + o);          // unconditional (clamped) load: all 8 in flight together
                if (!ok) v[b] = u32x4{0u, 0u, 0u, 0u};
                dst[b] = p < total ? kc * A.chb + ((pi * 16) ^ ((t2 & 1) << 5)) : -1;
            }
#pragma unroll
            for (int b = 0; b < 8; ++b)
                if (dst[b] >= 0) *reinterpret_cast<u32x4*>(smem + dst[b]) = v[b];
        }
    }
    // LDS byte offsets of this lane's lattice points at delta == lo (tap offset 0), chunk 0
    int boff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ph = 2 * j + (li >> 3), pw = li & 7;
        const int brow = wv * HH + ph;
        boff[j] = ((brow * HW + pw) * 64 + q * 16) ^ ((brow & 1) << 5);
    }
    const T* wl = reinterpret_cast<const T*>(A.w) + (int64_t)(row0 + li) * A.K + q * EPL;
    T* dxn = reinterpret_cast<T*>(A.dx) + (int64_t)n * A.I[0] * A.I[1] * A.I[2] * A.R;
    const T* rsn = A.res ? reinterpret_cast<const T*>(A.res) + (int64_t)n * A.I[0] * A.I[1] * A.I[2] * A.R : nullptr;
    const T* nyn = nullptr;
    float nsa[MT][4], nsb[MT][4];
    if constexpr (NB) {
        nyn = reinterpret_cast<const T*>(A.ny) + (int64_t)n * A.I[0] * A.I[1] * A.I[2] * A.R;
        if (tid < 32) {
            const int ch = row0 + tid;
            const bool ok = ch < A.ncout;
            const float mu = A.nmr[((int64_t)n * A.R + ch) * 2], rs = A.nmr[((int64_t)n * A.R + ch) * 2 + 1];
            const float sc = ok ? rs * A.ngamma[ok ? ch : 0] : 0.f;
            nb_c[tid][0] = ok ? mu : 0.f; nb_c[tid][1] = ok ? rs : 0.f; nb_c[tid][2] = sc;
            nb_c[tid][3] = ok ? A.nbeta[ok ? ch : 0] - mu * sc : 0.f;           // the forward pass's expressions (k_norm_apply)
            nb_s[tid][0] = 0.0; nb_s[tid][1] = 0.0;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { nsa[i][rr] = 0.f; nsb[i][rr] = 0.f; }
    }
    __syncthreads();
    float csc[NB ? MT : 1][4], csh[NB ? MT : 1][4];          // this lane's 8 channels: the forward pass's scale / shift (ReLU mask)
    if constexpr (NB) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { csc[i][rr] = nb_c[i * 16 + q * 4 + rr][2]; csh[i][rr] = nb_c[i * 16 + q * 4 + rr][3]; }
    }

    // Classes come in groups of G consecutive ones (G = 2: the two W parities of one (cd, ch)): their outputs interleave voxel by
    // voxel along W, i.e. the two classes fill the two halves of every 128-byte line. Stored one class at a time the half lines sit
    // in L2 until the other class of the workgroup is done (microseconds later, 12 MB of half-written lines in flight per XCD against
    // 4 MB of L2); the group keeps both accumulator sets and stores the two halves back to back.
    for (int cg = 0; cg < A.ncls; cg += G) {
        typename M::acc_t acc_k[G][MT][NT];
        bool live[G];
        // bf16 + residual (the fused gradient accumulation): the group's residual pieces are fetched NOW, 16 bytes per lane in the
        // layout of the epilogue's stores, and wait under the group's MFMA steps; the epilogue turns them back into the MFMA layout
        // with the same v_permlane16_swap (an involution). Loading them 8 bytes at a time right before the add exposed an HBM round
        // trip per point tile: 32 <- 64 channels at 160^3 took 0.60 ms with the residual against 0.33 ms without.
        constexpr bool RES16 = sizeof(T) == 2 && MT % 2 == 0;
        u32x4 rpre[RES16 ? G : 1][RES16 ? MT / 2 : 1][RES16 ? NT : 1];
        u32x4 ypre[NB ? G : 1][NB ? MT / 2 : 1][NB ? NT : 1];          // the pre-norm values at the group's outputs, same layout / timing
        auto load_y = [&]() {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const DgsClass& C = A.cls[cg + g];
                const bool lv = !(l0d >= C.L[0] || l0h >= C.L[1] || l0w >= C.L[2]);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int ld = l0d + wv, lh = l0h + 2 * j + (li >> 3), lw = l0w + (li & 7);
                    const bool pv = lv && ld < C.L[0] && lh < C.L[1] && lw < C.L[2];
                    const int od = ld * A.s[0] + C.c[0], oh = lh * A.s[1] + C.c[1], ow = lw * A.s[2] + C.c[2];
                    const int64_t vo = pv ? (((int64_t)od * A.I[1] + oh) * A.I[2] + ow) * A.R + row0 : 0;     // clamped: loaded, not used
#pragma unroll
                    for (int i = 0; i < MT; i += 2)
                        ypre[NB ? g : 0][NB ? i / 2 : 0][NB ? j : 0] = *reinterpret_cast<const u32x4*>(nyn + vo + i * 16 + (q >> 1) * 8 + (q & 1) * 16);
                }
            }
        };
        if constexpr (NB) load_y();
        auto load_res = [&]() {
            if (rsn) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const DgsClass& C = A.cls[cg + g];
                    const bool lv = !(l0d >= C.L[0] || l0h >= C.L[1] || l0w >= C.L[2]);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int ld = l0d + wv, lh = l0h + 2 * j + (li >> 3), lw = l0w + (li & 7);
                        const bool pv = lv && ld < C.L[0] && lh < C.L[1] && lw < C.L[2];
                        const int od = ld * A.s[0] + C.c[0], oh = lh * A.s[1] + C.c[1], ow = lw * A.s[2] + C.c[2];
                        const int64_t vo = pv ? (((int64_t)od * A.I[1] + oh) * A.I[2] + ow) * A.R + row0 : 0;     // clamped: loaded, not used
#pragma unroll
                        for (int i = 0; i < MT; i += 2)
                            rpre[RES16 ? g : 0][RES16 ? i / 2 : 0][RES16 ? j : 0] = *reinterpret_cast<const u32x4*>(rsn + vo + i * 16 + (q >> 1) * 8 + (q & 1) * 16);
                    }
                }
            }
        };
        if constexpr (RES16) load_res();
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const DgsClass& C = A.cls[cg + g];
            live[g] = !(l0d >= C.L[0] || l0h >= C.L[1] || l0w >= C.L[2]);      // uniform: the tile lies outside this class' lattice
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc_k[g][i][j] = M::zero();
            if (!live[g]) continue;
            const int nstep = C.ntap * nchunk;                                   // (tap, chunk) steps, chunk fastest
            u32x4 af[MT], afn[MT];
            auto load_w = [&](int st, u32x4* a_) {
                const int tp = st / nchunk, kc = st - tp * nchunk;
                const T* wt = wl + ((int64_t)A.taps[C.tap0 + tp].wt * A.R) * A.K + kc * KC;
#pragma unroll
                for (int i = 0; i < MT; ++i) a_[i] = *reinterpret_cast<const u32x4*>(wt + (int64_t)i * 16 * A.K);
            };
            if (nstep > 0) load_w(0, afn);
            for (int st = 0; st < nstep; ++st) {
                const int tp = st / nchunk, kc = st - tp * nchunk;
                const DgsTap tap = A.taps[C.tap0 + tp];
                const char* base = smem + kc * A.chb + tap.toff;
                u32x4 bf[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const u32x4*>(base + (boff[j] ^ tap.flip));
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = afn[i];
                if (st + 1 < nstep) load_w(st + 1, afn);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) M::mma(af[i], bf[j], acc_k[g][i][j]);
            }
        }
        // ---- the group's outputs: o = s * i + c
        f32x4 acc[G][MT][NT];                       // (uniform control flow here: DgMma<float>::f32 exchanges values between lanes)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[g][i][j] = M::f32(acc_k[g][i][j]);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int ld = l0d + wv, lh = l0h + 2 * j + (li >> 3), lw = l0w + (li & 7);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const DgsClass& C = A.cls[cg + g];
                const bool pv = live[g] && ld < C.L[0] && lh < C.L[1] && lw < C.L[2];
                if constexpr (sizeof(T) == 2) {
                    // bf16: row tiles in pairs, ONE 16-byte store per lane and pair (v_permlane16_swap: lane row q ends up with 8 consecutive
                    // channels, q = 0: 0-7, 1: 16-23, 2: 8-15, 3: 24-31; the partner lane (q ^ 1, same li) is the same point -> same pv)
                    typedef unsigned int dgs_v2u __attribute__((ext_vector_type(2)));
                    const int od = ld * A.s[0] + C.c[0], oh = lh * A.s[1] + C.c[1], ow = lw * A.s[2] + C.c[2];
                    const int64_t vo = pv ? (((int64_t)od * A.I[1] + oh) * A.I[2] + ow) * A.R + row0 : 0;
#pragma unroll
                    for (int i = 0; i < MT; i += 2) {
                        uint32_t pk[2][2], rk[2][2] = {{0u, 0u}, {0u, 0u}};
                        if (rsn) {             // store layout -> MFMA layout: rk[h] = the packed residual of row tile i + h
                            const u32x4 r16 = rpre[RES16 ? g : 0][RES16 ? i / 2 : 0][RES16 ? j : 0];
                            const dgs_v2u t0 = __builtin_amdgcn_permlane16_swap(r16[0], r16[2], false, false);
                            const dgs_v2u t1 = __builtin_amdgcn_permlane16_swap(r16[1], r16[3], false, false);
                            rk[0][0] = t0[0]; rk[1][0] = t0[1]; rk[0][1] = t1[0]; rk[1][1] = t1[1];
                        }
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float v0 = acc[g][i + h][j][0], v1 = acc[g][i + h][j][1], v2 = acc[g][i + h][j][2], v3 = acc[g][i + h][j][3];
                            if (rsn) {
                                v0 += H16<T>::lo(rk[h][0]); v1 += H16<T>::hi(rk[h][0]);
                                v2 += H16<T>::lo(rk[h][1]); v3 += H16<T>::hi(rk[h][1]);
                            }
                            pk[h][0] = H16<T>::pack2(v0, v1); pk[h][1] = H16<T>::pack2(v2, v3);
                        }
                        if constexpr (NB) {        // S1 / S2 from the ROUNDED values (what k_norm_bwd_apply will read back) and the pre-norm tensor
                            const u32x4 y16 = ypre[g][i / 2][j];
                            const dgs_v2u y0 = __builtin_amdgcn_permlane16_swap(y16[0], y16[2], false, false);
                            const dgs_v2u y1 = __builtin_amdgcn_permlane16_swap(y16[1], y16[3], false, false);
                            const uint32_t yk[2][2] = {{y0[0], y1[0]}, {y0[1], y1[1]}};
                            if (pv) {
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const float gv[4] = {H16<T>::lo(pk[h][0]), H16<T>::hi(pk[h][0]), H16<T>::lo(pk[h][1]), H16<T>::hi(pk[h][1])};
                                    const float yv[4] = {H16<T>::lo(yk[h][0]), H16<T>::hi(yk[h][0]), H16<T>::lo(yk[h][1]), H16<T>::hi(yk[h][1])};
#pragma unroll
                                    for (int rr = 0; rr < 4; ++rr) {
                                        float gm = gv[rr];
                                        if (A.nrelu && !(fmaf(yv[rr], csc[NB ? i + h : 0][rr], csh[NB ? i + h : 0][rr]) > 0.f)) gm = 0.f;   // the forward pass's expression
                                        nsa[i + h][rr] += gm;
                                        nsb[i + h][rr] = fmaf(gm, yv[rr], nsb[i + h][rr]);          // RAW moment sum g*y: centred once per workgroup below
                                    }
                                }
                            }
                        }
                        const dgs_v2u s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                        const dgs_v2u s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                        if (pv) *reinterpret_cast<u32x4*>(dxn + vo + i * 16 + (q >> 1) * 8 + (q & 1) * 16) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                    }
                } else if (pv) {
                    const int od = ld * A.s[0] + C.c[0], oh = lh * A.s[1] + C.c[1], ow = lw * A.s[2] + C.c[2];
                    const int64_t eo = (((int64_t)od * A.I[1] + oh) * A.I[2] + ow) * A.R + row0 + q * 4;
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        float v0 = acc[g][i][j][0], v1 = acc[g][i][j][1], v2 = acc[g][i][j][2], v3 = acc[g][i][j][3];
                        if (rsn) {
                            float r4[4];
                            M::load4(rsn + eo + i * 16, r4);
                            v0 += r4[0]; v1 += r4[1]; v2 += r4[2]; v3 += r4[3];
                        }
                        M::store4(dxn + eo + i * 16, v0, v1, v2, v3);
                    }
                }
            }
        }
    }
    if constexpr (NB) {
        // lanes li = 0..15 of a row q hold the same 8 channels: add over the 16 points, then the four waves (d-planes) through LDS,
        // then ONE fp64 atomic per (channel, sum) and workgroup into k_norm_bwd_reduce's replica layout
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float a = nsa[i][rr], b = nsb[i][rr];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (li == 0) {
                    atomicAdd(&nb_s[i * 16 + q * 4 + rr][0], (double)a);
                    atomicAdd(&nb_s[i * 16 + q * 4 + rr][1], (double)b);
                }
            }
        __syncthreads();
        if (tid < 64) {
            const int rep = blockIdx.x % NNDET_STATS_REPLICAS;
            double v = nb_s[tid >> 1][tid & 1];
            if (tid & 1) v = (double)nb_c[tid >> 1][1] * (v - (double)nb_c[tid >> 1][0] * nb_s[tid >> 1][0]);   // sum g*xhat = rstd * (sum g*y - mean * sum g)
            if (v != 0.0) atomicAdd(A.nred + (((int64_t)rep * A.N + n) * A.R + row0 + (tid >> 1)) * 2 + (tid & 1), v);
        }
    }
}

// ------------------------------------------------------------------------------------------------ persistent form (round 6)
// k_dgsp: the 32 <- 64 channel stride-(2, 2, 2) transition in 16 bits (encoder stage 0 <- 1, the full-resolution data gradient that also
// completes the gradient of the first block's output and carries its norm-backward sums). k_dgs moves 2.0 GB for it at 2.7 TB/s
// (0.75 ms alone at batch 4, 0.81 ms in the step; profiles/round5_dgs_pmc.txt): one workgroup per tile, and inside it six memory round
// trips in a row -- two for the register-staged halo, then per class pair "request residual + pre-norm values, run a 1 us MFMA loop whose
// first weight load (vmcnt is in-order) waits for them, epilogue" -- with 11.5 k instructions per wave and tile, a third of them address
// arithmetic (59 % of the wave cycles in s_waitcnt, 32 % issuing). Here
//   * ONE persistent workgroup per CU (4 waves, up to 512 registers each) walks the lattice tiles;
//   * all 27 x 2 x 2 weight fragments live in LDS for the life of the workgroup (108 KB, lane-linear: conflict-free ds_read_b128), so the
//     MFMA loops issue no vector-memory instruction at all;
//   * the dY halo of the NEXT tile goes global -> LDS by LDS-DMA (source pre-swizzled, out-of-tensor granules = out-of-range offsets ->
//     the hardware writes the zero padding) right after the last MFMA loop of the current tile, under its last epilogue;
//   * residual and pre-norm values of class pair g + 1 are requested BEFORE the MFMA loop of pair g into the second of two register sets
//     (16 x 16-byte buffer loads per lane) and are consumed one MFMA loop + one epilogue later; every address is
//     descriptor (image) + scalar (tile, class) + a lane constant: no per-access address arithmetic, validity by out-of-range offsets;
//   * stride (2, 2, 2) is compile time: the 8 parity classes with their 1-8 taps are fully unrolled with immediate LDS offsets, in the
//     tap / chunk order of k_dgs (dx is bit-identical to k_dgs).
// LDS: 108 KB weights + 51 KB halo (5 x 9 x 9 voxels x 128 B, single buffer: the MFMA loops are 40 % of a tile, the DMA lands under the
// epilogue that follows). NNDET_DGSP=0 restores k_dgs.
struct DgspArgs {
    const void* dy; const void* w; const void* res; void* dx;
    int32_t N, O[3], I[3], nt[3];
    int32_t total_tiles;
    const void* ny; const float* nmr; const float* ngamma; const float* nbeta; double* nred;
    int32_t nrelu, ncout;
};

__device__ __forceinline__ void dgsp_dma16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dgsp_rsrc(const char* p, int num_records) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(((uint64_t)hi << 32) | lo), 0, num_records, 0x00020000);
}
typedef __attribute__((address_space(3))) char* dgsp_lds_ptr;

template <typename T, bool RES, bool NB>
__global__ __launch_bounds__(256, 1) void k_dgsp(const DgspArgs A) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    constexpr int HD = 5, HH = 9, HW = 9, HV = HD * HH * HW, CHB = HV * 64, NGR = 2 * HV * 4, NPIECE = (NGR + 63) / 64;
    constexpr int WBYTES = 108 * 1024, NPW = (NPIECE + 3) / 4;
    constexpr int NLOAD = (RES ? 8 : 0) + (NB ? 8 : 0);               // vector-memory loads of one prefetch set per wave
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int G = gridDim.x, bx = blockIdx.x;
    const uint32_t sb = (uint32_t)(uintptr_t)(dgsp_lds_ptr)smem;

    // ---- weights -> LDS, once: fragment f = (tap * 2 + chunk) * 2 + row tile, 64 lanes x 16 bytes, lane-linear
    {
        const T* wl = reinterpret_cast<const T*>(A.w) + (int64_t)li * 64 + q * 8;
        for (int f = wv; f < 108; f += 4) {
            const int i = f & 1, kc = (f >> 1) & 1, tap = f >> 2;
            const u32x4 v = *reinterpret_cast<const u32x4*>(wl + ((int64_t)tap * 32 + i * 16) * 64 + kc * 32);
            *reinterpret_cast<u32x4*>(smem + f * 1024 + lane * 16) = v;
        }
    }
    // ---- lane constants
    const uint32_t wa0 = sb + lane * 16, wa1 = wa0 + 65536;             // weight fragments: base + f * 1024 (f >= 64 from the second base)
    uint32_t hb[4][2];                                                  // dY fragments: halo base of point tile j for the two swizzle phases
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ph = 2 * j + (li >> 3), pw = li & 7;
        const int brow = wv * HH + ph;
        const int b0 = ((brow * HW + pw) * 64 + q * 16) ^ ((brow & 1) << 5);
        hb[j][0] = sb + WBYTES + b0; hb[j][1] = sb + WBYTES + (b0 ^ 32);
    }
    asm volatile("" : "+v"(hb[0][0]), "+v"(hb[0][1]), "+v"(hb[1][0]), "+v"(hb[1][1]));
    asm volatile("" : "+v"(hb[2][0]), "+v"(hb[2][1]), "+v"(hb[3][0]), "+v"(hb[3][1]));
    // output side: byte offset of this lane's 16 bytes of lattice point (wv, 2 j + (li >> 3), li & 7) relative to the tile's class-0 origin
    const int I1 = A.I[1], I2 = A.I[2];
    int lo_out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        lo_out[j] = ((2 * wv * I1 + 2 * (2 * j + (li >> 3))) * I2 + 2 * (li & 7)) * 64 + ((q >> 1) * 8 + (q & 1) * 16) * 2;
    // halo DMA: piece p = wv + 4 k, granule g = p * 64 + lane -> chunk, voxel, stored part; source = logical part (swizzle undone)
    int hrel[NPW];
    uint32_t hsel[NPW];          // bit hd | bit 5 + hh | bit 14 + hw; bit 31 = no such granule
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int g = (wv + 4 * k) * 64 + lane;
        const int kc = g / (HV * 4), gi = g - kc * (HV * 4);
        const int hv = gi >> 2, slot = gi & 3;
        const int row = hv / HW, hw = hv - row * HW;
        const int hd = row / HH, hh = row - hd * HH;
        const int lp = slot ^ ((row & 1) << 1);
        hsel[k] = g < NGR ? (1u << hd) | (1u << (5 + hh)) | (1u << (14 + hw)) : 0x80000000u;
        hrel[k] = ((hd * A.O[1] + hh) * A.O[2] + hw) * 128 + kc * 64 + lp * 16;
    }
    const int img_out = A.I[0] * I1 * I2 * 64, img_dy = A.O[0] * A.O[1] * A.O[2] * 128;      // bytes per image (< 2^31: host)
    // class lattices (stride 2): L[axis][parity]
    const int L0[2] = {(A.I[0] + 1) >> 1, A.I[0] >> 1}, L1[2] = {(I1 + 1) >> 1, I1 >> 1}, L2[2] = {(I2 + 1) >> 1, I2 >> 1};

    // ---- tile walk (flat index over (n, td, th, tw); XCD-compact start as in k_wgrad3e)
    const int pstart = (G & 7) ? bx : (bx & 7) * (G >> 3) + (bx >> 3);
    const int ntile = pstart < A.total_tiles ? (A.total_tiles - pstart + G - 1) / G : 0;
    const int tpi = A.nt[0] * A.nt[1] * A.nt[2];
    struct Tile { int n, l0d, l0h, l0w, obase; };
    auto locate = [&](int flat) -> Tile {
        Tile t;
        t.n = flat / tpi;
        int tt = flat - t.n * tpi;
        const int tw = tt % A.nt[2]; tt /= A.nt[2];
        const int th = tt % A.nt[1], td = tt / A.nt[1];
        t.l0d = td * DGS_TD; t.l0h = th * DGS_TH; t.l0w = tw * DGS_TW;
        t.obase = ((2 * t.l0d * I1 + 2 * t.l0h) * I2 + 2 * t.l0w) * 64;
        return t;
    };
    auto issue_halo = [&](const Tile& t) {
        const auto drs = dgsp_rsrc(reinterpret_cast<const char*>(A.dy) + (int64_t)t.n * img_dy, img_dy);
        const int soff = ((t.l0d * A.O[1] + t.l0h) * A.O[2] + t.l0w) * 128;
        auto rng = [](int hi) -> uint32_t { return hi >= 32 ? 0xffffffffu : (hi <= 0 ? 0u : (1u << hi) - 1u); };
        const uint32_t mask = rng(min(HD, A.O[0] - t.l0d)) | (rng(min(HH, A.O[1] - t.l0h)) << 5) | (rng(min(HW, A.O[2] - t.l0w)) << 14);
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            if ((k + 1) * 4 <= NPIECE || wv + 4 * k < NPIECE) {
                const bool ok = (hsel[k] & mask) == hsel[k];
                dgsp_dma16(drs, ok ? hrel[k] : (int)0x80000000, soff, (uint32_t)__builtin_amdgcn_readfirstlane(sb + WBYTES + (wv + 4 * k) * 1024));
            }
        }
    };

    // ---- norm-backward sums (NB): constants of the lane's 8 channels for image n, partial sums across this workgroup's tiles of image n
    float csc[2][4], csh[2][4], nsa[2][4], nsb[2][4];
    int n_cur = -1;
    auto flush = [&]() {
        if constexpr (NB) {
            if (n_cur < 0) return;
            const int rep = (bx * 4 + wv) % NNDET_STATS_REPLICAS;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    float a = nsa[i][rr], b = nsb[i][rr];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                    const int ch = i * 16 + q * 4 + rr;
                    if (li == 0 && ch < A.ncout) {
                        const double mu = (double)A.nmr[((int64_t)n_cur * 32 + ch) * 2], rs = (double)A.nmr[((int64_t)n_cur * 32 + ch) * 2 + 1];
                        double* dst = A.nred + (((int64_t)rep * A.N + n_cur) * 32 + ch) * 2;
                        if (a != 0.f) atomicAdd(dst, (double)a);
                        const double v2 = rs * ((double)b - mu * (double)a);      // sum g * xhat = rstd * (sum g * y - mean * sum g)
                        if (v2 != 0.0) atomicAdd(dst + 1, v2);
                    }
                    nsa[i][rr] = 0.f; nsb[i][rr] = 0.f;
                }
        }
    };
    auto enter_image = [&](int n) {
        if constexpr (NB) {
            if (n == n_cur) return;
            flush();
            n_cur = n;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int ch = i * 16 + q * 4 + rr;
                    const bool ok = ch < A.ncout;
                    const float mu = A.nmr[((int64_t)n * 32 + ch) * 2], rs = A.nmr[((int64_t)n * 32 + ch) * 2 + 1];
                    const float sc = ok ? rs * A.ngamma[ok ? ch : 0] : 0.f;
                    csc[i][rr] = sc; csh[i][rr] = ok ? A.nbeta[ok ? ch : 0] - mu * sc : 0.f;       // the forward pass's expressions (k_norm_apply)
                    nsa[i][rr] = 0.f; nsb[i][rr] = 0.f;
                }
        }
    };

    // ---- prefetch sets: residual / pre-norm values of one class pair (cd, ch), both W parities, 4 point tiles
    struct PSet { u32x4 r[2][4], y[2][4]; int vo[2][4]; };
    auto request = [&](PSet& P, const Tile& t, int cd, int ch) {
        const auto rrs = dgsp_rsrc(reinterpret_cast<const char*>(RES ? A.res : A.dx) + (int64_t)t.n * img_out, img_out);
        const auto yrs = dgsp_rsrc(reinterpret_cast<const char*>(NB ? A.ny : A.dx) + (int64_t)t.n * img_out, img_out);
        const bool dv = t.l0d + wv < L0[cd];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int soff = t.obase + ((cd * I1 + ch) * I2 + g) * 64;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool pv = dv && (t.l0h + 2 * j + (li >> 3) < L1[ch]) && (t.l0w + (li & 7) < L2[g]);
                P.vo[g][j] = pv ? lo_out[j] : (int)0x80000000;             // out of range: loads return 0, the store is dropped
                if constexpr (RES) P.r[g][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, P.vo[g][j], soff, 0));
                if constexpr (NB) P.y[g][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs, P.vo[g][j], soff, 0));
            }
        }
    };

    // ---- one class pair: MFMA loops of both W parities (taps in k_dgs's order: d, h, w entries; parity 1 = {(t 0, delta 1), (t 2, delta 0)})
    f32x4 acc[2][2][4];
    typedef __attribute__((address_space(3))) u32x4* lds_u32x4_ptr;
    auto lds128 = [](uint32_t a) -> u32x4 { return *(lds_u32x4_ptr)(uintptr_t)a; };
    auto mfma_pair = [&](auto cd_, auto ch_) {
        constexpr int cd = decltype(cd_)::value, ch = decltype(ch_)::value;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[g][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            constexpr int nd = cd ? 2 : 1, nh = ch ? 2 : 1;
            const int nw = g ? 2 : 1;
#pragma unroll
            for (int x = 0; x < nd; ++x)
#pragma unroll
                for (int y = 0; y < nh; ++y)
#pragma unroll
                    for (int z = 0; z < 2; ++z) {
                        if (z >= nw) continue;
                        const int td = cd ? (x ? 2 : 0) : 1, dd = cd ? (x ? 0 : 1) : 0;
                        const int th = ch ? (y ? 2 : 0) : 1, dh = ch ? (y ? 0 : 1) : 0;
                        const int tw = g ? (z ? 2 : 0) : 1, dw = g ? (z ? 0 : 1) : 0;
                        const int wt = (td * 3 + th) * 3 + tw;
                        const int trow = dd * HH + dh;
                        const int toff = (trow * HW + dw) * 64, ph = trow & 1;
#pragma unroll
                        for (int kc = 0; kc < 2; ++kc) {
                            u32x4 af[2], bf[4];
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int f = (wt * 2 + kc) * 2 + i;
                                af[i] = f < 64 ? lds128(wa0 + f * 1024) : lds128(wa1 + (f - 64) * 1024);
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) bf[j] = lds128(hb[j][ph] + kc * CHB + toff);
#pragma unroll
                            for (int i = 0; i < 2; ++i)
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[g][i][j] = H16<T>::mma(af[i], bf[j], acc[g][i][j]);
                        }
                    }
        }
    };
    // ---- epilogue of a class pair: (+ residual) -> pack -> (norm-backward sums from the ROUNDED values) -> one 16-byte store per lane and point
    auto epilogue = [&](PSet& P, const Tile& t, int cd, int ch) {
        const auto ors = dgsp_rsrc(reinterpret_cast<char*>(A.dx) + (int64_t)t.n * img_out, img_out);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int soff = t.obase + ((cd * I1 + ch) * I2 + g) * 64;
                uint32_t pk[2][2], rk[2][2] = {{0u, 0u}, {0u, 0u}};
                if constexpr (RES) {                 // store layout -> MFMA layout (v_permlane16_swap is an involution)
                    const u32x4 r16 = P.r[g][j];
                    const v2u t0 = __builtin_amdgcn_permlane16_swap(r16[0], r16[2], false, false);
                    const v2u t1 = __builtin_amdgcn_permlane16_swap(r16[1], r16[3], false, false);
                    rk[0][0] = t0[0]; rk[1][0] = t0[1]; rk[0][1] = t1[0]; rk[1][1] = t1[1];
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float v0 = acc[g][h][j][0], v1 = acc[g][h][j][1], v2 = acc[g][h][j][2], v3 = acc[g][h][j][3];
                    if constexpr (RES) {
                        v0 += H16<T>::lo(rk[h][0]); v1 += H16<T>::hi(rk[h][0]);
                        v2 += H16<T>::lo(rk[h][1]); v3 += H16<T>::hi(rk[h][1]);
                    }
                    pk[h][0] = H16<T>::pack2(v0, v1); pk[h][1] = H16<T>::pack2(v2, v3);
                }
                if constexpr (NB) {
                    const u32x4 y16 = P.y[g][j];
                    const v2u y0 = __builtin_amdgcn_permlane16_swap(y16[0], y16[2], false, false);
                    const v2u y1 = __builtin_amdgcn_permlane16_swap(y16[1], y16[3], false, false);
                    const uint32_t yk[2][2] = {{y0[0], y1[0]}, {y0[1], y1[1]}};
                    if (P.vo[g][j] >= 0) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float gv[4] = {H16<T>::lo(pk[h][0]), H16<T>::hi(pk[h][0]), H16<T>::lo(pk[h][1]), H16<T>::hi(pk[h][1])};
                            const float yv[4] = {H16<T>::lo(yk[h][0]), H16<T>::hi(yk[h][0]), H16<T>::lo(yk[h][1]), H16<T>::hi(yk[h][1])};
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                float gm = gv[rr];
                                if (A.nrelu && !(fmaf(yv[rr], csc[h][rr], csh[h][rr]) > 0.f)) gm = 0.f;     // the forward pass's expression
                                nsa[h][rr] += gm;
                                nsb[h][rr] = fmaf(gm, yv[rr], nsb[h][rr]);                               // raw moment, centred at the flush
                            }
                        }
                    }
                }
                const v2u s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                const v2u s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
                // (no SGPR offset on the store: hipcc guards the "VALU overwrites the data registers of a 16-byte store" hazard only for
                //  stores without one -- conv_igemm.hip: k_ig3r, tools/scan_store_hazard.py; an out-of-range lane offset stays out of range)
                __builtin_amdgcn_raw_buffer_store_b128(v4u_t{s0[0], s1[0], s0[1], s1[1]}, ors, P.vo[g][j] + soff, 0, 0);
            }
    };

    if (ntile == 0) return;
    Tile cur = locate(pstart), nxt = cur;
    PSet SA, SB;
    issue_halo(cur);
    enter_image(cur.n);
    request(SA, cur, 0, 0);                                 // (always: it also forms the store offsets)
    for (int k = 0; k < ntile; ++k) {
        const bool has_next = k + 1 < ntile;
        if (has_next) nxt = locate(pstart + (k + 1) * G);
        // group 0 = (cd 0, ch 0)
        request(SB, cur, 0, 1);
        if constexpr (NLOAD == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // the halo pieces are older than this set
        else if constexpr (NLOAD == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                    // halo (and, first time, the weights) visible to every wave
        mfma_pair(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        epilogue(SA, cur, 0, 0);
        // group 1 = (0, 1)
        request(SA, cur, 1, 0);
        mfma_pair(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        epilogue(SB, cur, 0, 1);
        // group 2 = (1, 0)
        request(SB, cur, 1, 1);
        mfma_pair(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        epilogue(SA, cur, 1, 0);
        // group 3 = (1, 1); the next tile's first set is requested before its MFMA loop, the next halo right after it
        if (has_next) request(SA, nxt, 0, 0);
        mfma_pair(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        __syncthreads();                                    // every wave is done reading this tile's halo
        if (has_next) issue_halo(nxt);
        epilogue(SB, cur, 1, 1);
        if (has_next) { enter_image(nxt.n); cur = nxt; }
    }
    flush();
}

// launch rule: 16 bits, 32 <- 64 channels (padded), stride (2, 2, 2), 32-bit byte offsets per image
static int dgsp_on() { const char* e = getenv("NNDET_DGSP"); return e ? atoi(e) : 1; }     // (read per call: the tests compare the forms)
static bool dgsp_covers(const NndetConv* c) {
    if (!dgsp_on() || !nndet_is16(c->dtype) || c->cin_p != 32 || c->cout_p != 64) return false;
    for (int i = 0; i < 3; ++i) if (c->s[i] != 2 || c->k[i] != 3 || c->p[i] != 1) return false;
    return (int64_t)c->in_d * c->in_h * c->in_w * 64 < (1LL << 31) && (int64_t)c->out_d * c->out_h * c->out_w * 128 < (1LL << 31);
}
template <typename T, bool RES, bool NB> static int dgsp_launch(const DgspArgs& a, int grid, hipStream_t st) {
    constexpr int LDS = 108 * 1024 + 51 * 1024;
    static NndetDevOnce attr;
    if (attr.need()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dgsp<T, RES, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr.done();
    }
    k_dgsp<T, RES, NB><<<grid, 256, LDS, st>>>(a);
    LAUNCH_CHECK();
    return 0;
}
static int dgsp_run(const NndetConv* c, const void* dy, const void* w, const void* res, void* dx, hipStream_t st, const DgsNormRed* nr) {
    if (res && res != dx) return 1;                        // (the residual is read through the output's descriptor layout: in place only)
    DgspArgs a;
    memset(&a, 0, sizeof(a));
    a.dy = dy; a.w = w; a.res = res; a.dx = dx; a.N = c->batch;
    const int osp[3] = {c->out_d, c->out_h, c->out_w}, isp[3] = {c->in_d, c->in_h, c->in_w};
    const int T3[3] = {DGS_TD, DGS_TH, DGS_TW};
    for (int i = 0; i < 3; ++i) {
        if (osp[i] != (isp[i] + 2 - 3) / 2 + 1) return NNDET_EINVAL;
        a.O[i] = osp[i]; a.I[i] = isp[i];
        a.nt[i] = ceil_div((isp[i] + 1) / 2, T3[i]);
    }
    a.total_tiles = a.N * a.nt[0] * a.nt[1] * a.nt[2];
    if (nr) { a.ny = nr->y; a.nmr = nr->mean_rstd; a.ngamma = nr->gamma; a.nbeta = nr->beta; a.nred = nr->red_ws; a.nrelu = nr->relu; a.ncout = nr->c; }
    const char* ge = getenv("NNDET_DGSP_WGS");
    int grid = ge ? atoi(ge) : 256;
    if (grid < 1 || grid > 1024) grid = 256;
    if (grid > a.total_tiles) grid = a.total_tiles;
    const bool hf = c->dtype == NNDET_F16;
#define DGSP_GO(R_, N_) (hf ? dgsp_launch<f16_t, R_, N_>(a, grid, st) : dgsp_launch<bf16_t, R_, N_>(a, grid, st))
    if (res) return nr ? DGSP_GO(true, true) : DGSP_GO(true, false);
    return nr ? NNDET_EINVAL : DGSP_GO(false, false);
#undef DGSP_GO
}

// ------------------------------------------------------------------------------------------------ host side
static uint32_t dgs_magic(int d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)d + 1ull); }

int dgs_covers(const NndetConv* c) {
    static const int on = getenv("NNDET_DGS") ? atoi(getenv("NNDET_DGS")) : 1;
    if (!on || c->transposed || c->cin_p == 1) return 0;
    bool strided = false;
    for (int i = 0; i < 3; ++i) {
        if (c->k[i] != 3 || c->p[i] != 1 || (c->s[i] != 1 && c->s[i] != 2)) return 0;
        strided |= c->s[i] == 2;
    }
    if (!strided) return 0;
    const int esz = nndet_esize(c->dtype);
    const int kcb = nndet_is16(c->dtype) ? 32 : 16;
    int hv = 1;
    const int T[3] = {DGS_TD, DGS_TH, DGS_TW};
    for (int i = 0; i < 3; ++i) hv *= T[i] + (c->s[i] == 2 ? 1 : 2);
    const int64_t lds = (int64_t)hv * 64 * (c->cout_p / kcb);
    // every chunk of the halo must be resident; above 64 KB only one workgroup fits a CU and nothing hides its serial class walk
    // (measured 64 <- 128 channels, 104 KB: 0.228 ms against 0.165 ms for k_igemm)
    if (c->cout_p % kcb || lds > 64 * 1024) return 0;
    if ((int64_t)hv * 4 * (c->cout_p / kcb) > 256 * 32) return 0;          // <= 32 staged pieces per thread
    const int64_t dyb = (int64_t)c->out_d * c->out_h * c->out_w * c->cout_p * esz, dxb = (int64_t)c->in_d * c->in_h * c->in_w * c->cin_p * esz;
    return dyb < (1LL << 31) && dxb < (1LL << 31) ? 1 : 0;
}

template <typename T, int MT, int MAXP, int G, bool NB = false>
static int dgs_launch(const DgsArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    static NndetDevOnce attr;
    if (attr.need()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dgs<T, MT, MAXP, G, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        attr.done();
    }
    k_dgs<T, MT, MAXP, G, NB><<<grid, 256, lds, st>>>(a);
    LAUNCH_CHECK();
    return 0;
}

// Can the data gradient of this convolution also accumulate the norm-backward sums of the block that produced its input (DgsNormRed)?
// k_dgs launches in 16 bits with 32 rows per workgroup and the W-parity class pairs (stride 2 along W).
int dgs_fuses_norm_reduce(const NndetConv* c) {
    if (!dgs_covers(c) || !nndet_is16(c->dtype) || c->s[2] != 2) return 0;
    int hv = 1;
    const int T[3] = {DGS_TD, DGS_TH, DGS_TW};
    for (int i = 0; i < 3; ++i) hv *= T[i] + (c->s[i] == 2 ? 1 : 2);
    const int64_t lds = (int64_t)hv * 64 * (c->cout_p / 32);
    return !(c->cin_p % 64 == 0 && lds > 70 * 1024) ? 1 : 0;            // (the 64-row variant has no such epilogue)
}

// returns 1 = not covered (the caller falls back to k_igemm)
int dgs_run(const NndetConv* c, const void* dy, const void* w, const void* res, void* dx, hipStream_t st, const DgsNormRed* nr) {
    if (!dgs_covers(c)) return 1;
    if (nr && !dgs_fuses_norm_reduce(c)) return NNDET_EINVAL;
    if (dgsp_covers(c)) {                                   // the persistent form (round 6) for the full-resolution transition
        const int prc = dgsp_run(c, dy, w, res, dx, st, nr);
        if (prc != 1) return prc;
    }
    DgsArgs a;
    memset(&a, 0, sizeof(a));
    a.dy = dy; a.w = w; a.res = res; a.dx = dx;
    if (nr) { a.ny = nr->y; a.nmr = nr->mean_rstd; a.ngamma = nr->gamma; a.nbeta = nr->beta; a.nred = nr->red_ws; a.nrelu = nr->relu; a.ncout = nr->c; }
    a.N = c->batch; a.K = c->cout_p; a.R = c->cin_p;
    const int osp[3] = {c->out_d, c->out_h, c->out_w}, isp[3] = {c->in_d, c->in_h, c->in_w};
    const int T[3] = {DGS_TD, DGS_TH, DGS_TW};
    int Lmax[3], hi[3];
    // per axis and class parity: the (tap, delta) pairs with (c + p - t) % s == 0, delta = (c + p - t) / s
    int ntp[3][2], tl[3][2][3], dl[3][2][3];
    for (int i = 0; i < 3; ++i) {
        a.O[i] = osp[i]; a.I[i] = isp[i]; a.s[i] = c->s[i];
        if (osp[i] != (isp[i] + 2 - 3) / c->s[i] + 1) return NNDET_EINVAL;
        int lo = 1 << 30; hi[i] = -(1 << 30);
        for (int cc = 0; cc < c->s[i]; ++cc) {
            ntp[i][cc] = 0;
            for (int t = 0; t < 3; ++t) {
                const int num = cc + 1 - t;
                if (((num % c->s[i]) + c->s[i]) % c->s[i] != 0) continue;
                const int d = num >= 0 ? num / c->s[i] : -((-num) / c->s[i]);
                tl[i][cc][ntp[i][cc]] = t; dl[i][cc][ntp[i][cc]] = d; ++ntp[i][cc];
                if (d < lo) lo = d;
                if (d > hi[i]) hi[i] = d;
            }
        }
        a.lo[i] = lo;
        a.H[i] = T[i] + hi[i] - lo;
        Lmax[i] = (isp[i] + c->s[i] - 1) / c->s[i];       // lattice of class 0 (the largest)
        a.nt[i] = ceil_div(Lmax[i], T[i]);
    }
    const int hv = a.H[0] * a.H[1] * a.H[2];
    a.mHW = dgs_magic(a.H[2]); a.mHH = dgs_magic(a.H[1]); a.mHV4 = dgs_magic(hv * 4);
    a.chb = hv * 64;
    int ncls = 0, ntaps = 0;
    for (int cd = 0; cd < c->s[0]; ++cd) for (int ch = 0; ch < c->s[1]; ++ch) for (int cw = 0; cw < c->s[2]; ++cw) {
        DgsClass& C = a.cls[ncls++];
        const int cc[3] = {cd, ch, cw};
        for (int i = 0; i < 3; ++i) { C.c[i] = cc[i]; C.L[i] = isp[i] > cc[i] ? (isp[i] - cc[i] + c->s[i] - 1) / c->s[i] : 0; }
        C.tap0 = ntaps;
        for (int x = 0; x < ntp[0][cd]; ++x) for (int y = 0; y < ntp[1][ch]; ++y) for (int z = 0; z < ntp[2][cw]; ++z) {
            if (ntaps >= 27) return NNDET_EINVAL;
            DgsTap& t = a.taps[ntaps++];
            const int dd = dl[0][cd][x] - a.lo[0], dh = dl[1][ch][y] - a.lo[1], dw = dl[2][cw][z] - a.lo[2];
            const int trow = dd * a.H[1] + dh;
            t.toff = (trow * a.H[2] + dw) * 64;
            t.flip = (trow & 1) << 5;
            t.wt = (tl[0][cd][x] * 3 + tl[1][ch][y]) * 3 + tl[2][cw][z];
        }
        C.ntap = ntaps - C.tap0;
    }
    a.ncls = ncls;
    const int kcb = nndet_is16(c->dtype) ? 32 : 16;
    const int nchunk = a.K / kcb;
    const size_t lds = (size_t)a.chb * nchunk;
    const int pieces = ceil_div(hv * 4 * nchunk, 256);
    const int mt = (a.R % 64 == 0 && lds > 70 * 1024) ? 4 : 2;     // one workgroup per CU anyway: 64 rows per workgroup halve the staging
    const dim3 grid(a.nt[0] * a.nt[1] * a.nt[2], a.R / (mt * 16), a.N);
    const int dt = c->dtype;
    const bool g2 = c->s[2] == 2;          // classes are enumerated with the W parity fastest: (2k, 2k + 1) differ in cw only
#define DGS_T(T_, MT_, MP_) (g2 ? dgs_launch<T_, MT_, MP_, 2>(a, grid, lds, st) : dgs_launch<T_, MT_, MP_, 1>(a, grid, lds, st))
#define DGS_GO(MT_, MP_) (dt == NNDET_BF16 ? DGS_T(bf16_t, MT_, MP_) : dt == NNDET_F16 ? DGS_T(f16_t, MT_, MP_) : DGS_T(float, MT_, MP_))
    if (nr) {                     // (dgs_fuses_norm_reduce: 16 bits, mt == 2, g2)
        if (dt == NNDET_BF16) return pieces <= 16 ? dgs_launch<bf16_t, 2, 16, 2, true>(a, grid, lds, st) : dgs_launch<bf16_t, 2, 32, 2, true>(a, grid, lds, st);
        return pieces <= 16 ? dgs_launch<f16_t, 2, 16, 2, true>(a, grid, lds, st) : dgs_launch<f16_t, 2, 32, 2, true>(a, grid, lds, st);
    }
    if (pieces <= 16) return mt == 2 ? DGS_GO(2, 16) : DGS_GO(4, 16);
    return mt == 2 ? DGS_GO(2, 32) : DGS_GO(4, 32);
#undef DGS_GO
#undef DGS_T
}
