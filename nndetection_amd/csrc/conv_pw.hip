// Pointwise ("no halo") convolutions on MFMA for gfx950: 1x1x1 Conv3d (forward / data gradient) and ConvTranspose3d with
// kernel == stride (forward = one independent 1x1x1 GEMM per kernel position, data gradient = a gather over the kernel
// positions). These are the decoder's lateral and top-down convolutions (nndet/arch/decoder/base.py:216-304,391-417), which are
// HBM-bound streaming operations (16 - 100 FLOP/B) -- the generic implicit-GEMM kernel stages them through LDS with
// two workgroup barriers per tile and reached 39 % of the HBM roofline at full resolution.
//
// Here nothing is staged: a point's 32-channel chunk IS one MFMA B-fragment row (16 bytes per lane, 64 contiguous bytes per
// point), so every wave streams 16-point tiles straight from global memory into MFMA operands, U tiles in flight per wave, no
// barrier in the loop. The (small) weight matrices sit in LDS as ready-made A fragments (1 KiB per fragment, lane-contiguous:
// conflict-free ds_read_b128). Output: 4 consecutive channels per lane (8 bytes bf16 / 16 bytes fp32), + bias + residual.
//   mode 0 (scatter): loop over the LATTICE points p (1x1x1: every voxel; transposed forward: the input voxels); for each kernel
//                     position c:  y[S * p + c][r] = sum_k W_c[r][k] x[p][k]  (+ bias[r] + res[S * p + c][r])
//   mode 1 (gather) : transposed data gradient: dx[p][r] = sum_c sum_k W_c[r][k] dy[S * p + c][k]
#include "common.h"
#include "conv_common.h"

template <typename T> struct PwMma {     // the 16-bit storage types (bf16_t, f16_t)
    static constexpr int KC = 32, EPL = 8;
    typedef f32x4 acc_t;
    __device__ static __forceinline__ acc_t zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ static __forceinline__ f32x4 f32(const acc_t& c) { return c; }
    __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) { c = H16<T>::mma(a, b, c); }
    __device__ static __forceinline__ void store4(T* p, const f32x4& v) {
        uint2 u;
        u.x = H16<T>::pack2(v[0], v[1]); u.y = H16<T>::pack2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = u;
    }
    __device__ static __forceinline__ f32x4 load4(const T* p) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        return f32x4{H16<T>::lo(u.x), H16<T>::hi(u.x), H16<T>::lo(u.y), H16<T>::hi(u.y)};
    }
};
template <> struct PwMma<float> {          // the parity path accumulates in float64 (conv_igemm.hip: Mma<float>)
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
    __device__ static __forceinline__ void store4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
    __device__ static __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
};

// column sums of the streamed operand (fused bias gradient): add the EPL channel values of one fragment to per-lane sums
template <typename T> __device__ __forceinline__ void pw_accum(const u32x4& v, float* acc, float w) {   // 16-bit types
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[2 * i] += w * H16<T>::lo(v[i]);
        acc[2 * i + 1] += w * H16<T>::hi(v[i]);
    }
}
template <> __device__ __forceinline__ void pw_accum<float>(const u32x4& v, float* acc, float w) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += w * __uint_as_float(v[i]);
}
__device__ __forceinline__ float pw_row_sum(float v) {   // sum over the 16 lanes of a DPP row (= the 16 points of a tile)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

struct PwArgs {
    const void* x; const void* w; const float* bias; const void* res; void* y;
    int64_t npts;             // lattice points over the whole batch (N * L0 * L1 * L2)
    int32_t L[3], S[3];       // lattice dims per image, stride (= kernel) per axis (1,1,1 for 1x1x1)
    int32_t Cx, Cy;           // physical channels read per point / written per point
    int32_t ncls;             // kernel positions S0 * S1 * S2
    int32_t ntiles;           // ceil(npts / 16)
    uint32_t mL2, mL1, mL0;   // magic multipliers for / L[2], / L[1], / L[0] (0 = divisor 1)
    float* dbias;             // data-gradient launches: also accumulate sum_points x[p][k] (= the bias gradient: x is dY there), NULL = off
    int32_t nbias;            // logical channel count of dbias
    const float* ss;          // deferred input norm [N][Cx][2] (1x1x1 forward only), NULL = off
    int32_t ss_relu;
    int32_t tiles_per_img;    // 16-point tiles per image (ss != NULL requires the image size to be a multiple of 16 points)
};

// strided-tensor point index of lattice point p (decomposed on the fly) and class (c0, c1, c2)
__device__ __forceinline__ int64_t pw_strided_index(const PwArgs& A, int64_t p, int c0, int c1, int c2) {
    // p < 2^31 is checked on the host, so the magic-multiplier division applies
    const unsigned pp = (unsigned)p;
    const unsigned t1 = A.mL2 ? __umulhi(pp, A.mL2) : pp;
    const unsigned w = pp - t1 * (unsigned)A.L[2];
    const unsigned t2 = A.mL1 ? __umulhi(t1, A.mL1) : t1;
    const unsigned h = t1 - t2 * (unsigned)A.L[1];
    const unsigned n = A.mL0 ? __umulhi(t2, A.mL0) : t2;
    const unsigned d = t2 - n * (unsigned)A.L[0];
    const int64_t O1 = (int64_t)A.L[1] * A.S[1], O2 = (int64_t)A.L[2] * A.S[2], O0 = (int64_t)A.L[0] * A.S[0];
    return (((int64_t)n * O0 + (d * A.S[0] + c0)) * O1 + (h * A.S[1] + c1)) * O2 + (w * A.S[2] + c2);
}

// MT row tiles of 16 output channels (Cy = 16 * MT ... or a multiple handled by blockIdx.y), NKC chunks of KC input channels.
// Weights in LDS: fragment f = (cls * MT + i) * NKC + kc at byte f * 1024 + lane * 16.
template <typename T, int MT, int NKC, int MODE, int U>
__global__ __launch_bounds__(256) void k_pw(const PwArgs A) {
    using M = PwMma<T>;
    constexpr int KC = M::KC, EPL = M::EPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.y * (MT * 16);
    // ---- weights -> LDS fragments (packed layout [cls][rows_p][k_p], rows_p = Cy resp. gather: rows = Cy as well)
    {
        const T* wp = reinterpret_cast<const T*>(A.w);
        const int Kp = (MODE == 0) ? A.Cx : A.Cx;          // contraction length per class = channels read per point
        const int nfrag = A.ncls * MT * NKC;
        for (int f = wv; f < nfrag; f += 4) {
            const int kc = f % NKC, t = f / NKC, i = t % MT, c = t / MT;
            const T* src = wp + ((int64_t)c * A.Cy + row0 + i * 16 + li) * Kp + kc * KC + q * EPL;
            *reinterpret_cast<u32x4*>(smem + f * 1024 + lane * 16) = *reinterpret_cast<const u32x4*>(src);
        }
    }
    __syncthreads();
    const T* xb = reinterpret_cast<const T*>(A.x);
    T* yb = reinterpret_cast<T*>(A.y);
    const T* rb = reinterpret_cast<const T*>(A.res);
    float bia[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) bia[i][r] = A.bias ? A.bias[row0 + i * 16 + q * 4 + r] : 0.f;
    const int wave_global = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    const int S1 = A.S[1], S2 = A.S[2];

    float bsum[NKC][EPL];                    // fused bias gradient: this lane's channel sums over its points
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
        for (int e = 0; e < EPL; ++e) bsum[kc][e] = 0.f;
    const bool do_bias = A.dbias != nullptr && blockIdx.y == 0;
    float asc[NKC][EPL], ash[NKC][EPL];      // deferred input norm of this lane's channels (q * EPL .. of every chunk), image ss_n
    int ss_n = -1;
    for (int t0 = wave_global * U; t0 < A.ntiles; t0 += nwaves * U) {
        if constexpr (MODE == 0) {
            u32x4 b[U][NKC];
            int64_t pt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t p = (int64_t)(t0 + u) * 16 + li;
                pt[u] = p < A.npts ? p : -1;
                const int64_t pc = p < A.npts ? p : 0;       // clamped (valid) address, result discarded
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc) b[u][kc] = *reinterpret_cast<const u32x4*>(xb + pc * A.Cx + kc * KC + q * EPL);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t0 + u >= A.ntiles) break;               // uniform
                if (do_bias) {
                    const float wv_ = pt[u] >= 0 ? 1.f : 0.f;
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) pw_accum<T>(b[u][kc], bsum[kc], wv_);
                }
                if (A.ss) {
                    const int n_img = (t0 + u) / A.tiles_per_img;        // wave-uniform: a tile never straddles two images
                    if (n_img != ss_n) {
                        ss_n = n_img;
#pragma unroll
                        for (int kc = 0; kc < NKC; ++kc) load_affine<EPL>(A.ss, n_img, A.Cx, kc * KC + q * EPL, asc[kc], ash[kc]);
                    }
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) b[u][kc] = AffinePiece<T>::apply(b[u][kc], asc[kc], ash[kc], A.ss_relu);
                }
                for (int c = 0; c < A.ncls; ++c) {
                    f32x4 acc[MT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        typename M::acc_t ak = M::zero();
#pragma unroll
                        for (int kc = 0; kc < NKC; ++kc) {
                            const u32x4 a = *reinterpret_cast<const u32x4*>(smem + ((c * MT + i) * NKC + kc) * 1024 + lane * 16);
                            M::mma(a, b[u][kc], ak);
                        }
                        acc[i] = M::f32(ak);             // (wave-uniform control flow: PwMma<float>::f32 exchanges values between lanes)
                    }
                    const bool pv = pt[u] >= 0;
                    int64_t o = pv ? pt[u] : 0;
                    if (A.ncls > 1) {
                        const int c2 = c % S2, ct = c / S2;
                        o = pw_strided_index(A, o, ct / S1, ct % S1, c2);
                    }
                    if constexpr (sizeof(T) == 2 && MT % 2 == 0) {
                        // bf16, row tiles in pairs: ONE 16-byte store per lane and pair instead of two 8-byte ones (v_permlane16_swap: lane
                        // row q ends up with 8 consecutive channels, q = 0: 0-7, 1: 16-23, 2: 8-15, 3: 24-31; the partner lane (q ^ 1,
                        // same li) holds the same point, so it shares `pv`)
                        typedef unsigned int pw_v2u __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int i = 0; i < MT; i += 2) {
                            uint32_t pk[2][2];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                f32x4 v = acc[i + h];
                                v[0] += bia[i + h][0]; v[1] += bia[i + h][1]; v[2] += bia[i + h][2]; v[3] += bia[i + h][3];
                                if (rb && pv) {
                                    const f32x4 r4 = M::load4(rb + o * A.Cy + row0 + q * 4 + (i + h) * 16);
                                    v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
                                }
                                pk[h][0] = H16<T>::pack2(v[0], v[1]); pk[h][1] = H16<T>::pack2(v[2], v[3]);
                            }
                            const pw_v2u s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                            const pw_v2u s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                            if (pv) *reinterpret_cast<u32x4*>(yb + o * A.Cy + row0 + i * 16 + (q >> 1) * 8 + (q & 1) * 16) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                        }
                    } else if (pv) {
                        T* yo = yb + o * A.Cy + row0 + q * 4;
#pragma unroll
                        for (int i = 0; i < MT; ++i) {
                            f32x4 v = acc[i];
                            v[0] += bia[i][0]; v[1] += bia[i][1]; v[2] += bia[i][2]; v[3] += bia[i][3];
                            if (rb) {
                                const f32x4 r4 = M::load4(rb + o * A.Cy + row0 + q * 4 + i * 16);
                                v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
                            }
                            M::store4(yo + i * 16, v);
                        }
                    }
                }
            }
        } else {
            // gather: all kernel positions of a lattice point contribute to ONE output row block
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t0 + u >= A.ntiles) break;               // uniform
                const int64_t p = (int64_t)(t0 + u) * 16 + li;
                const bool ok = p < A.npts;
                const int64_t pc = ok ? p : 0;
                typename M::acc_t acc_k[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) acc_k[i] = M::zero();
                for (int c0 = 0; c0 < A.ncls; c0 += 4) {     // 4 kernel positions in flight
                    u32x4 b[4][NKC];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int c = c0 + cc < A.ncls ? c0 + cc : c0;
                        const int c2 = c % S2, ct = c / S2;
                        const int64_t o = pw_strided_index(A, pc, ct / S1, ct % S1, c2);
#pragma unroll
                        for (int kc = 0; kc < NKC; ++kc) b[cc][kc] = *reinterpret_cast<const u32x4*>(xb + o * A.Cx + kc * KC + q * EPL);
                    }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        if (c0 + cc >= A.ncls) break;        // uniform
                        if (do_bias) {
#pragma unroll
                            for (int kc = 0; kc < NKC; ++kc) pw_accum<T>(b[cc][kc], bsum[kc], ok ? 1.f : 0.f);
                        }
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int kc = 0; kc < NKC; ++kc) {
                                const u32x4 a = *reinterpret_cast<const u32x4*>(smem + (((c0 + cc) * MT + i) * NKC + kc) * 1024 + lane * 16);
                                M::mma(a, b[cc][kc], acc_k[i]);
                            }
                    }
                }
                f32x4 acc[MT];                               // (wave-uniform control flow here)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = M::f32(acc_k[i]);
                if constexpr (sizeof(T) == 2 && MT % 2 == 0) {      // paired 16-byte stores (see the scatter branch)
                    typedef unsigned int pw_v2u __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int i = 0; i < MT; i += 2) {
                        const pw_v2u s0 = __builtin_amdgcn_permlane16_swap(H16<T>::pack2(acc[i][0], acc[i][1]), H16<T>::pack2(acc[i + 1][0], acc[i + 1][1]), false, false);
                        const pw_v2u s1 = __builtin_amdgcn_permlane16_swap(H16<T>::pack2(acc[i][2], acc[i][3]), H16<T>::pack2(acc[i + 1][2], acc[i + 1][3]), false, false);
                        if (ok) *reinterpret_cast<u32x4*>(yb + p * A.Cy + row0 + i * 16 + (q >> 1) * 8 + (q & 1) * 16) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                    }
                } else if (ok) {
                    T* yo = yb + p * A.Cy + row0 + q * 4;
#pragma unroll
                    for (int i = 0; i < MT; ++i) M::store4(yo + i * 16, acc[i]);
                }
            }
        }
    }
    if (A.dbias != nullptr && blockIdx.y == 0) {       // (uniform per workgroup) waves -> LDS -> ONE atomic per channel and workgroup
        __shared__ float bred[NKC * KC];
        for (int i = tid; i < NKC * KC; i += 256) bred[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const float v = pw_row_sum(bsum[kc][e]);
                if (li == 0) atomicAdd(&bred[kc * KC + q * EPL + e], v);
            }
        __syncthreads();
        for (int i = tid; i < NKC * KC; i += 256)
            if (i < A.nbias && bred[i] != 0.f) atomicAdd(A.dbias + i, bred[i]);
    }
}

// ------------------------------------------------------------------------------------------------ host side
template <typename T, int MT, int NKC, int MODE, int U>
static int pw_launch(const PwArgs& A, int rowblocks, size_t lds, hipStream_t st) {
    static NndetDevOnce attr_done;
    if (attr_done.need()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pw<T, MT, NKC, MODE, U>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr_done.done();
    }
    // enough waves to cover the HBM latency-bandwidth product: 8 workgroups (32 waves) per CU at most, fewer for small problems
    int64_t wgs = ceil_div64(A.ntiles, 4 * U);
    if (wgs > 256 * 8) wgs = 256 * 8;
    if (wgs < 1) wgs = 1;
    k_pw<T, MT, NKC, MODE, U><<<dim3((unsigned)wgs, rowblocks), 256, lds, st>>>(A);
    LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int pw_dispatch(const PwArgs& A, int mt_total, int nkc, int mode, size_t lds_per_rowtile, hipStream_t st) {
    // row tiles per workgroup: 2 (32 rows) when the weights of more would not fit 64 KiB of LDS or the accumulators many registers
    const int MT = (mt_total % 4 == 0 && lds_per_rowtile * 4 <= 64 * 1024) ? 4 : 2;
    const int rowblocks = mt_total / MT;
    const size_t lds = lds_per_rowtile * MT;
#define PW_CASE(mt, kc)                                                                                      \
    if (MT == mt && nkc == kc)                                                                                 \
        return mode == 0 ? pw_launch<T, mt, kc, 0, 4>(A, rowblocks, lds, st) : pw_launch<T, mt, kc, 1, 2>(A, rowblocks, lds, st);
    PW_CASE(2, 1) PW_CASE(2, 2) PW_CASE(2, 4) PW_CASE(4, 1) PW_CASE(4, 2) PW_CASE(4, 4) PW_CASE(2, 8) PW_CASE(4, 8)
#undef PW_CASE
    return NNDET_EINVAL;
}

// kind: 0 forward, 1 backward-data. Returns 1 when the problem is not a pointwise one this kernel covers (caller falls back to the
// implicit-GEMM kernel), 0 on success, else an error.
// dbias (kind 1 only, may be NULL): the data-gradient launch also accumulates the bias gradient (column sums of dY)
static int pw_plan(const NndetConv* c, int kind, const void* x, const void* w, const float* bias, const void* res, void* y, float* dbias,
                   PwArgs* out, int* mt_total_o, int* nkc_o, int* mode_o, size_t* lds_o) {
    PwArgs& A = *out;
    static const int enabled = getenv("NNDET_PW") ? atoi(getenv("NNDET_PW")) : 1;
    if (!enabled) return 1;
    const bool tr = c->transposed != 0;
    for (int i = 0; i < 3; ++i) {
        if (c->p[i] != 0 || c->k[i] != c->s[i]) return 1;
        if (!tr && c->k[i] != 1) return 1;
    }
    const int esz = nndet_esize(c->dtype), KC = nndet_is16(c->dtype) ? 32 : 16;
    memset(&A, 0, sizeof(A));
    A.x = x; A.w = w; A.bias = bias; A.res = res; A.y = y;
    const int ncls = c->k[0] * c->k[1] * c->k[2];
    int mode;
    if (!tr) {                       // 1x1x1: forward and data gradient are the same streaming GEMM
        mode = 0;
        A.L[0] = c->in_d; A.L[1] = c->in_h; A.L[2] = c->in_w;
        A.S[0] = A.S[1] = A.S[2] = 1;
        A.Cx = kind == 0 ? c->cin_p : c->cout_p;
        A.Cy = kind == 0 ? c->cout_p : c->cin_p;
    } else if (kind == 0) {          // transposed forward: scatter over the kernel positions
        mode = 0;
        A.L[0] = c->in_d; A.L[1] = c->in_h; A.L[2] = c->in_w;
        for (int i = 0; i < 3; ++i) A.S[i] = c->s[i];
        A.Cx = c->cin_p; A.Cy = c->cout_p;
    } else {                         // transposed data gradient: gather
        mode = 1;
        A.L[0] = c->in_d; A.L[1] = c->in_h; A.L[2] = c->in_w;
        for (int i = 0; i < 3; ++i) A.S[i] = c->s[i];
        A.Cx = c->cout_p; A.Cy = c->cin_p;
        if (bias || res) return 1;
    }
    A.ncls = ncls;
    if (kind == 1 && dbias) { A.dbias = dbias; A.nbias = c->cout; }
    if (kind == 0 && c->in_affine) {
        const int64_t per_img = (int64_t)A.L[0] * A.L[1] * A.L[2];
        if (tr || per_img % 16 != 0) return 1;            // generic kernel (per-piece image index there)
        A.ss = c->in_affine; A.ss_relu = c->in_relu; A.tiles_per_img = (int32_t)(per_img / 16);
    }
    if (A.Cx % KC != 0 || A.Cy % 32 != 0) return 1;
    const int nkc = A.Cx / KC, mt_total = A.Cy / 16;
    if (nkc > 8) return 1;
    const size_t lds_per_rowtile = (size_t)ncls * nkc * 1024;
    if (lds_per_rowtile * 2 > 64 * 1024) return 1;      // weights of even a 32-row block do not fit: generic kernel
    A.npts = (int64_t)c->batch * A.L[0] * A.L[1] * A.L[2];
    {   // Small pyramid levels (< 50 000 lattice points per batch: levels 3 - 5 of the 160x160x96 patch) go to the implicit-GEMM kernel: here
        // every workgroup stages up to 66 KB of weight fragments for a handful of points (33 - 70 us launches for 16 us of traffic), there
        // the 64-point tiles + split-K spread the layer over the chip. Measured on the step: 13.98 -> 13.89 ms (three alternations,
        // profiles/round3_ab_pw_minpts.txt). NNDET_PW_MINPTS overrides the threshold (0: every pointwise layer stays here).
        const char* mp = getenv("NNDET_PW_MINPTS");                  // (read per call: tests flip it)
        const int64_t minpts = mp ? atoll(mp) : 50000;
        if (A.npts < minpts) return 1;
    }
    const int64_t nstrided = A.npts * ncls;
    if (A.npts >= (1LL << 31) || nstrided * (A.Cx > A.Cy ? A.Cx : A.Cy) * esz >= (1LL << 62)) return 1;
    A.ntiles = (int32_t)ceil_div64(A.npts, 16);
    auto magic = [](int d) -> uint32_t { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)d + 1ull); };
    A.mL2 = magic(A.L[2]); A.mL1 = magic(A.L[1]); A.mL0 = magic(A.L[0]);
    // the magic division n / d == umulhi(n, 2^32 / d + 1) is exact for n * d < 2^32: lattice counts here are < 2^31 / 16 at most
    if ((uint64_t)A.npts * (uint64_t)(A.L[2] > A.L[1] ? (A.L[2] > A.L[0] ? A.L[2] : A.L[0]) : (A.L[1] > A.L[0] ? A.L[1] : A.L[0])) >= (1ull << 32) && ncls > 1) return 1;
    *mt_total_o = mt_total; *nkc_o = nkc; *mode_o = mode; *lds_o = lds_per_rowtile;
    return 0;
}

int pw_run(const NndetConv* c, int kind, const void* x, const void* w, const float* bias, const void* res, void* y, hipStream_t st,
           float* dbias) {
    PwArgs A;
    int mt_total, nkc, mode;
    size_t lds;
    const int rc = pw_plan(c, kind, x, w, bias, res, y, dbias, &A, &mt_total, &nkc, &mode, &lds);
    if (rc) return rc;
#define PW_GO(T_) pw_dispatch<T_>(A, mt_total, nkc, mode, lds, st)
    return NNDET_DISPATCH_DTYPE(c->dtype, PW_GO);
}

int pw_covers(const NndetConv* c, int kind) {
    PwArgs A;
    int mt_total, nkc, mode;
    size_t lds;
    return pw_plan(c, kind, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &A, &mt_total, &nkc, &mode, &lds) == 0;
}
