// Implicit-GEMM 3D convolution on MFMA for gfx950 (forward, data gradient, transposed conv): NDHWC,
// bf16 or fp32 storage, fp32 accumulation.
//
// One kernel family covers Conv3d forward, Conv3d backward-data, ConvTranspose3d(k == s) forward and
// backward-data by describing each as a "gather GEMM" over a LATTICE of output points:
//     out[o_off + o_step * i][r] = sum_{tap t} sum_k  W_t[r][k] * in[in_step * i + delta_t][k]
//   - Conv3d forward:      lattice = output grid, in_step = stride, delta_t = t - pad, all k^3 taps
//   - Conv3d backward-data: one lattice per output parity class c in [0, s)^3 (o_off = c, o_step = s),
//                           in_step = 1, only the taps with (c + pad - t) % s == 0, delta = (c + pad - t) / s
//                           -> no wasted MFMA work on structurally-zero taps of strided convolutions
//   - ConvTranspose3d (k == s, pad 0) forward: class c has the single tap t = c with delta 0 (a pure GEMM)
//   - ConvTranspose3d backward-data: a strided conv (in_step = s) with all taps.
//
// Workgroup = 256 threads = 4 waves. The input halo tile of the workgroup's lattice tile is staged
// through LDS once per 64-byte channel chunk (32 bf16 / 16 fp32 channels) with 16-byte coalesced NDHWC
// loads (zero-filled outside the tensor = the conv padding) and is then re-used by every tap; the
// activation fragments of all taps are ds_read_b128 from that tile, the weight fragments (shared by all
// four waves, L1/L2 resident) come straight from global memory. MFMA operands: A = weights (rows = output
// channels), B = activations (cols = lattice points), so a lane ends up with 4 consecutive output
// channels of one voxel -> 8/16-byte NDHWC stores.
//   bf16: v_mfma_f32_16x16x32_bf16 (one per 16-byte fragment pair)
//   fp32: v_mfma_f32_16x16x4_f32 x4 (exact fp32, the parity path); the K slot <-> channel assignment is
//         any bijection as long as A and B agree, so both dtypes use the same 16-byte fragment geometry.
// Optional epilogue: + bias, per-(n, channel) sum / sum-of-squares of the rounded outputs (InstanceNorm /
// GroupNorm statistics fused into the producer; fp64 atomics into NNDET_STATS_REPLICAS replicas).
#include "common.h"
#include "conv_common.h"

typedef unsigned int v2u_t __attribute__((__vector_size__(8)));
typedef unsigned int v4u_t __attribute__((__vector_size__(16)));

template <typename T> struct Mma {     // the 16-bit storage types (bf16_t, f16_t)
    static constexpr int KC = 32;   // channels per 64-byte chunk
    static constexpr int EPL = 8;   // elements per 16-byte fragment
    typedef f32x4 acc_t;
    __device__ static __forceinline__ acc_t zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) { c = H16<T>::mma(a, b, c); }
    __device__ static __forceinline__ f32x4 f32(const acc_t& c) { return c; }
};
// fp32 activations = the PARITY path (north_star: outputs within 1e-4 of the reference). Round 4: products of fp32 values are exact in
// float64 and the sum runs in float64 on v_mfma_f64_16x16x4_f64, so every convolution output is the exactly accumulated sum rounded ONCE
// to fp32 -- closer to a float64 evaluation of the network than the reference's own fp32 (oneDNN) kernels are
// (tests/test_parity_full_gpu.py: *_arbitrated_by_fp64, net_*_grad64.npz). Operand layout as the fp32 MFMA (A: row = lane % 16,
// k = lane / 16; B likewise); the float64 RESULT layout differs: register r of lane (q = lane / 16, li) holds row q + 4 r
// (tools/probe_mfma64.hip, profiles/round4_probe_mfma64.txt), the fp32 one row 4 q + r. `f32()` rounds and moves the values to the
// fp32 layout the epilogues are written for (16 ds_bpermute per tile, once per output); it must be called with all 64 lanes active.
template <> struct Mma<float> {
    static constexpr int KC = 16;
    static constexpr int EPL = 4;
    typedef f64x4_t acc_t;
    __device__ static __forceinline__ acc_t zero() { return f64x4_t{0.0, 0.0, 0.0, 0.0}; }
    __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, acc_t& c) {
        const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[0], (double)fb[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[1], (double)fb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[2], (double)fb[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[3], (double)fb[3], c, 0, 0, 0);
    }
    __device__ static __forceinline__ f32x4 f32(const acc_t& c) { return f64acc_rows_to_f32(c); }
};

// stores 4 consecutive channels and returns (through a..d) the values as stored (i.e. rounded to T)
template <typename T> __device__ __forceinline__ void store4r(T* p, float& a, float& b, float& c, float& d) {   // 16-bit types
    uint2 v;
    v.x = H16<T>::pack2(a, b);
    v.y = H16<T>::pack2(c, d);
    *reinterpret_cast<uint2*>(p) = v;
    a = H16<T>::lo(v.x); b = H16<T>::hi(v.x);
    c = H16<T>::lo(v.y); d = H16<T>::hi(v.y);
}
template <> __device__ __forceinline__ void store4r<float>(float* p, float& a, float& b, float& c, float& d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

template <typename T> __device__ __forceinline__ void load4(const T* p, float* v) {   // 16-bit types
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = H16<T>::lo(u.x); v[1] = H16<T>::hi(u.x);
    v[2] = H16<T>::lo(u.y); v[3] = H16<T>::hi(u.y);
}
template <> __device__ __forceinline__ void load4<float>(const float* p, float* v) {
    const float4 f = *reinterpret_cast<const float4*>(p); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
}

// max 16-byte halo pieces per thread per chunk: 16 (halo <= 1024 voxels = 64 KiB) for unit-stride tiles,
// 24 (<= 1536 voxels = 96 KiB) for the strided configurations (template parameter MAXP)

struct IgClass {
    int32_t out_off[3];
    int32_t L[3];         // lattice size of this class
    int32_t in_base[3];   // min delta per axis (input coordinate of lattice point 0, tap offset 0)
    int32_t tap0, ntap;
};
struct IgTap { int32_t d[3]; int32_t wt; };   // delta - in_base (>= 0), weight tap index
struct IgTapX { int32_t toff; int32_t flip; int32_t woff; int32_t pad; };   // LDS byte offset of the tap, swizzle flip (0 / 32), weight byte offset wt * Cy * Cx * sizeof(T)

struct IgArgs {
    const void* x; const void* w; const float* bias; void* y; double* stats;
    const void* res;      // optional residual, same layout as y: y = conv + bias + res (decoder top-down add fused away)
    const float* ss;      // deferred input norm: [N][Cx][2] (scale, shift), NULL = read x as it is (NndetConv.in_affine)
    int32_t ss_relu;
    int32_t N;
    int32_t I[3], Cx;     // input tensor spatial dims, physical channels (= K)
    int32_t O[3], Cy;     // output tensor spatial dims, physical channels (= rows)
    int32_t in_step[3], out_step[3];
    int32_t T[3];         // lattice tile
    int32_t nt[3];        // tiles per axis
    int32_t H[3];         // halo dims (max over classes)
    uint32_t mHW, mHH;    // magic multipliers: n / H[2] == umulhi(n, mHW), n / H[1] == umulhi(n, mHH) (0 = divisor 1)
    int32_t lT1, lT2;     // log2 of the (power-of-two) tile dims T[1], T[2]
    int32_t wbytes;       // size of the packed weight tensor in bytes (buffer descriptor of the PIPE kernels)
    int32_t swz;          // 1: XOR LDS byte-offset bit 5 with bit `swzsh` of the halo row index (conflict-free ds_read_b128)
    int32_t swzsh;        // 0 for unit-stride tiles (row parity); 1 when the fragment's rows are 2 apart (stride 2 along H)
    int32_t deint, HWE;   // stride 2 along W: a halo row is stored de-interleaved, even w first (HWE of them), then odd w, so that the
                          // 16 points of a fragment (w, w+2, ...) are 64 bytes apart instead of 128 (PIPE kernels)
    int32_t ncls;
    IgClass cls[8];
    IgTap taps[27];
    IgTapX tapx[28];      // per-tap LDS offset / swizzle flip / weight byte offset, precomputed on the host (PIPE kernels)
    // Split-K (small problems, k_igemm only): blockIdx.x = tile * ksplit + ks, workgroup ks walks the channel chunks
    // [ks * nchunk / ksplit, (ks + 1) * nchunk / ksplit) and stores its raw fp32 accumulators to part[ks][n][voxel][Cy]; bias, residual,
    // rounding and the norm statistics happen in k_ig_splitk_reduce. ksplit <= 1 / part == NULL: off.
    int32_t ksplit, pad_;
    float* part;
    // Norm-backward sums in the epilogue (round 6, k_ig3<..., NB = true>, data gradient only): y is the COMPLETE gradient w.r.t. the
    // normalised (+ReLU) output of the conv -> norm -> ReLU block that produced this convolution's input, so the sums that block's
    // k_norm_bwd_reduce would read it back for -- S1 = sum g, S2 = sum g * xhat per (image, channel), g = y * [ReLU mask] -- are
    // accumulated here from the values being stored and the block's pre-norm tensor `ny` (DgsArgs of conv_dgs.hip: same fields, same sums)
    const void* ny; const float* nmr; const float* ngamma; const float* nbeta; double* nred;
    int32_t nrelu, ncout;
};

// Ragged batch (NndetItems): per-item dims and first voxel row; only k_ig3<..., ITEMS = true> reads it (blockIdx.z = item,
// blockIdx.x = tile of the LARGEST item: workgroups beyond an item's own tile count exit at once)
struct IgItems {
    int32_t n, pad_;
    int32_t dims[NNDET_MAX_ITEMS][3];
    int64_t row_off[NNDET_MAX_ITEMS];
};

// Wave layout inside the workgroup: WR waves along the output rows (channels) x 4/WR waves along the lattice points.
// A wave owns MT row tiles x NT point tiles of 16. Splitting rows across waves (WR = 2) halves the weight-fragment
// traffic from L2 (every wave used to stream the weights of ALL rows: 432 KB per workgroup and chunk, the bottleneck of
// the C >= 64 layers) at the price of twice as many (cheap, conflict-free) LDS activation reads.
// MINW = requested waves per SIMD (= workgroups per CU for 256-thread workgroups): caps the register allocation.
// PIPE (the strided configurations): the tap loop is software-pipelined and pinned with sched_barrier -- weight fragments by
// buffer loads two taps ahead (ring of 3, loop unrolled by 3: no register copies), tap descriptors (scalar loads) two taps ahead,
// the second half of a tap's activation fragments read before the MFMAs of its first half and the first half of the next tap
// before the MFMAs of the second half. Measured +20 % on the stride-2 forward convs; for 1-tap problems (1x1x1, transposed) the
// same loop was 25 % slower than the plain one, which is why it is a per-configuration choice
// (profiles/round1_micro_v8_generic_pinned_rejected.txt).
template <typename T, int WR, int MT, int NT, int MAXP, int MINW, bool PIPE = false, bool AFF = false>
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? 1 : MINW)) void k_igemm(const IgArgs A) {     // (fp32: float64 accumulators, all 512 registers)
    using M = Mma<T>;
    constexpr int KC = M::KC, EPL = M::EPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int wr = wv % WR, wc = wv / WR;     // row group / point group of this wave

    // Several classes (parity classes of a strided backward-data / the k^3 positions of a transposed conv) read the SAME
    // input tile. Workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the id is decoded as
    // (tile / 8, class, tile % 8): the classes of one tile run at the same time on the same XCD and the tile comes from HBM
    // once instead of once per class.
    const int ksp = A.part ? A.ksplit : 1;
    const int ks_i = A.part ? (int)(blockIdx.x % ksp) : 0;
    const int bx = A.part ? (int)(blockIdx.x / ksp) : (int)blockIdx.x;
    int cls_i = 0, tt = bx;
    const int n = blockIdx.z;
    if (A.ncls > 1) {
        const int g = bx >> 3;
        cls_i = g % A.ncls;
        tt = (g / A.ncls) * 8 + (bx & 7);
        if (tt >= A.nt[0] * A.nt[1] * A.nt[2]) return;   // uniform
    }
    const IgClass& C = A.cls[cls_i];
    const int tw_i = tt % A.nt[2]; tt /= A.nt[2];
    const int th_i = tt % A.nt[1];
    const int td_i = tt / A.nt[1];
    const int l0d = td_i * A.T[0], l0h = th_i * A.T[1], l0w = tw_i * A.T[2];
    if (l0d >= C.L[0] || l0h >= C.L[1] || l0w >= C.L[2]) return;   // uniform: tile outside this class' lattice
    const int row0 = blockIdx.y * (WR * MT * 16);

    const int HD = A.H[0], HH = A.H[1], HW = A.H[2];
    const int HV4 = HD * HH * HW * 4;
    const int i0d = l0d * A.in_step[0] + C.in_base[0];
    const int i0h = l0h * A.in_step[1] + C.in_base[1];
    const int i0w = l0w * A.in_step[2] + C.in_base[2];
    const T* xn = reinterpret_cast<const T*>(A.x) + (int64_t)n * A.I[0] * A.I[1] * A.I[2] * A.Cx;

    // global element offsets of this thread's halo pieces (identical for every channel chunk)
    int32_t goff[MAXP];
    int32_t pofs[PIPE ? MAXP : 1];   // PIPE kernels: LDS byte offset of piece s (de-interleaved / swizzled layout); else p * 16 ^ swizzle
    uint32_t swmask = 0;   // bit s: parity of the halo row of piece s (LDS swizzle)
#pragma unroll
    for (int s = 0; s < MAXP; ++s) {
        const int p = tid + s * 256;
        int32_t o = -1;
        if constexpr (PIPE) pofs[s] = p * 16;
        if (p < HV4) {
            // no integer division on the GPU: magic-multiplier division by the (runtime) halo dims
            const int hv = p >> 2, part = p & 3;
            const int t2 = A.mHW ? (int)__umulhi((unsigned)hv, A.mHW) : hv;
            const int hw = hv - t2 * HW;
            const int hd = A.mHH ? (int)__umulhi((unsigned)t2, A.mHH) : t2;
            const int hh = t2 - hd * HH;
            swmask |= (uint32_t)((t2 >> A.swzsh) & A.swz) << s;
            if constexpr (PIPE) {
                const int hwp = A.deint ? (hw & 1) * A.HWE + (hw >> 1) : hw;
                pofs[s] = (((t2 * HW + hwp) * 4 + part) * 16) ^ ((((t2 >> A.swzsh) & A.swz)) << 5);
            }
            const int id = i0d + hd, ih = i0h + hh, iw = i0w + hw;
            if ((unsigned)id < (unsigned)A.I[0] && (unsigned)ih < (unsigned)A.I[1] && (unsigned)iw < (unsigned)A.I[2])
                o = ((id * A.I[1] + ih) * A.I[2] + iw) * A.Cx + part * EPL;
        }
        goff[s] = o;
    }
    // LDS byte offsets of this lane's lattice points (tap offset 0) for its NT B-fragments
    int boff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int p = (wc * NT + j) * 16 + li;
        const int pw = p & (A.T[2] - 1);
        const int t2 = p >> A.lT2;
        const int ph = t2 & (A.T[1] - 1), pd = t2 >> A.lT1;
        const int brow = (pd * A.in_step[0]) * HH + ph * A.in_step[1];
        boff[j] = ((brow * HW + pw * (A.deint ? 1 : A.in_step[2])) * 64 + q * 16) ^ (((brow >> A.swzsh) & A.swz) << 5);
    }
    typename M::acc_t acc_k[MT][NT];          // (float64 for fp32 activations, see Mma<float>)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc_k[i][j] = M::zero();

    const T* wl = reinterpret_cast<const T*>(A.w) + (int64_t)(row0 + wr * MT * 16 + li) * A.Cx + q * EPL;
    const int nchunk_all = A.Cx / KC;
    const int kc_begin = ks_i * nchunk_all / ksp, nchunk = (ks_i + 1) * nchunk_all / ksp;      // this workgroup's share of the chunks
#ifndef IGP_DBG
#define IGP_DBG 0       // timing experiments only (wrong results): 1 no weight loads in the tap loop, 2 no halo loads, 4 no LDS fragment reads
#endif
    if constexpr (PIPE) {
        const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A.w), 0, A.wbytes, 0x00020000);
        int voff[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) voff[i] = ((row0 + (wr * MT + i) * 16 + li) * A.Cx + q * EPL) * (int)sizeof(T);
        const IgTapX* const tx = A.tapx + C.tap0;
        const int last = C.ntap - 1;
        constexpr int NH = NT / 2;
        for (int kc = kc_begin; kc < nchunk; ++kc) {
            __syncthreads();
            float asc[EPL], ash[EPL];          // deferred input norm of this thread's 16-byte part of the chunk (AFF variants only)
            if constexpr (AFF) load_affine<EPL>(A.ss, n, A.Cx, kc * KC + (tid & 3) * EPL, asc, ash);
            auto aff = [&](const u32x4& v_) { if constexpr (AFF) return AffinePiece<T>::apply(v_, asc, ash, A.ss_relu); else return v_; };
#pragma unroll
            for (int s0 = 0; s0 < MAXP; s0 += 8) {
                if (s0 * 256 >= HV4) break;   // uniform
                u32x4 v[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int o = goff[s0 + b];
                    if (IGP_DBG & 2) v[b] = u32x4{(unsigned)o, 1u, 2u, 3u}; else
                    v[b] = *reinterpret_cast<const u32x4*>(xn + (o < 0 ? 0 : o) + kc * KC);
                }
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int p = tid + (s0 + b) * 256;
                    if (p < HV4)
                        *reinterpret_cast<u32x4*>(smem + pofs[s0 + b]) = goff[s0 + b] < 0 ? u32x4{0u, 0u, 0u, 0u} : aff(v[b]);
                }
            }
            __syncthreads();
            if (last < 0) continue;
            const int kcoff = kc * KC * (int)sizeof(T);
            IgTapX cur = tx[0], nxt = tx[min(1, last)];
            u32x4 afr[3][MT], bf0[NH], bf1[NT - NH];
            auto load_w = [&](const IgTapX& t, u32x4* a_) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (IGP_DBG & 1) a_[i] = u32x4{(unsigned)t.woff, (unsigned)voff[i], 2u, 3u}; else
                    a_[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, voff[i], t.woff + kcoff, 0));
                }
            };
            auto lds_h0 = [&](const IgTapX& t) {
#pragma unroll
                for (int j = 0; j < NH; ++j) {
                    if (IGP_DBG & 4) bf0[j] = u32x4{(unsigned)(boff[j] ^ t.flip), (unsigned)t.toff, 2u, 3u}; else
                    bf0[j] = *reinterpret_cast<const u32x4*>(smem + ((boff[j] ^ t.flip) + t.toff));
                }
            };
            auto lds_h1 = [&](const IgTapX& t) {
#pragma unroll
                for (int j = NH; j < NT; ++j) {
                    if (IGP_DBG & 4) bf1[j - NH] = u32x4{(unsigned)(boff[j] ^ t.flip), (unsigned)t.toff, 2u, 3u}; else
                    bf1[j - NH] = *reinterpret_cast<const u32x4*>(smem + ((boff[j] ^ t.flip) + t.toff));
                }
            };
            load_w(cur, afr[0]);
            load_w(nxt, afr[1]);
            lds_h0(cur);
            for (int tp = 0; tp <= last; tp += 3) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    if (tp + u > last) break;          // uniform
                    const IgTapX nn = tx[min(tp + u + 2, last)];
                    load_w(nn, afr[(u + 2) % 3]);      // (clamped: the last taps re-load their own fragments, harmless)
                    lds_h1(cur);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NH; ++j) M::mma(afr[u][i], bf0[j], acc_k[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    lds_h0(nxt);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = NH; j < NT; ++j) M::mma(afr[u][i], bf1[j - NH], acc_k[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    cur = nxt; nxt = nn;
                }
            }
        }
    } else
    for (int kc = kc_begin; kc < nchunk; ++kc) {
        __syncthreads();
        float asc[EPL], ash[EPL];              // deferred input norm of this thread's 16-byte part of the chunk (AFF variants only)
        if constexpr (AFF) load_affine<EPL>(A.ss, n, A.Cx, kc * KC + (tid & 3) * EPL, asc, ash);
        auto aff = [&](const u32x4& v_) { if constexpr (AFF) return AffinePiece<T>::apply(v_, asc, ash, A.ss_relu); else return v_; };
        // Stage the halo in batches of 8 UNCONDITIONAL 16-byte loads per thread (out-of-tensor pieces load a clamped,
        // valid address and are zeroed afterwards): all loads of a batch are in flight together. A conditional load per
        // piece makes hipcc branch + wait vmcnt(0) per piece, i.e. 16-24 serial HBM round trips per chunk.
#pragma unroll
        for (int s0 = 0; s0 < MAXP; s0 += 8) {
            if (s0 * 256 >= HV4) break;   // uniform
            u32x4 v[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int o = goff[s0 + b];
                v[b] = *reinterpret_cast<const u32x4*>(xn + (o < 0 ? 0 : o) + kc * KC);
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int p = tid + (s0 + b) * 256;
                if (p < HV4)
                    *reinterpret_cast<u32x4*>(smem + ((p * 16) ^ (((swmask >> (s0 + b)) & 1u) << 5))) =
                        goff[s0 + b] < 0 ? u32x4{0u, 0u, 0u, 0u} : aff(v[b]);
            }
        }
        __syncthreads();
        // software pipeline over the taps: the weight fragments of tap tp+1 (L1/L2 latency) are in flight while the
        // MFMAs of tap tp issue; the activation fragments (LDS latency) are read at the top of their own tap
        u32x4 af[MT], afn[MT], bf[NT];
        auto load_w = [&](int tp, u32x4* a_) {
            const T* wt = wl + ((int64_t)A.taps[C.tap0 + tp].wt * A.Cy) * A.Cx + kc * KC;
#pragma unroll
            for (int i = 0; i < MT; ++i) a_[i] = *reinterpret_cast<const u32x4*>(wt + (int64_t)i * 16 * A.Cx);
        };
        if (C.ntap > 0) load_w(0, afn);
        for (int tp = 0; tp < C.ntap; ++tp) {
            const IgTap& tap = A.taps[C.tap0 + tp];
            const int trow = tap.d[0] * HH + tap.d[1];
            const int toff = (trow * HW + tap.d[2]) * 64;
            const int flip = ((trow >> A.swzsh) & A.swz) << 5;      // row parity of the tap flips the swizzle bit
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const u32x4*>(smem + ((boff[j] ^ flip) + toff));
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = afn[i];
            if (tp + 1 < C.ntap) load_w(tp + 1, afn);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) M::mma(af[i], bf[j], acc_k[i][j]);
        }
    }

    // ---------------- epilogue: lane holds rows row0 + i*16 + q*4 + {0..3} of voxel (tile j, li)
    f32x4 acc[MT][NT];                              // (uniform control flow here: Mma<float>::f32 exchanges values between lanes)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = M::f32(acc_k[i][j]);
    if (A.part) {                                  // split-K: raw partial sums, everything else in k_ig_splitk_reduce (uniform branch)
        float* pb = A.part + ((int64_t)ks_i * A.N + n) * A.O[0] * A.O[1] * A.O[2] * A.Cy;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int pp = (wc * NT + j) * 16 + li;
            const int pt2 = pp >> A.lT2;
            const int ld = l0d + (pt2 >> A.lT1), lh = l0h + (pt2 & (A.T[1] - 1)), lw = l0w + (pp & (A.T[2] - 1));
            if ((ld < C.L[0]) && (lh < C.L[1]) && (lw < C.L[2])) {
                const int od = C.out_off[0] + ld * A.out_step[0], oh = C.out_off[1] + lh * A.out_step[1], ow = C.out_off[2] + lw * A.out_step[2];
                float* po = pb + (((int64_t)od * A.O[1] + oh) * A.O[2] + ow) * A.Cy + row0 + wr * MT * 16 + q * 4;
#pragma unroll
                for (int i = 0; i < MT; ++i) *reinterpret_cast<f32x4*>(po + i * 16) = acc[i][j];
            }
        }
        return;
    }
    T* yb = reinterpret_cast<T*>(A.y);
    float ssum[MT][4], ssq[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        // lattice coordinates of this lane's point (recomputed here instead of living in 3*NT registers through the main loop:
        // the saved registers buy a third wave per SIMD for latency hiding)
        const int pp = (wc * NT + j) * 16 + li;
        const int pt2 = pp >> A.lT2;
        const int ld = l0d + (pt2 >> A.lT1), lh = l0h + (pt2 & (A.T[1] - 1)), lw = l0w + (pp & (A.T[2] - 1));
        const bool valid = (ld < C.L[0]) && (lh < C.L[1]) && (lw < C.L[2]);
        const int od = C.out_off[0] + ld * A.out_step[0];
        const int oh = C.out_off[1] + lh * A.out_step[1];
        const int ow = C.out_off[2] + lw * A.out_step[2];
        T* yo = yb + ((((int64_t)n * A.O[0] + od) * A.O[1] + oh) * A.O[2] + ow) * A.Cy;
        if constexpr (sizeof(T) == 2 && MT == 2) {
            // one 16-byte store per lane instead of two 8-byte ones: see k_ig3 (v_permlane16_swap; partner lane = same point)
            uint32_t pk[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r0 = row0 + (wr * MT + i) * 16 + q * 4;
                float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
                if (A.bias) { v0 += A.bias[r0]; v1 += A.bias[r0 + 1]; v2 += A.bias[r0 + 2]; v3 += A.bias[r0 + 3]; }
                if (A.res && valid) {
                    float r4[4];
                    load4<T>(reinterpret_cast<const T*>(A.res) + (yo - yb) + r0, r4);
                    v0 += r4[0]; v1 += r4[1]; v2 += r4[2]; v3 += r4[3];
                }
                pk[i][0] = H16<T>::pack2(v0, v1); pk[i][1] = H16<T>::pack2(v2, v3);
                if (A.stats && valid) {
                    v0 = H16<T>::lo(pk[i][0]); v1 = H16<T>::hi(pk[i][0]);
                    v2 = H16<T>::lo(pk[i][1]); v3 = H16<T>::hi(pk[i][1]);
                    ssum[i][0] += v0; ssum[i][1] += v1; ssum[i][2] += v2; ssum[i][3] += v3;
                    ssq[i][0] += v0 * v0; ssq[i][1] += v1 * v1; ssq[i][2] += v2 * v2; ssq[i][3] += v3 * v3;
                }
            }
            const v2u_t s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
            const v2u_t s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
            if (valid) *reinterpret_cast<u32x4*>(yo + row0 + wr * MT * 16 + (q >> 1) * 8 + (q & 1) * 16) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        } else if (valid) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r0 = row0 + (wr * MT + i) * 16 + q * 4;
                float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
                if (A.bias) { v0 += A.bias[r0]; v1 += A.bias[r0 + 1]; v2 += A.bias[r0 + 2]; v3 += A.bias[r0 + 3]; }
                if (A.res) {
                    float r4[4];
                    load4<T>(reinterpret_cast<const T*>(A.res) + (yo - yb) + r0, r4);
                    v0 += r4[0]; v1 += r4[1]; v2 += r4[2]; v3 += r4[3];
                }
                store4r<T>(yo + r0, v0, v1, v2, v3);
                if (A.stats) {
                    ssum[i][0] += v0; ssum[i][1] += v1; ssum[i][2] += v2; ssum[i][3] += v3;
                    ssq[i][0] += v0 * v0; ssq[i][1] += v1 * v1; ssq[i][2] += v2 * v2; ssq[i][3] += v3 * v3;
                }
            }
        }
    }
    if (A.stats) {
        // reduce over the 16 voxel lanes (li), then over the 4 waves through LDS, then one fp64 atomic per row
        __syncthreads();                       // all waves are done reading the halo tile
        double* red = reinterpret_cast<double*>(smem);   // [WR*MT*16][2]
        if (tid < WR * MT * 16 * 2) red[tid] = 0.0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = ssum[i][r], s2 = ssq[i][r];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); s2 += __shfl_xor(s2, o, 64); }
                if (li == 0) {
                    atomicAdd(&red[((wr * MT + i) * 16 + q * 4 + r) * 2 + 0], (double)s);
                    atomicAdd(&red[((wr * MT + i) * 16 + q * 4 + r) * 2 + 1], (double)s2);
                }
            }
        __syncthreads();
        if (tid < WR * MT * 16 * 2) {
            const int rep = (blockIdx.x + blockIdx.z * 7) % NNDET_STATS_REPLICAS;
            double* dst = A.stats + (((int64_t)rep * A.N + n) * A.Cy + row0 + (tid >> 1)) * 2 + (tid & 1);
            atomicAdd(dst, red[tid]);
        }
    }
}

// Second stage of a split-K launch: y = round(sum_ks part[ks] + bias + residual), norm statistics of the rounded values.
// grid (ceil(voxels / 32), Cy / 32, N), block 256 = 32 voxels x 8 groups of 4 channels. The partial sums are added in ks order.
template <typename T>
__global__ __launch_bounds__(256) void k_ig_splitk_reduce(const float* __restrict__ part, int ksplit, int N, int64_t vox, int Cy,
                                                          const float* __restrict__ bias, const T* __restrict__ res, T* __restrict__ y,
                                                          double* __restrict__ stats) {
    __shared__ float red[4][32][2];
    const int tid = threadIdx.x, cg = tid & 7, vl = tid >> 3, wv = tid >> 6;
    const int n = blockIdx.z, c = blockIdx.y * 32 + cg * 4;
    const int64_t v = (int64_t)blockIdx.x * 32 + vl;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (v < vox) {
        const int64_t e = ((int64_t)n * vox + v) * Cy + c;
        f32x4 a = *reinterpret_cast<const f32x4*>(part + e);
        for (int ks = 1; ks < ksplit; ++ks) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(part + (int64_t)ks * N * vox * Cy + e);
            a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
        }
        v0 = a[0]; v1 = a[1]; v2 = a[2]; v3 = a[3];
        if (bias) { v0 += bias[c]; v1 += bias[c + 1]; v2 += bias[c + 2]; v3 += bias[c + 3]; }
        if (res) {
            float r4[4];
            load4<T>(res + e, r4);
            v0 += r4[0]; v1 += r4[1]; v2 += r4[2]; v3 += r4[3];
        }
        store4r<T>(y + e, v0, v1, v2, v3);          // returns the values as stored
    }
    if (stats) {
        // DETERMINISTIC statistics: fixed shuffle tree over the 8 voxels of a wave (lanes cg + 8 i), the four waves in order, one fp64
        // atomic per (block, channel) like the convolution epilogues (voxels beyond the tensor contribute exact zeros)
        float sv[8] = {v0, v1, v2, v3, v0 * v0, v1 * v1, v2 * v2, v3 * v3};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = sv[k];
            t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
            sv[k] = t;
        }
        if ((tid & 63) < 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { red[wv][cg * 4 + r][0] = sv[r]; red[wv][cg * 4 + r][1] = sv[4 + r]; }
        }
        __syncthreads();
        if (tid < 64) {
            const int ch = tid >> 1, k = tid & 1;
            const double tot = ((double)red[0][ch][k] + (double)red[1][ch][k]) + ((double)red[2][ch][k] + (double)red[3][ch][k]);
            const int rep = (blockIdx.x + blockIdx.z * 7) % NNDET_STATS_REPLICAS;
            atomicAdd(stats + (((int64_t)rep * N + n) * Cy + blockIdx.y * 32 + ch) * 2 + k, tot);
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3x3, stride 1
// Specialisation for the layers that carry most of the FLOPs: 3x3x3 / stride 1 / pad 1 convolutions, forward and
// backward-data (the same gather with the taps mirrored). The generic kernel spends 3.7 VALU + 1.8 SALU instructions per
// MFMA on LDS / weight address arithmetic with run-time tile dims (profiles/round1_pmc_v3_e1_conv.txt). Here the tile is
// a compile-time TD x 8 x 8 block (halo (TD+2) x 10 x 10), the 27 taps are fully unrolled, and
//   * every activation fragment is ONE ds_read_b128 with an immediate offset from one of two per-lane base registers
//     (the two swizzle phases: bit 5 of the byte address is XORed with the parity of the halo row, and since HH = 10 is
//     even that parity is (lane row parity) ^ (tap b parity) -- a compile-time choice between the two bases);
//   * every weight fragment is a global_load with a SCALAR base (tap, chunk) + a per-lane 32-bit offset;
//   * the InstanceNorm / GroupNorm statistics are reduced over the 16 voxel lanes with DPP adds instead of ds_bpermute.
// Tile <-> lane mapping: point p = (wc * NT + j) * 16 + li, pw = li & 7, ph = 2 * (j & 3) + (li >> 3),
// pd = wc * NT / 4 + (j >> 2).
// 4 consecutive channels through a buffer descriptor (voffset = lane byte offset, soffset = scalar byte offset); the store
// returns the values as stored (rounded to T) like store4r
template <typename T> __device__ __forceinline__ void buf_store4r(__amdgpu_buffer_rsrc_t rs, int vo, int so, float& a, float& b, float& c, float& d) {
    v2u_t v;                                                                                        // 16-bit types
    v[0] = H16<T>::pack2(a, b);
    v[1] = H16<T>::pack2(c, d);
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, vo, so, 0);
    a = H16<T>::lo(v[0]); b = H16<T>::hi(v[0]);
    c = H16<T>::lo(v[1]); d = H16<T>::hi(v[1]);
}
template <> __device__ __forceinline__ void buf_store4r<float>(__amdgpu_buffer_rsrc_t rs, int vo, int so, float& a, float& b, float& c, float& d) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, f32x4{a, b, c, d}), rs, vo, so, 0);
}
template <typename T> __device__ __forceinline__ void buf_load4(__amdgpu_buffer_rsrc_t rs, int vo, int so, float* v) {   // 16-bit types
    const v2u_t u = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);
    v[0] = H16<T>::lo(u[0]); v[1] = H16<T>::hi(u[0]);
    v[2] = H16<T>::lo(u[1]); v[3] = H16<T>::hi(u[1]);
}
template <> __device__ __forceinline__ void buf_load4<float>(__amdgpu_buffer_rsrc_t rs, int vo, int so, float* v) {
    const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
    v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
}

__device__ __forceinline__ float dpp_row_sum(float v) {   // sum over the 16 lanes of a DPP row, result in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}

// one wave-instruction of LDS-DMA: 64 lanes x 16 bytes from buffer offsets `voff` (out-of-range offsets write zeros: the conv padding)
// to the 1 KB at LDS byte address lds_dst (wave-uniform). hipcc neither preserves M0 around asm nor counts the load in its vmcnt
// bookkeeping: M0 is written in the same statement, and the completion is waited for by hand (s_waitcnt vmcnt(0) before the barrier).
__device__ __forceinline__ void ig3_lds_dma16(__amdgpu_buffer_rsrc_t rs, int voff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}

// DMA (128-row configuration, ONE workgroup per CU): the halo of channel chunk kc + 1 goes global -> LDS by LDS-DMA into the second
// 64 KB buffer WHILE the taps of chunk kc are multiplied -- one 1 KB piece per tap, issued from inside the tap loop (vmcnt is in-order:
// pieces issued in one go in front of a tap's weight loads would be waited for together with them), weight fragments 4 taps ahead.
template <typename T, int WR, int MT, int NT, int MINW, bool AFF = false, bool ITEMS = false, bool DMA = false, bool NB = false>
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? 1 : MINW)) void k_ig3(const IgArgs A, const IgItems IT) {
    using M = Mma<T>;
    static_assert(!NB || (sizeof(T) == 2 && MT == 2 && !ITEMS), "the norm-backward epilogue: 16-bit types, two row tiles per wave, uniform batches");
    constexpr int NBR = WR * MT * 16;                  // rows (= channels of the gradient) of this workgroup
    __shared__ float nb_c[NB ? NBR : 1][4];          // per channel: mean, rstd, scale, shift of the norm whose backward sums are taken
    __shared__ double nb_s[NB ? NBR : 1][2];         // S1, raw sum g * y of this workgroup
    constexpr int KC = M::KC, EPL = M::EPL;
    constexpr int TD = NT / WR, TH = 8, TW = 8;
    constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HV4 = HD * HH * HW * 4;
    // staging: 6 halo rows (of HW * 4 = 40 16-byte pieces) per step, 240 of the 256 threads active. A thread keeps its
    // column piece and walks down the rows, so per piece only the row decomposition is computed, the LDS destination is
    // dst0 + s * 3840 (an immediate) and the row parity of the swizzle is the same for every step (6 is even).
    constexpr int ROWP = HW * 4, RPS = 256 / ROWP, NROW = HD * HH, MAXP = (NROW + RPS - 1) / RPS;
    static_assert(NT % 4 == 0 && (4 / WR) * NT * 16 == TD * TH * TW && HH % 2 == 0 && RPS % 2 == 0, "tile <-> lane mapping");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int wr = wv % WR, wc = wv / WR;

    const int n = blockIdx.z;
    // spatial size of this workgroup's volume: the launch's (uniform batch) or the item's own (ragged batch: input == output size)
    int I0 = A.I[0], I1 = A.I[1], I2 = A.I[2], O0 = A.O[0], O1 = A.O[1], O2 = A.O[2], nt1 = A.nt[1], nt2 = A.nt[2];
    int tt;
    int64_t row_off = 0;
    if constexpr (ITEMS) {
        I0 = O0 = IT.dims[n][0]; I1 = O1 = IT.dims[n][1]; I2 = O2 = IT.dims[n][2];
        nt1 = (I1 + TH - 1) / TH; nt2 = (I2 + TW - 1) / TW;
        tt = blockIdx.x;
        if (tt >= ((I0 + TD - 1) / TD) * nt1 * nt2) return;      // uniform
        row_off = IT.row_off[n];
    } else {
        tt = xcd_compact(blockIdx.x, gridDim.x, 512);
    }
    const int tw_i = tt % nt2; tt /= nt2;
    const int th_i = tt % nt1;
    const int td_i = tt / nt1;
    const int l0d = td_i * TD, l0h = th_i * TH, l0w = tw_i * TW;
    const int row0 = blockIdx.y * (WR * MT * 16);
    const int img_bytes = I0 * I1 * I2 * A.Cx * (int)sizeof(T);     // < 2^31 (checked on the host)
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(A.x)) + (ITEMS ? row_off * A.Cx * (int64_t)sizeof(T) : (int64_t)n * img_bytes),
        0, img_bytes, 0x00020000);

    int32_t goff[MAXP];
    const int cp = tid % ROWP, sr0 = tid / ROWP;       // column piece (voxel hw = cp >> 2, 16-byte part cp & 3), first row
    const bool st_active = tid < RPS * ROWP;
    const int st_dst0 = ((sr0 * ROWP + cp) * 16) ^ ((sr0 & 1) << 5);
    {
        const int iw = l0w - 1 + (cp >> 2);
        const bool okw = st_active && (unsigned)iw < (unsigned)I2;
        const int rowb = I2 * A.Cx * (int)sizeof(T);
        const int colb = iw * A.Cx * (int)sizeof(T) + (cp & 3) * 16;
#pragma unroll
        for (int s = 0; s < MAXP; ++s) {
            const int r = sr0 + s * RPS;
            const int hd = r / HH, hh = r - hd * HH;
            const int id = l0d - 1 + hd, ih = l0h - 1 + hh;
            // out-of-tensor pieces (= the zero padding) get an offset beyond num_records: the buffer load returns 0 for them
            const bool ok = okw && (unsigned)id < (unsigned)I0 && (unsigned)ih < (unsigned)I1 && (s + 1 < MAXP || r < NROW);
            goff[s] = ok ? (id * I1 + ih) * rowb + colb : (int32_t)0x80000000;
        }
    }
    const int par = (li >> 3) & 1;
    const int lanevox = (wc * (NT / 4) * HH + (li >> 3)) * HW + (li & 7);
    const int sb0off = lanevox * 64 + ((q ^ (par << 1)) << 4);      // the two swizzle phases of this lane's base address
    const int sb1off = lanevox * 64 + ((q ^ (par << 1) ^ 2) << 4);

    typename M::acc_t acc_k[MT][NT];          // (float64 for fp32 activations, see Mma<float>)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc_k[i][j] = M::zero();

    // weights: buffer loads (SRD of the packed weight tensor in SGPRs) = per-lane 32-bit offset of each of the MT fragments
    // + a SCALAR offset per (tap, chunk): no per-tap vector address arithmetic, no 64-bit lane addresses to keep live
    const bool rev = A.taps[0].d[0] != 0;                    // backward-data: the tap at halo offset d is weight tap 26 - idx(d)
    const int tap_bytes = A.Cy * A.Cx * (int)sizeof(T);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A.w), 0, 27 * tap_bytes, 0x00020000);
    const int tap0_off = rev ? 26 * tap_bytes : 0, tap_step = rev ? -tap_bytes : tap_bytes;
    int voff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) voff[i] = ((row0 + (wr * MT + i) * 16 + li) * A.Cx + q * EPL) * (int)sizeof(T);

    const int nchunk = A.Cx / KC;
    // DMA staging: wave wv moves pieces wv * NPC .. wv * NPC + NPC - 1 (1 KB each); LDS granule G = piece * 64 + lane holds halo row
    // G / ROWP, voxel (G % ROWP) / 4, 16-byte part ((G % 4) ^ (row parity << 1)) -- the layout the VGPR staging below produces, with the
    // swizzle applied to the SOURCE address because the LDS-DMA destination is lane-linear
    constexpr int NPC = DMA ? (HV4 + 255) / 256 : 1;
    constexpr int HBUF = 65536;
    static_assert(!DMA || (HV4 * 16 <= HBUF && !AFF && sizeof(T) == 2), "LDS-DMA staging: 16-bit types, halo <= 64 KB, no affine on load");
    int poff[NPC];
    if constexpr (DMA) {
        const int vox_bytes = A.Cx * (int)sizeof(T);
#pragma unroll
        for (int p = 0; p < NPC; ++p) {
            const int G = (wv * NPC + p) * 64 + lane;
            const int row = G / ROWP, col = G - row * ROWP;
            const int hd = row / HH, hh = row - hd * HH, hw = col >> 2, part = (col & 3) ^ ((row & 1) << 1);
            const int id = l0d - 1 + hd, ih = l0h - 1 + hh, iw = l0w - 1 + hw;
            const bool ok = G < HV4 && (unsigned)id < (unsigned)I0 && (unsigned)ih < (unsigned)I1 && (unsigned)iw < (unsigned)I2;
            poff[p] = ok ? ((id * I1 + ih) * I2 + iw) * vox_bytes + part * 16 : (int32_t)0x80000000;
        }
    }
    auto dma_piece = [&](int p, int kc_, int buf) {
        if constexpr (DMA)
            ig3_lds_dma16(xrs, poff[p] < 0 ? poff[p] : poff[p] + kc_ * KC * (int)sizeof(T), (uint32_t)(buf * HBUF + (wv * NPC + p) * 1024));
    };
    if constexpr (DMA) {
#pragma unroll
        for (int p = 0; p < NPC; ++p) dma_piece(p, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int kc = 0; kc < nchunk; ++kc) {
        if constexpr (!DMA) __syncthreads();
        float asc[EPL], ash[EPL];              // deferred input norm: this thread stages part (cp & 3) of every halo voxel (AFF variants)
        if constexpr (AFF) load_affine<EPL>(A.ss, n, A.Cx, kc * KC + (cp & 3) * EPL, asc, ash);
        if constexpr (!DMA) {
#pragma unroll
        for (int s0 = 0; s0 < MAXP; s0 += 8) {
            u32x4 v[8];
#pragma unroll
            for (int b = 0; b < 8; ++b)
                if (s0 + b < MAXP)
                    v[b] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, goff[s0 + b], kc * KC * (int)sizeof(T), 0));
            if (st_active) {
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (s0 + b < MAXP && (s0 + b + 1 < MAXP || sr0 + (s0 + b) * RPS < NROW)) {
                        // out-of-tensor pieces (offset >= 2^31: the hardware returned zeros) are the conv padding: they stay zero
                        if constexpr (AFF) v[b] = goff[s0 + b] < 0 ? u32x4{0u, 0u, 0u, 0u} : AffinePiece<T>::apply(v[b], asc, ash, A.ss_relu);
                        *reinterpret_cast<u32x4*>(smem + st_dst0 + (s0 + b) * (RPS * ROWP * 16)) = v[b];
                    }
            }
        }
        __syncthreads();
        }
        const char* const hbase = smem + (DMA ? (kc & 1) * HBUF : 0);       // the halo buffer of this chunk
        const int wk = tap0_off + kc * KC * (int)sizeof(T);
        // Weight fragments are prefetched WD taps ahead into a ring of WD + 1 register sets. hipcc's scheduler sinks such
        // loads down to their first use (then every tap waits a full L2 round trip with vmcnt(0)), so the issue point is
        // pinned with sched_barrier: loads of tap tp + WD, barrier, LDS reads + MFMAs of tap tp.
        constexpr int WD = DMA ? 4 : 2;
        u32x4 af[WD + 1][MT];
        constexpr int BORD[3] = {0, 1, 2};
        auto tap_off = [&](int tp) { return wk + ((tp / 9 * 3 + BORD[tp % 3]) * 3 + (tp / 3) % 3) * tap_step; };
#pragma unroll
        for (int t0 = 0; t0 < WD; ++t0)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[t0][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, voff[i], tap_off(t0), 0));
        // Activation fragments: one step (4 point tiles) ahead, double buffered, also pinned: the LDS reads of step h + 1 are
        // issued before the MT * 4 MFMAs of step h, which cover their latency.
        constexpr int SPT = NT / 4, NSTEP = 27 * SPT;
        u32x4 bf[2][4];
        auto lds_step = [&](int h, u32x4* dst) {
            const int tp = h / SPT, j0 = (h % SPT) * 4;
            const int a = tp / 9, c = (tp / 3) % 3, b = BORD[tp % 3];
            const char* sb = hbase + ((b & 1) ? sb1off : sb0off);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j0 + jj;
                dst[jj] = *reinterpret_cast<const u32x4*>(sb + ((((j >> 2) + a) * HH + 2 * (j & 3) + b) * HW + c) * 64);
            }
        };
        lds_step(0, bf[0]);
#pragma unroll
        for (int h = 0; h < NSTEP; ++h) {
            const int tp = h / SPT;
#ifndef IG3_DBG
#define IG3_DBG 0       // timing experiments only (wrong results): 1 no weight loads in the tap loop, 2 no LDS fragment reads in it
#endif
            if (!(IG3_DBG & 1) && h % SPT == 0 && tp + WD < 27) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[(tp + WD) % (WD + 1)][i] =
                        __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, voff[i], tap_off(tp + WD), 0));
            }
            if constexpr (DMA) {                           // one piece of the NEXT chunk's halo per tap (NPC <= 16 < 27 taps)
                if (h % SPT == 1 && tp < NPC && kc + 1 < nchunk) dma_piece(tp, kc + 1, (kc + 1) & 1);
            }
            if (!(IG3_DBG & 2) && h + 1 < NSTEP) lds_step(h + 1, bf[(h + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) M::mma(af[tp % (WD + 1)][i], bf[h & 1][jj], acc_k[i][(h % SPT) * 4 + jj]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (DMA) {                               // the next chunk's halo has landed; everybody is done with this one
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---------------- epilogue: buffer stores, 32-bit lane offset + scalar offset per point tile j
    f32x4 acc[MT][NT];                              // (uniform control flow here: Mma<float>::f32 exchanges values between lanes)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = M::f32(acc_k[i][j]);
    float ssum[MT][4], ssq[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
    const int out_bytes = O0 * O1 * O2 * A.Cy * (int)sizeof(T);     // < 2^31 (checked on the host)
    const int64_t out_base = ITEMS ? row_off * A.Cy * (int64_t)sizeof(T) : (int64_t)n * out_bytes;
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(A.y) + out_base, 0, out_bytes, 0x00020000);
    const auto rrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(A.res)) + out_base, 0, A.res ? out_bytes : 0, 0x00020000);
    const int lw = l0w + (li & 7), lh0 = l0h + (li >> 3), ld0 = l0d + wc * (NT / 4);
    const int orow = O2 * A.Cy * (int)sizeof(T), oslab = O1 * orow;
    const int rl = row0 + wr * MT * 16 + q * 4;                                  // first of this lane's 4 rows (i = 0)
    const int vb = ld0 * oslab + lh0 * orow + (lw * A.Cy + rl) * (int)sizeof(T);
    float bia[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) bia[i][r] = A.bias ? A.bias[rl + i * 16 + r] : 0.f;
    float csc[NB ? MT : 1][4], csh[NB ? MT : 1][4], nsa[NB ? MT : 1][4], nsb[NB ? MT : 1][4];
    const auto nyrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(NB ? A.ny : A.y)) + out_base, 0, NB ? out_bytes : 0, 0x00020000);
    if constexpr (NB) {
        if (tid < NBR) {
            const int ch = row0 + tid;
            const bool ok = ch < A.ncout;
            const float mu = A.nmr[((int64_t)n * A.Cy + ch) * 2], rs = A.nmr[((int64_t)n * A.Cy + ch) * 2 + 1];
            const float sc = ok ? rs * A.ngamma[ok ? ch : 0] : 0.f;
            nb_c[tid][0] = ok ? mu : 0.f; nb_c[tid][1] = ok ? rs : 0.f; nb_c[tid][2] = sc;
            nb_c[tid][3] = ok ? A.nbeta[ok ? ch : 0] - mu * sc : 0.f;           // the forward pass's expressions (k_norm_apply)
            nb_s[tid][0] = 0.0; nb_s[tid][1] = 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                csc[i][r] = nb_c[(wr * MT + i) * 16 + q * 4 + r][2]; csh[i][r] = nb_c[(wr * MT + i) * 16 + q * 4 + r][3];
                nsa[i][r] = 0.f; nsb[i][r] = 0.f;
            }
    }
    constexpr int YB = 4;                              // pre-norm pieces requested together (the stores in between would order them one by one)
    u32x4 ypre[NB ? YB : 1];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ld = ld0 + (j >> 2), lh = lh0 + 2 * (j & 3);
        const bool valid = (ld < O0) && (lh < O1) && (lw < O2);
        const int so = (j >> 2) * oslab + 2 * (j & 3) * orow;
        if constexpr (NB) {
            if (j % YB == 0) {
#pragma unroll
                for (int jj = 0; jj < YB && j + jj < NT; ++jj) {
                    const int ld_ = ld0 + ((j + jj) >> 2), lh_ = lh0 + 2 * ((j + jj) & 3);
                    const bool v_ = (ld_ < O0) && (lh_ < O1) && (lw < O2);
                    const int vb16_ = vb + (((q >> 1) * 8 + (q & 1) * 16) - q * 4) * (int)sizeof(T);
                    ypre[jj] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(nyrs, v_ ? vb16_ : (int)0x80000000,
                                                                                              ((j + jj) >> 2) * oslab + 2 * ((j + jj) & 3) * orow, 0));
                }
            }
        }
        if constexpr (sizeof(T) == 2 && MT == 2) {
            // bf16, two row tiles per wave: ONE 16-byte store per lane instead of two 8-byte ones (the epilogue is store-issue bound).
            // v_permlane16_swap exchanges the packed row-tile-0 values of the odd lane rows with the row-tile-1 values of the even
            // ones; lane row q then holds 8 consecutive channels: q = 0: 0-7, 1: 16-23, 2: 8-15, 3: 24-31 (tools/probe_permlane.hip).
            // The partner lane (q ^ 1, same li) belongs to the same point, so `valid` is the same for both.
            uint32_t pk[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v0 = acc[i][j][0] + bia[i][0], v1 = acc[i][j][1] + bia[i][1], v2 = acc[i][j][2] + bia[i][2], v3 = acc[i][j][3] + bia[i][3];
                if (A.res) {
                    float r4[4];
                    buf_load4<T>(rrs, valid ? vb + i * 16 * (int)sizeof(T) : (int)0x80000000, so, r4);
                    v0 += r4[0]; v1 += r4[1]; v2 += r4[2]; v3 += r4[3];
                }
                pk[i][0] = H16<T>::pack2(v0, v1); pk[i][1] = H16<T>::pack2(v2, v3);
                if (A.stats && valid) {
                    v0 = H16<T>::lo(pk[i][0]); v1 = H16<T>::hi(pk[i][0]);
                    v2 = H16<T>::lo(pk[i][1]); v3 = H16<T>::hi(pk[i][1]);
                    ssum[i][0] += v0; ssum[i][1] += v1; ssum[i][2] += v2; ssum[i][3] += v3;
                    ssq[i][0] += v0 * v0; ssq[i][1] += v1 * v1; ssq[i][2] += v2 * v2; ssq[i][3] += v3 * v3;
                }
            }
            if constexpr (NB) {        // S1 / S2 from the ROUNDED values (what k_norm_bwd_apply will read back) and the pre-norm tensor (k_dgs<.., NB>)
                const u32x4 y16 = ypre[j % YB];            // in the layout of the 16-byte stores: the same swap (an involution) gives the MFMA layout
                const v2u_t y0 = __builtin_amdgcn_permlane16_swap(y16[0], y16[2], false, false);
                const v2u_t y1 = __builtin_amdgcn_permlane16_swap(y16[1], y16[3], false, false);
                const uint32_t yk[2][2] = {{y0[0], y1[0]}, {y0[1], y1[1]}};
                if (valid) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float gv[4] = {H16<T>::lo(pk[h][0]), H16<T>::hi(pk[h][0]), H16<T>::lo(pk[h][1]), H16<T>::hi(pk[h][1])};
                        const float yv[4] = {H16<T>::lo(yk[h][0]), H16<T>::hi(yk[h][0]), H16<T>::lo(yk[h][1]), H16<T>::hi(yk[h][1])};
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            float gm = gv[rr];
                            if (A.nrelu && !(fmaf(yv[rr], csc[NB ? h : 0][rr], csh[NB ? h : 0][rr]) > 0.f)) gm = 0.f;   // the forward pass's expression
                            nsa[NB ? h : 0][rr] += gm;
                            nsb[NB ? h : 0][rr] = fmaf(gm, yv[rr], nsb[NB ? h : 0][rr]);      // RAW moment sum g * y: centred once per workgroup below
                        }
                    }
                }
            }
            const v2u_t s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
            const v2u_t s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
            const v4u_t st16 = {s0[0], s1[0], s0[1], s1[1]};
            const int vb16 = vb + (((q >> 1) * 8 + (q & 1) * 16) - q * 4) * (int)sizeof(T);
            __builtin_amdgcn_raw_buffer_store_b128(st16, yrs, valid ? vb16 : (int)0x80000000, so, 0);      // out-of-range offsets are dropped
        } else if (valid) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float v0 = acc[i][j][0] + bia[i][0], v1 = acc[i][j][1] + bia[i][1], v2 = acc[i][j][2] + bia[i][2], v3 = acc[i][j][3] + bia[i][3];
                if (A.res) {
                    float r4[4];
                    buf_load4<T>(rrs, vb + i * 16 * (int)sizeof(T), so, r4);
                    v0 += r4[0]; v1 += r4[1]; v2 += r4[2]; v3 += r4[3];
                }
                buf_store4r<T>(yrs, vb + i * 16 * (int)sizeof(T), so, v0, v1, v2, v3);
                if (A.stats) {
                    ssum[i][0] += v0; ssum[i][1] += v1; ssum[i][2] += v2; ssum[i][3] += v3;
                    ssq[i][0] += v0 * v0; ssq[i][1] += v1 * v1; ssq[i][2] += v2 * v2; ssq[i][3] += v3 * v3;
                }
            }
        }
    }
    if constexpr (NB) {
        // lanes li = 0..15 of a row q hold the same 8 channels: add over the 16 points, then the waves of a row block through LDS, then
        // ONE fp64 atomic per (channel, sum) and workgroup into k_norm_bwd_reduce's replica layout
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float a = nsa[i][rr], b = nsb[i][rr];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (li == 0) {
                    atomicAdd(&nb_s[(wr * MT + i) * 16 + q * 4 + rr][0], (double)a);
                    atomicAdd(&nb_s[(wr * MT + i) * 16 + q * 4 + rr][1], (double)b);
                }
            }
        __syncthreads();
        if (tid < NBR * 2) {
            const int rep = (blockIdx.x + blockIdx.z * 7) % NNDET_STATS_REPLICAS;
            double v = nb_s[tid >> 1][tid & 1];
            if (tid & 1) v = (double)nb_c[tid >> 1][1] * (v - (double)nb_c[tid >> 1][0] * nb_s[tid >> 1][0]);   // sum g*xhat = rstd * (sum g*y - mean * sum g)
            if (v != 0.0) atomicAdd(A.nred + (((int64_t)rep * A.N + n) * A.Cy + row0 + (tid >> 1)) * 2 + (tid & 1), v);
        }
    }
    if (A.stats) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);   // [WR*MT*16][2]
        if (tid < WR * MT * 16 * 2) red[tid] = 0.0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = dpp_row_sum(ssum[i][r]), s2 = dpp_row_sum(ssq[i][r]);
                if (li == 0) {
                    atomicAdd(&red[((wr * MT + i) * 16 + q * 4 + r) * 2 + 0], (double)s);
                    atomicAdd(&red[((wr * MT + i) * 16 + q * 4 + r) * 2 + 1], (double)s2);
                }
            }
        __syncthreads();
        if (tid < WR * MT * 16 * 2) {
            const int rep = (blockIdx.x + blockIdx.z * 7) % NNDET_STATS_REPLICAS;
            double* dst = A.stats + (((int64_t)rep * A.N + n) * A.Cy + row0 + (tid >> 1)) * 2 + (tid & 1);
            atomicAdd(dst, red[tid]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3x3, stride 1, 32 -> 32 channels
// k_ig3r: persistent kernel for the full-resolution 32 -> 32 layers (bf16), the layer shape with the most FLOPs of the network.
// k_ig3 tops out at ~0.9 PFLOP/s there: with only 2 row tiles to amortise a weight fragment over, the 2 weight buffer loads per 16
// MFMAs saturate the vector-memory pipe (tools/probe_loop.hip: 1.0 PF with them, 1.6 PF without). Here
//   * ALL weights live in registers: 27 taps x 2 row tiles x 4 VGPRs = 216 registers per lane (one wave per SIMD, 512-register
//     budget); a workgroup loads them once and then walks over ~75 tiles of 8x8x8 outputs;
//   * the halo of the NEXT tile goes global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds; no staging registers, no ds_write
//     pass; lanes whose voxel is outside the tensor use an out-of-range offset and the hardware writes the zero padding:
//     tools/probe_ldsdma.hip) into the second 64 KB buffer while the MFMAs of the current tile run; one barrier per tile;
//   * the MFMA order is point-group major (d-plane 0 of the wave: 27 taps, then d-plane 1), so the 8 accumulators of a finished
//     plane are free while the other plane computes: the epilogue (bias, bf16 pack, store, norm statistics) of one plane is cut
//     into single-instruction micro-ops and issued one per MFMA slot under the MFMAs of the next plane / next tile;
//   * the statistics stay in registers across the tiles of an image (one DPP reduction + 64 fp64 atomics per wave and image).
// Every MFMA slot is pinned with sched_barrier: with one wave per SIMD nothing else hides a badly placed instruction.
struct Ig3rArgs {
    const void* x; const void* w; const float* bias; void* y; double* stats;
    int32_t N, D, H, W;          // spatial dims of input == output, all multiples of 8
    int32_t nt1, nt2, ntiles;    // tiles along H, W; tiles per image
    uint32_t m_tiles, m_hw, m_w; // magic multipliers for / ntiles, / (nt1 * nt2), / nt2 (0 = divisor 1)
    int32_t rev;                 // backward-data: the tap at halo offset idx is weight tap 26 - idx
    int32_t total, per_xcd;      // N * ntiles; tiles per XCD (workgroup ids go round-robin over the 8 XCDs)
};

__device__ __forceinline__ uint32_t mdiv(uint32_t n, uint32_t m) { return m ? __umulhi(n, m) : n; }
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, int voff, uint32_t lds_dst) {
    // M0 is written in the same statement that reads it (the compiler does not preserve it around asm); hipcc does not count this
    // load: its completion is waited for by hand (vmcnt(0) before the tile barrier)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}

#ifndef IG3R_PD
#define IG3R_PD 2
#endif
#ifndef IG3R_DPH
#define IG3R_DPH 0      // phase (d-plane) whose MFMAs cover the LDS-DMA issue
#define IG3R_DS0 9      // first step (after the stores of the previous plane: measured 3 % faster than from step 1)
#define IG3R_PPS 1      // pieces per step
#define IG3R_SL0 6      // MFMA slot of the first piece of a step
#endif
#ifndef IG3R_SPLIT
#define IG3R_SPLIT 0
#endif
#ifndef IG3R_SKEW
#define IG3R_SKEW 0
#endif
#ifndef IG3R_DBG
#define IG3R_DBG 0      // timing experiments only (wrong results): 1 no LDS-DMA in the loop, 2 no epilogue, 4 no tile barrier
#endif
template <typename T, bool STATS>     // T: bf16_t / f16_t (same 16x16x32 MFMA shape and fragment layout, only the mnemonic differs)
__global__ __launch_bounds__(256, 1) void k_ig3r(const Ig3rArgs A) {
    constexpr int HH = 10, HW = 10, BUF = 65536, NPIECE = 16;
    constexpr int NM = STATS ? 39 : 15;                      // epilogue micro-ops per point tile (= the two row-tile accumulators of 16 points)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, stride = gridDim.x >> 3;
    int g = xcd * A.per_xcd + slot;
    const int g_end = min((xcd + 1) * A.per_xcd, A.total);
    if (g >= g_end) return;                                   // uniform
    const int rowb = A.W * 64, slab = A.H * rowb, img_bytes = A.D * slab;        // < 2^31 (checked on the host)

    // ---- staging: wave wv moves pieces wv * 16 .. wv * 16 + 15 (1 KB each) of the 10x10x10 halo; LDS granule G = piece * 64 + lane
    // holds halo row G / 40, voxel (G % 40) / 4, 16-byte part ((G % 4) ^ (row parity << 1)) (the XOR swizzle of k_ig3, applied to the
    // SOURCE address because the LDS-DMA destination is lane-linear)
    int rel[NPIECE];
    uint32_t bsel[NPIECE];
#pragma unroll
    for (int p = 0; p < NPIECE; ++p) {
        const int G = (wv * NPIECE + p) * 64 + lane;
        const int row = G / 40, col = G - row * 40;
        const int hd = row / HH, hh = row - hd * HH, hw = col >> 2, part = (col & 3) ^ ((row & 1) << 1);
        rel[p] = (hd * A.H + hh) * rowb + hw * 64 + part * 16;
        bsel[p] = G < 4000 ? (1u << hd) | (1u << (10 + hh)) | (1u << (20 + hw)) : 0x80000000u;
    }
    // scalars of a tile: image, byte offset of the halo origin in the input image, of the tile origin in the output image, validity
    // mask (bit hd | bit 10 + hh | bit 20 + hw set = that halo plane / row / column lies inside the tensor)
    auto decode = [&](int gg, int& n, int& tin, int& tout, uint32_t& M) {
        n = (int)mdiv((uint32_t)gg, A.m_tiles);
        const int t = gg - n * A.ntiles;
        const int td = (int)mdiv((uint32_t)t, A.m_hw);
        const int r = t - td * A.nt1 * A.nt2;
        const int th = (int)mdiv((uint32_t)r, A.m_w);
        const int tw = r - th * A.nt2;
        const int l0d = td * 8, l0h = th * 8, l0w = tw * 8;
        tin = ((l0d - 1) * A.H + (l0h - 1)) * rowb + (l0w - 1) * 64;
        tout = (l0d * A.H + l0h) * rowb + l0w * 64;
        const uint32_t dm = 0x3ffu & ~(l0d == 0 ? 1u : 0u) & ~(l0d + 8 == A.D ? 0x200u : 0u);
        const uint32_t hm = 0x3ffu & ~(l0h == 0 ? 1u : 0u) & ~(l0h + 8 == A.H ? 0x200u : 0u);
        const uint32_t wm = 0x3ffu & ~(l0w == 0 ? 1u : 0u) & ~(l0w + 8 == A.W ? 0x200u : 0u);
        M = dm | (hm << 10) | (wm << 20);
    };
    auto x_rsrc = [&](int n) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A.x)) + (int64_t)n * img_bytes, 0, img_bytes, 0x00020000);
    };
    auto y_rsrc = [&](int n, bool live) {
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(A.y) + (int64_t)n * img_bytes, 0, live ? img_bytes : 0, 0x00020000);
    };

    // ---- weights -> registers
    u32x4 wr[27][2];
    {
        const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A.w), 0, 27 * 2048, 0x00020000);
#pragma unroll
        for (int tp = 0; tp < 27; ++tp)
#pragma unroll
            for (int i = 0; i < 2; ++i)
            {   // step tp visits halo offset (a, b, c) = (tp / 9, tp % 3, (tp / 3) % 3): the tap order of k_ig3, so both kernels add
                // the 27 x 32 products of an output in the same order and give bit-identical results
                const int widx = ((tp / 9) * 3 + tp % 3) * 3 + (tp / 3) % 3;
                wr[tp][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, ((i * 16 + li) * 32 + q * 8) * 2, (A.rev ? 26 - widx : widx) * 2048, 0));
            }
    }
    float bia[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) bia[i][r] = A.bias ? A.bias[i * 16 + q * 4 + r] : 0.f;

    // ---- first tile: stage into buffer 0
    int n_cur, tin_cur, tout_cur; uint32_t M_cur;
    decode(g, n_cur, tin_cur, tout_cur, M_cur);
    {
        const auto xrs = x_rsrc(n_cur);
#pragma unroll
        for (int p = 0; p < NPIECE; ++p) {
            const int vo = (M_cur & bsel[p]) == bsel[p] ? tin_cur + rel[p] : (int)0x80000000;
            lds_dma16(xrs, vo, (uint32_t)((wv * NPIECE + p) * 1024));
        }
    }
    // fragment addresses: the two swizzle phases of this lane's base (see k_ig3); ^= BUF switches the halo buffer
    const int par = (li >> 3) & 1;
    const int lanevox = (wv * 2 * HH + (li >> 3)) * HW + (li & 7);
    int sb0 = lanevox * 64 + ((q ^ (par << 1)) << 4);
    int sb1 = lanevox * 64 + ((q ^ (par << 1) ^ 2) << 4);
    auto lds_frag = [&](int gph, int tp, int jj) {
        const int a = tp / 9, b = tp % 3, c = (tp / 3) % 3;
        return *reinterpret_cast<const u32x4*>(smem + ((b & 1) ? sb1 : sb0) + (((gph + a) * HH + 2 * jj + b) * HW + c) * 64);
    };
    // One 16-byte store per lane and point tile instead of two 8-byte ones (the epilogue is store-ISSUE bound): v_permlane16_swap
    // exchanges the packed row-tile-0 values of the odd lane rows with the row-tile-1 values of the even ones, after which lane row q
    // holds 8 CONSECUTIVE channels of its point: q = 0: 0-7, 1: 16-23, 2: 8-15, 3: 24-31 (tools/probe_permlane.hip)
    const int vlane = ((li >> 3) * A.W + (li & 7)) * 64 + ((q >> 1) * 8 + (q & 1) * 16) * 2;

    f32x4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ssum[2][4], ssq[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
    auto flush = [&](int n) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = dpp_row_sum(ssum[i][r]), s2 = dpp_row_sum(ssq[i][r]);
                if (li == 0) {
                    const int rep = (blockIdx.x * 4 + wv) % NNDET_STATS_REPLICAS;
                    double* dst = A.stats + (((int64_t)rep * A.N + n) * 32 + i * 16 + q * 4 + r) * 2;
                    atomicAdd(dst, (double)s);
                    atomicAdd(dst + 1, (double)s2);
                }
                ssum[i][r] = 0.f; ssq[i][r] = 0.f;
            }
    };
    // epilogue micro-op m of fragment (i, j): the values pass through ev / epk between slots
    float ev[2][4];
    uint32_t epk[2][2];
    v4u_t est;
    auto epi_on = [&](float (&ev)[2][4], uint32_t (&epk)[2][2], v4u_t& est, int j, int m, __amdgpu_buffer_rsrc_t yrs, int soff) {
        if (m < 8) ev[m >> 2][m & 3] = acc[m >> 2][j][m & 3] + bia[m >> 2][m & 3];
        else if (m < 12) { const int i = (m - 8) >> 1, h = (m - 8) & 1; epk[i][h] = H16<T>::pack2(ev[i][2 * h], ev[i][2 * h + 1]); }
        else if (m < 14) {
            const int h = m - 12;
            const v2u_t r = __builtin_amdgcn_permlane16_swap(epk[0][h], epk[1][h], false, false);
            est[h] = r[0]; est[2 + h] = r[1];
        }
        else if (m == 14) __builtin_amdgcn_raw_buffer_store_b128(est, yrs, vlane, soff, 0);
        else if (m < 23) { const int i = (m - 15) >> 2, r = (m - 15) & 3; ev[i][r] = (r & 1) ? H16<T>::hi(epk[i][r >> 1]) : H16<T>::lo(epk[i][r >> 1]); }
        else if (m < 31) { const int i = (m - 23) >> 2, r = (m - 23) & 3; ssum[i][r] += ev[i][r]; }
        else { const int i = (m - 31) >> 2, r = (m - 31) & 3; ssq[i][r] = fmaf(ev[i][r], ev[i][r], ssq[i][r]); }
    };
    auto epi = [&](int j, int m, __amdgpu_buffer_rsrc_t yrs, int soff) { epi_on(ev, epk, est, j, m, yrs, soff); };

    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // activation fragments: read PD steps (of 8 MFMAs) ahead into a ring of PD + 1 register sets (54 steps per tile: 54 % (PD + 1) == 0)
    constexpr int PD = IG3R_PD, RING = PD + 1;
    static_assert(54 % RING == 0, "ring phase must repeat per tile");
    u32x4 bf[RING][4];
#pragma unroll
    for (int hh = 0; hh < PD; ++hh)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bf[hh][jj] = lds_frag(0, hh, jj);

    bool first = true;
    int cur_buf = 0;                                                    // LDS byte offset of the halo buffer being read (scalar twin of sb0 & BUF)
    int stat_n = n_cur;
    int n_prev = n_cur, tout_prev = 0;
    for (;;) {
        const int g_next = g + stride;
        const bool has_next = g_next < g_end;
        int n_next, tin_next, tout_next; uint32_t M_next;
        decode(has_next ? g_next : g, n_next, tin_next, tout_next, M_next);
        if (!has_next) M_next = 0;                                      // every piece out of range: the DMA writes zeros nobody reads
        const auto xrs = x_rsrc(n_next);
        const uint32_t dma_dst = (uint32_t)((cur_buf ^ BUF) + wv * NPIECE * 1024);
        int dma_vo = 0;
#pragma unroll
        for (int gph = 0; gph < 2; ++gph) {
            // the plane whose epilogue runs under this phase: plane 1 of the previous tile (phase 0) / plane 0 of this tile (phase 1)
            const auto yrs = gph == 0 ? y_rsrc(n_prev, !first) : y_rsrc(n_cur, true);
            const int ebase = (gph == 0 ? tout_prev : tout_cur) + (wv * 2 + (gph ^ 1)) * slab;
            if (gph == 1 && STATS) {
                if (first) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
                } else if (stat_n != n_cur) flush(stat_n);
                stat_n = n_cur;
            }
#pragma unroll
            for (int tp = 0; tp < 27; ++tp) {
                const int h = gph * 27 + tp;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int i = s >> 2, jj = s & 3;
                    // weights as AGPR operands (asm: the builtin form keeps them in VGPRs, i.e. 262 v_accvgpr_read + 104 s_nop per tile
                    // for the 60+ registers that do not fit); first tap: C = inline 0. Hazards: B comes from ds_read (counted wait by
                    // the compiler), an accumulator is next read >= 8 MFMA slots after its last MFMA
                    if constexpr (sizeof(T) == 2 && !__is_same(T, bf16_t)) {
                        if (tp == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(acc[i][gph * 4 + jj]) : "a"(wr[tp][i]), "v"(bf[h % RING][jj]));
                        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][gph * 4 + jj]) : "a"(wr[tp][i]), "v"(bf[h % RING][jj]));
                    } else {
                        if (tp == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc[i][gph * 4 + jj]) : "a"(wr[tp][i]), "v"(bf[h % RING][jj]));
                        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][gph * 4 + jj]) : "a"(wr[tp][i]), "v"(bf[h % RING][jj]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // fillers of this MFMA slot
                    if (s < 4) {
                        const int hn = h + PD;                          // the step whose fragments are read now
                        if (hn >= 54) {                                 // = step hn - 54 of the next tile, from the other halo buffer
                            if (hn == 54 && s == 0) {
                                if (!(IG3R_DBG & 4)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                                sb0 ^= BUF; sb1 ^= BUF; cur_buf ^= BUF;
#if IG3R_SKEW
                                // the barrier releases the four waves in the same cycle and their LDS-DMA pieces would collide in the
                                // address path step after step: delay wave w by w * IG3R_SKEW * 8 cycles once per tile
                                for (int k = 0; k < wv * IG3R_SKEW; ++k) asm volatile("s_nop 7");
#endif
                            }
                            bf[hn % RING][s] = lds_frag(0, hn - 54, s);
                        } else {
                            bf[hn % RING][s] = lds_frag(hn / 27, hn % 27, s);
                        }
                    }
                    const int e = tp * 8 + s - 8;                       // epilogue micro-op of this slot
                    if (e >= 0 && e < 4 * NM && !(IG3R_DBG & 2)) {
                        const int f = e / NM, m = e % NM;
                        epi((gph ^ 1) * 4 + f, m, yrs, ebase + 2 * f * rowb);
                    }
                    // LDS-DMA of the next tile: IG3R_PPS pieces per step from step IG3R_DS0 of phase IG3R_DPH on; the offset is computed one
                    // slot before the issue slot
#if IG3R_SPLIT
                    if (tp >= IG3R_DS0 && tp < IG3R_DS0 + NPIECE / 2 && !(IG3R_DBG & 1)) {      // 8 pieces under each plane
                        const int p = gph * (NPIECE / 2) + tp - IG3R_DS0;
                        if (s == IG3R_SL0 - 1) dma_vo = (M_next & bsel[p]) == bsel[p] ? tin_next + rel[p] : (int)0x80000000;
                        if (s == IG3R_SL0) lds_dma16(xrs, dma_vo, dma_dst + p * 1024);
                    }
#else
                    if (gph == IG3R_DPH && tp >= IG3R_DS0 && tp < IG3R_DS0 + NPIECE / IG3R_PPS && !(IG3R_DBG & 1)) {
#pragma unroll
                        for (int k = 0; k < IG3R_PPS; ++k) {
                            const int p = (tp - IG3R_DS0) * IG3R_PPS + k, sl = IG3R_SL0 + k * (IG3R_PPS > 2 ? 1 : 2);
                            if (s == sl - 1) dma_vo = (M_next & bsel[p]) == bsel[p] ? tin_next + rel[p] : (int)0x80000000;
                            if (s == sl) lds_dma16(xrs, dma_vo, dma_dst + p * 1024);
                        }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        first = false;
        n_prev = n_cur; tout_prev = tout_cur;
        n_cur = n_next; tout_cur = tout_next;
        if (!has_next) break;
        g = g_next;
    }
    // plane 1 of the last tile. Its accumulators were written by the inline-asm MFMAs just above: the compiler does not know that
    // these are MFMAs and inserts none of the wait states a VALU read of a matrix-core result needs (inside the loop every
    // accumulator is read >= 8 MFMA slots after its last MFMA); without them the last tile of every workgroup came out with a few
    // stale values at full size (tools/diag_ig3r.py). Wait out the matrix pipe before the first read.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // Second hazard, the one tools/diag_ig3r.py actually showed (lanes 12-15 of the second data dword of the LAST tile of every
    // workgroup overwritten, at full size only): a 16-byte buffer store followed at once by a VALU write of its data registers.
    // hipcc guards that write-after-read hazard only for stores WITHOUT an SGPR offset (GCNHazardRecognizer::createsVALUHazard);
    // with one it assumes the hardware is safe, and under load it is not. Inside the loop an MFMA slot separates the store from the
    // next micro-op; here every point tile gets its OWN registers, the four stores are issued back to back after all the arithmetic
    // and nothing overwrites their data before the wave ends (tools/scan_store_hazard.py checks the emitted ISA for the pattern).
    {
        const auto yrs = y_rsrc(n_prev, true);
        const int ebase = tout_prev + (wv * 2 + 1) * slab;
        float ev4[4][2][4];
        uint32_t epk4[4][2][2];
        v4u_t est4[4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int m = 0; m < 14; ++m) epi_on(ev4[f], epk4[f], est4[f], 4 + f, m, yrs, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 4; ++f) epi_on(ev4[f], epk4[f], est4[f], 4 + f, 14, yrs, ebase + 2 * f * rowb);
        asm volatile("s_nop 3" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (STATS) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int m = 15; m < NM; ++m) epi_on(ev4[f], epk4[f], est4[f], 4 + f, m, yrs, 0);
            flush(stat_n);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct Plan {
    IgArgs a;
    int cfg;          // index into CFG_*
    dim3 grid;
    size_t lds;
    int ksplit;       // > 1: split-K launch (k_igemm + k_ig_splitk_reduce), needs splitk_bytes of workspace
    size_t splitk_bytes;
};

// cfg: 0 unit stride, rows % 64 != 0 : 32 rows x 512 points (WR = 1: the 32-row layers are not weight-traffic bound) ;
//      1 unit stride : 64 rows x 256 points (WR = 2) ; 2 strided : 64 rows x 128 points (WR = 2) ;
//      3 strided, rows % 64 != 0 : 32 rows x 128 points (WR = 2)
// A persistent, register-prefetched variant of this kernel (one workgroup per CU, commit 'igemm: persistent
// software-pipelined kernel') measured 25-50 % SLOWER than two independent workgroups per CU: with one wave per SIMD every
// L2 / LDS stall of the tap loop is exposed. Two co-resident workgroups hide each other's staging and stalls.
//      4 (experimental, NNDET_IGEMM_A256=1): like 0 but 256 points per workgroup, 4 workgroups per CU
//      5 / 6 / 7: k_ig3 (3x3x3 stride 1, compile-time tile) with 32 rows x 512 points / 64 rows x 256 points / 64 rows x 512
//      points. The main loop of 7 (NT = 16 point tiles per wave, one weight fragment pair per 32 MFMAs) sustains 1600 TF/s in
//      tools/probe_loop.hip where the NT = 8 loops stop at ~1000 (the weight buffer loads saturate the vector memory pipe).
//      8 / 9 / 10: strided, 64 points per workgroup (halo 9^3 x 64 B = 46 KB at stride 2 -> 3 workgroups per CU instead of 1 at
//      the 88 KB of the 128-point tile): 64 rows as 2 x 2 waves (8), 64 rows as 4 row waves x 64 points (9), 32 rows (10)
static const int CFG_ROWS[12] = {32, 64, 64, 32, 32, 32, 64, 64, 64, 64, 32, 128};
static const int CFG_PTS[12] = {512, 256, 128, 128, 256, 512, 256, 512, 64, 64, 64, 512};

static bool choose_tile(const int Lmax[3], const int in_step[3], const int span[3], int points, int maxp, int T[3], int H[3]) {
    double best = 1e300;
    bool found = false;
    for (int td = 1; td <= points; td *= 2)
        for (int th = 1; td * th <= points; th *= 2) {
            const int tw = points / (td * th);
            if (td * th * tw != points) continue;
            const int t[3] = {td, th, tw};
            int h[3];
            int64_t hv = 1;
            double padded = 1.0;
            for (int a = 0; a < 3; ++a) {
                h[a] = (t[a] - 1) * in_step[a] + span[a];
                hv *= h[a];
                padded *= (double)ceil_div(Lmax[a], t[a]) * t[a];
            }
            if (hv * 4 > 256 * maxp) continue;
            const double cost = padded * (1.0 + 0.15 * (double)hv / points);   // padding waste first, halo amplification second
            if (cost < best) { best = cost; found = true; for (int a = 0; a < 3; ++a) { T[a] = t[a]; H[a] = h[a]; } }
        }
    return found;
}

// kind: 0 forward, 1 backward-data
static int build_plan(const NndetConv* c, int kind, Plan* P, bool force_spec = false) {
    IgArgs& a = P->a;
    memset(&a, 0, sizeof(a));
    const int KCb = nndet_is16(c->dtype) ? 32 : 16;
    const bool tr = c->transposed != 0;
    if (tr) for (int i = 0; i < 3; ++i) if (c->k[i] != c->s[i] || c->p[i] != 0) return NNDET_EINVAL;
    // which tensor is read / written
    const bool reads_in = (kind == 0);   // forward reads the conv input, backward-data reads dY
    const int in_sp[3] = {c->in_d, c->in_h, c->in_w}, out_sp[3] = {c->out_d, c->out_h, c->out_w};
    const int* xs = reads_in ? in_sp : out_sp;
    const int* ys = reads_in ? out_sp : in_sp;
    a.N = c->batch;
    a.Cx = reads_in ? c->cin_p : c->cout_p;
    a.Cy = reads_in ? c->cout_p : c->cin_p;
    if (a.Cx % KCb != 0 || a.Cy % 32 != 0) return NNDET_EINVAL;
    for (int i = 0; i < 3; ++i) { a.I[i] = xs[i]; a.O[i] = ys[i]; }
    // "gather" form (one class, strided input walk) vs "class" form (parity classes, unit input step)
    const bool gather = (kind == 0) != tr;   // conv fwd, convT bwd-data
    int span[3] = {1, 1, 1}, Lmax[3];
    int ntaps = 0;
    if (gather) {
        a.ncls = 1;
        IgClass& C = a.cls[0];
        for (int i = 0; i < 3; ++i) {
            a.in_step[i] = c->s[i]; a.out_step[i] = 1; C.out_off[i] = 0; C.L[i] = ys[i];
            C.in_base[i] = -c->p[i]; span[i] = c->k[i]; Lmax[i] = ys[i];
            // consistency of the declared shapes
            const int expect = tr ? in_sp[i] * c->s[i] : (in_sp[i] + 2 * c->p[i] - c->k[i]) / c->s[i] + 1;
            if (out_sp[i] != expect) return NNDET_EINVAL;
        }
        C.tap0 = 0;
        for (int td = 0; td < c->k[0]; ++td) for (int th = 0; th < c->k[1]; ++th) for (int tw = 0; tw < c->k[2]; ++tw) {
            if (ntaps >= 27) return NNDET_EINVAL;
            IgTap& t = a.taps[ntaps++];
            t.d[0] = td; t.d[1] = th; t.d[2] = tw;
            t.wt = (td * c->k[1] + th) * c->k[2] + tw;
        }
        C.ntap = ntaps;
    } else {
        const int S = c->s[0] * c->s[1] * c->s[2];
        if (S > 8) return NNDET_EINVAL;
        a.ncls = S;
        for (int i = 0; i < 3; ++i) { a.in_step[i] = 1; a.out_step[i] = c->s[i]; Lmax[i] = 0; }
        int ci = 0;
        for (int cd = 0; cd < c->s[0]; ++cd) for (int ch = 0; ch < c->s[1]; ++ch) for (int cw = 0; cw < c->s[2]; ++cw, ++ci) {
            IgClass& C = a.cls[ci];
            const int cc[3] = {cd, ch, cw};
            int lo[3], hi[3], cnt[3];
            int tl[3][3], dl[3][3];   // valid taps / deltas per axis
            for (int i = 0; i < 3; ++i) {
                C.out_off[i] = cc[i];
                C.L[i] = ys[i] > cc[i] ? (ys[i] - cc[i] + c->s[i] - 1) / c->s[i] : 0;
                if (C.L[i] > Lmax[i]) Lmax[i] = C.L[i];
                cnt[i] = 0; lo[i] = 1 << 30; hi[i] = -(1 << 30);
                for (int t = 0; t < c->k[i]; ++t) {
                    const int num = cc[i] + c->p[i] - t;
                    if (((num % c->s[i]) + c->s[i]) % c->s[i] != 0) continue;
                    const int d = num >= 0 ? num / c->s[i] : -((-num) / c->s[i]);
                    if (cnt[i] >= 3) return NNDET_EINVAL;
                    tl[i][cnt[i]] = t; dl[i][cnt[i]] = d; ++cnt[i];
                    if (d < lo[i]) lo[i] = d;
                    if (d > hi[i]) hi[i] = d;
                }
                if (cnt[i] == 0) { lo[i] = 0; hi[i] = 0; }
                C.in_base[i] = lo[i];
                if (hi[i] - lo[i] + 1 > span[i]) span[i] = hi[i] - lo[i] + 1;
            }
            C.tap0 = ntaps;
            if (cnt[0] && cnt[1] && cnt[2])
                for (int x = 0; x < cnt[0]; ++x) for (int y = 0; y < cnt[1]; ++y) for (int z = 0; z < cnt[2]; ++z) {
                    if (ntaps >= 27) return NNDET_EINVAL;
                    IgTap& t = a.taps[ntaps++];
                    t.d[0] = dl[0][x] - lo[0]; t.d[1] = dl[1][y] - lo[1]; t.d[2] = dl[2][z] - lo[2];
                    t.wt = (tl[0][x] * c->k[1] + tl[1][y]) * c->k[2] + tl[2][z];
                }
            C.ntap = ntaps - C.tap0;
        }
        for (int i = 0; i < 3; ++i) {
            const int expect = tr ? in_sp[i] * c->s[i] : (in_sp[i] + 2 * c->p[i] - c->k[i]) / c->s[i] + 1;
            if (out_sp[i] != expect) return NNDET_EINVAL;
        }
    }
    const bool strided = a.in_step[0] > 1 || a.in_step[1] > 1 || a.in_step[2] > 1;
    const bool r64 = (a.Cy % 64) == 0;
    static const int a256 = getenv("NNDET_IGEMM_A256") ? atoi(getenv("NNDET_IGEMM_A256")) : 0;
    const char* sv_env = getenv("NNDET_IGEMM_STRIDED");            // 0: 128-point tiles, 1: 64 points 2x2 waves, 2: 64 points 4 row waves
    const int sv = sv_env ? atoi(sv_env) : 2;                      // measured: profiles/round2_micro_strided_tiles.txt
    P->cfg = strided ? (r64 ? (sv == 1 ? 8 : (sv == 2 ? 9 : 2)) : (sv ? 10 : 3)) : (r64 ? 1 : (a256 ? 4 : 0));
    for (int i = 0; i < 3; ++i) if (Lmax[i] <= 0) return NNDET_EINVAL;
    // Small pyramid levels: with 256 / 512-point tiles a level of 150 ... 4800 positions gives 8 ... 200 workgroups for 256 CUs and
    // every one of them walks serially through 27 taps x all K chunks of a mostly padded tile. Below `small_wg` workgroups the
    // 64-point tiles are used instead (4 - 8x as many workgroups, each with 1/4 - 1/8 of the serial work): latency, not throughput.
    const char* swg_env = getenv("NNDET_IGEMM_SMALLWG");          // read per call so tests can flip it
    const int small_wg = swg_env ? atoi(swg_env) : 200;   // measured: 40 -> 160 workgroups halves the kernel, 200 -> 800 gains nothing
    bool small = false;
    if (!strided && small_wg > 0 && !force_spec) {
        const int64_t lat = (int64_t)Lmax[0] * Lmax[1] * Lmax[2];
        const int64_t wgs = ceil_div64(lat, CFG_PTS[P->cfg]) * (a.Cy / CFG_ROWS[P->cfg]) * a.N * a.ncls;
        if (wgs < small_wg) { small = true; P->cfg = r64 ? 9 : 10; }
    }
    const int points = CFG_PTS[P->cfg];
    if (!choose_tile(Lmax, a.in_step, span, points, strided ? (P->cfg >= 8 ? 16 : 24) : 16, a.T, a.H)) return NNDET_EINVAL;
    for (int i = 0; i < 3; ++i) a.nt[i] = ceil_div(Lmax[i], a.T[i]);
    auto magic = [](int d) -> uint32_t { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)d + 1ull); };
    auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    a.mHW = magic(a.H[2]); a.mHH = magic(a.H[1]);
    a.lT1 = ilog2(a.T[1]); a.lT2 = ilog2(a.T[2]);
    // LDS layout. Unit-stride tiles: XOR bit 5 with the row parity. Strided tiles (PIPE kernels): when the stride along H is 2 the
    // rows of a fragment are 2 apart -> swizzle on bit 1 of the row index (valid because the tile origin row is even: in_step[0] * H1
    // * pd + 2 * ph); when the stride along W is 2 the rows are stored de-interleaved. PMC before: 71 % of the LDS cycles of the
    // stride-2 forward kernel were bank conflicts (profiles/round2_pmc_e1_32to64_s2.txt).
    a.swz = strided ? 0 : 1;
    a.swzsh = 0; a.deint = 0; a.HWE = 0;
    static const int lds_v2 = getenv("NNDET_IGEMM_LDSV2") ? atoi(getenv("NNDET_IGEMM_LDSV2")) : 1;
    if (strided && lds_v2) {
        if (a.in_step[1] == 2 && ((a.in_step[0] * a.H[1]) % 2 == 0 || a.in_step[0] % 2 == 0)) { a.swz = 1; a.swzsh = 1; }
        if (a.in_step[2] == 2) { a.deint = 1; a.HWE = (a.H[2] + 1) / 2; }
    }
    {
        const int esz = nndet_esize(c->dtype);
        const int64_t tapb = (int64_t)a.Cy * a.Cx * esz;
        const int64_t wb = tapb * (c->k[0] * c->k[1] * c->k[2]);
        if (wb >= (1LL << 31)) return NNDET_EINVAL;
        a.wbytes = (int32_t)wb;
        for (int t = 0; t < ntaps; ++t) {
            const int trow = a.taps[t].d[0] * a.H[1] + a.taps[t].d[1];
            const int tw = a.taps[t].d[2];
            a.tapx[t].toff = (trow * a.H[2] + (a.deint ? (tw & 1) * a.HWE + (tw >> 1) : tw)) * 64;
            a.tapx[t].flip = ((trow >> a.swzsh) & a.swz) << 5;
            a.tapx[t].woff = (int32_t)(a.taps[t].wt * tapb);
        }
    }
    // 3x3x3 / stride 1 / pad 1 (forward or backward-data): compile-time (TD, 8, 8) tile kernel k_ig3, unless the fixed tile
    // pads the volume noticeably more than the tile choose_tile() found
    // NNDET_IGEMM_SPEC: 0 = never, 1 (default) = by the padding rule, 2 = always (tests); read per call so tests can flip it
    const char* spec_env = getenv("NNDET_IGEMM_SPEC");
    const int spec_on = force_spec ? 2 : (spec_env ? atoi(spec_env) : 1);
    if (spec_on && !small && !strided && !tr && a.ncls == 1 && ntaps == 27 && c->k[0] == 3 && c->k[1] == 3 && c->k[2] == 3 &&
        c->p[0] == 1 && c->p[1] == 1 && c->p[2] == 1 && c->s[0] == 1 && c->s[1] == 1 && c->s[2] == 1) {
        // 64-row layers: (8,8,8) tiles with NT = 16 (2 workgroups per CU, ~1.6x the main-loop rate) or (4,8,8) tiles with NT = 8
        // (3 per CU): compare padded work / (rate x occupancy of the last round of workgroups)
        int st[3] = {8, 8, 8};
        int spec_cfg = r64 ? 7 : 5;
        if (r64) {
            auto cost = [&](int td, double rate, int slots) {
                const double tiles = (double)ceil_div(Lmax[0], td) * ceil_div(Lmax[1], 8) * ceil_div(Lmax[2], 8);
                const double wgs = tiles * (a.Cy / 64) * a.N;
                const double rounds = (double)ceil_div64((int64_t)wgs, slots);
                return rounds * slots * td / rate;     // ~ time: rounds x per-workgroup work / rate
            };
            const char* nt_env = getenv("NNDET_IGEMM_NT");            // 8 / 16: force one variant (tests, experiments)
            const int force = nt_env ? atoi(nt_env) : 0;
            if (force == 8 || (force != 16 && cost(4, 1.0, 768) < cost(8, 1.6, 512))) { st[0] = 4; spec_cfg = 6; }
            // experimental (NNDET_IGEMM_R128=1): 128 rows x 512 points per workgroup, ONE workgroup per CU with 512 registers per wave
            const char* r128 = getenv("NNDET_IGEMM_R128");
            if (r128 && atoi(r128) == 1 && a.Cy % 128 == 0 && spec_cfg == 7 && nndet_is16(c->dtype) && !c->in_affine) spec_cfg = 11;
        }
        double pg = 1.0, ps = 1.0;
        for (int i = 0; i < 3; ++i) { pg *= (double)a.nt[i] * a.T[i]; ps *= (double)ceil_div(Lmax[i], st[i]) * st[i]; }
        const int64_t img_b = (int64_t)a.I[0] * a.I[1] * a.I[2] * a.Cx * nndet_esize(c->dtype);
        const int64_t w_b = (int64_t)27 * a.Cy * a.Cx * nndet_esize(c->dtype);
        const int64_t out_b = (int64_t)a.O[0] * a.O[1] * a.O[2] * a.Cy * nndet_esize(c->dtype);
        if ((ps <= 1.05 * pg || spec_on == 2) && img_b < (1LL << 31) && w_b < (1LL << 31) && out_b < (1LL << 31)) {   // 32-bit buffer offsets
            P->cfg = spec_cfg;
            for (int i = 0; i < 3; ++i) { a.T[i] = st[i]; a.H[i] = st[i] + 2; a.nt[i] = ceil_div(Lmax[i], st[i]); }
        }
    }
    const int ntiles = a.nt[0] * a.nt[1] * a.nt[2];
    P->grid = dim3(a.ncls > 1 ? ceil_div(ntiles, 8) * 8 * a.ncls : ntiles, a.Cy / CFG_ROWS[P->cfg], a.N);
    P->lds = (size_t)a.H[0] * a.H[1] * a.H[2] * 64;
    if (P->lds < 1024) P->lds = 1024;   // room for the stats reduction
    // Split-K for the deep stages / small pyramid levels: 100 ... 400 workgroups, each walking serially through up to 10 channel
    // chunks x 27 taps with a global -> LDS round trip per chunk, leave most of the 256 CUs idle or with a single resident wave per
    // SIMD (50 - 300 TFLOP/s, profiles/round3_v4_kernel_stats_by_grid.txt). Splitting the chunks over `ksplit` workgroups multiplies
    // the resident waves and divides the serial chain. NNDET_IGEMM_SPLITK=0 disables it, =N forces N (tests).
    P->ksplit = 1; P->splitk_bytes = 0;
    const bool generic = P->cfg < 5 || (P->cfg >= 8 && P->cfg <= 10);              // k_igemm configurations (k_ig3 / k_ig3r have no split-K epilogue)
    const int nchunk = a.Cx / KCb;
    const int64_t wgs = (int64_t)P->grid.x * P->grid.y * P->grid.z;
    const char* sk_env = getenv("NNDET_IGEMM_SPLITK");
    const int sk = sk_env ? atoi(sk_env) : -1;
    if (generic && sk != 0 && nchunk >= 2) {
        int ks = 1;
        if (sk > 1) ks = sk < nchunk ? sk : nchunk;
        // (automatic only in 16 bits: the fp32 kernels are the 1e-4 parity path, whose accumulation order the reference goldens pin)
        else if (nndet_is16(c->dtype) && wgs <= 320 && nchunk >= 4) {       // (400 workgroups x 8 chunks measured 18 % SLOWER split in two: profiles/round3_micro_splitk.txt)
            ks = (int)ceil_div64(1024, wgs);
            if (ks > nchunk / 2) ks = nchunk / 2;
        }
        if (ks >= 2) {
            P->ksplit = ks;
            P->splitk_bytes = (size_t)ks * a.N * a.O[0] * a.O[1] * a.O[2] * a.Cy * sizeof(float);
        }
    }
    return 0;
}

static const IgItems g_no_items = {};

template <typename T, bool AFF>
static int launch_cfg(const Plan& P, hipStream_t st) {
    switch (P.cfg) {
        case 0: k_igemm<T, 1, 2, 8, 16, 2, false, AFF><<<P.grid, 256, P.lds, st>>>(P.a); break;
        case 1: k_igemm<T, 2, 2, 8, 16, 3, false, AFF><<<P.grid, 256, P.lds, st>>>(P.a); break;
        case 2: k_igemm<T, 2, 2, 4, 24, 3, true, AFF><<<P.grid, 256, P.lds, st>>>(P.a); break;
        case 3: k_igemm<T, 2, 1, 4, 24, 4, true, AFF><<<P.grid, 256, P.lds, st>>>(P.a); break;
        case 8: k_igemm<T, 2, 2, 2, 16, 3, true, AFF><<<P.grid, 256, P.lds, st>>>(P.a); break;
        case 9: k_igemm<T, 4, 1, 4, 16, 3, true, AFF><<<P.grid, 256, P.lds, st>>>(P.a); break;
        case 10: k_igemm<T, 2, 1, 2, 16, 4, true, AFF><<<P.grid, 256, P.lds, st>>>(P.a); break;
        case 5: k_ig3<T, 1, 2, 8, 2, AFF><<<P.grid, 256, P.lds, st>>>(P.a, g_no_items); break;
        case 6: k_ig3<T, 2, 2, 8, 3, AFF><<<P.grid, 256, P.lds, st>>>(P.a, g_no_items); break;
        case 7: k_ig3<T, 2, 2, 16, 2, AFF><<<P.grid, 256, P.lds, st>>>(P.a, g_no_items); break;
        case 11: if constexpr (!AFF && sizeof(T) == 2) { k_ig3<T, 2, 4, 16, 1, false, false, true><<<P.grid, 256, 2 * 65536, st>>>(P.a, g_no_items); break; } else return NNDET_EINVAL;
        default: k_igemm<T, 1, 2, 4, 16, 4, false, AFF><<<P.grid, 256, P.lds, st>>>(P.a); break;
    }
    LAUNCH_CHECK();
    return 0;
}

// data gradient + norm-backward sums (IgArgs::ny ...): the k_ig3 configurations with two row tiles per wave, 16-bit types
template <typename T>
static int launch_cfg_nb(const Plan& P, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        switch (P.cfg) {
            case 5: k_ig3<T, 1, 2, 8, 2, false, false, false, true><<<P.grid, 256, P.lds, st>>>(P.a, g_no_items); break;
            case 6: k_ig3<T, 2, 2, 8, 3, false, false, false, true><<<P.grid, 256, P.lds, st>>>(P.a, g_no_items); break;
            case 7: k_ig3<T, 2, 2, 16, 2, false, false, false, true><<<P.grid, 256, P.lds, st>>>(P.a, g_no_items); break;
            default: return NNDET_EINVAL;
        }
        LAUNCH_CHECK();
        return 0;
    } else {
        return NNDET_EINVAL;
    }
}

template <typename T, int WR, int MT, int NT, int MAXP, int MINW, bool PIPE = false>
static int set_lds_attr() {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_igemm<T, WR, MT, NT, MAXP, MINW, PIPE, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_igemm<T, WR, MT, NT, MAXP, MINW, PIPE, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
    return rc;
}
template <typename T, int WR, int MT, int NT, int MINW>
static int set_lds_attr3() {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3<T, WR, MT, NT, MINW, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3<T, WR, MT, NT, MINW, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3<T, WR, MT, NT, MINW, false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
    if constexpr (sizeof(T) == 2 && MT == 2)
        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3<T, WR, MT, NT, MINW, false, false, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048 - 4096);
    return rc;
}
template <typename T>
static int set_lds_attr3_dma() {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3<T, 2, 4, 16, 1, false, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3<T, 2, 4, 16, 1, false, true, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536);
    return rc;
}
static NndetDevOnce g_attr_done;
static int ensure_attrs() {
    if (!g_attr_done.need()) return 0;
    int rc = 0;
    rc |= set_lds_attr3_dma<bf16_t>(); rc |= set_lds_attr3_dma<f16_t>();
    rc |= set_lds_attr<bf16_t, 1, 2, 8, 16, 2>(); rc |= set_lds_attr<bf16_t, 2, 2, 8, 16, 3>(); rc |= set_lds_attr<bf16_t, 2, 2, 4, 24, 3, true>(); rc |= set_lds_attr<bf16_t, 2, 1, 4, 24, 4, true>(); rc |= set_lds_attr<bf16_t, 1, 2, 4, 16, 4>();
    rc |= set_lds_attr<float, 1, 2, 8, 16, 2>(); rc |= set_lds_attr<float, 2, 2, 8, 16, 3>(); rc |= set_lds_attr<float, 2, 2, 4, 24, 3, true>(); rc |= set_lds_attr<float, 2, 1, 4, 24, 4, true>(); rc |= set_lds_attr<float, 1, 2, 4, 16, 4>();
    rc |= set_lds_attr<bf16_t, 2, 2, 2, 16, 3, true>(); rc |= set_lds_attr<bf16_t, 4, 1, 4, 16, 3, true>(); rc |= set_lds_attr<bf16_t, 2, 1, 2, 16, 4, true>();
    rc |= set_lds_attr<float, 2, 2, 2, 16, 3, true>(); rc |= set_lds_attr<float, 4, 1, 4, 16, 3, true>(); rc |= set_lds_attr<float, 2, 1, 2, 16, 4, true>();
    rc |= set_lds_attr3<bf16_t, 1, 2, 8, 2>(); rc |= set_lds_attr3<bf16_t, 2, 2, 8, 3>(); rc |= set_lds_attr3<bf16_t, 2, 2, 16, 2>();
    rc |= set_lds_attr3<float, 1, 2, 8, 2>(); rc |= set_lds_attr3<float, 2, 2, 8, 3>(); rc |= set_lds_attr3<float, 2, 2, 16, 2>();
    rc |= set_lds_attr<f16_t, 1, 2, 8, 16, 2>(); rc |= set_lds_attr<f16_t, 2, 2, 8, 16, 3>(); rc |= set_lds_attr<f16_t, 2, 2, 4, 24, 3, true>(); rc |= set_lds_attr<f16_t, 2, 1, 4, 24, 4, true>(); rc |= set_lds_attr<f16_t, 1, 2, 4, 16, 4>();
    rc |= set_lds_attr<f16_t, 2, 2, 2, 16, 3, true>(); rc |= set_lds_attr<f16_t, 4, 1, 4, 16, 3, true>(); rc |= set_lds_attr<f16_t, 2, 1, 2, 16, 4, true>();
    rc |= set_lds_attr3<f16_t, 1, 2, 8, 2>(); rc |= set_lds_attr3<f16_t, 2, 2, 8, 3>(); rc |= set_lds_attr3<f16_t, 2, 2, 16, 2>();
    if (rc) return rc;
    g_attr_done.done();
    return 0;
}

// persistent register-weight kernel: bf16, 32 -> 32 (padded) channels, every dim a multiple of the 8x8x8 tile, enough tiles to
// amortise the 55 KB weight load per workgroup. NNDET_IG3R=0 disables it (read per call so tests can flip it).
static bool ig3r_applicable(const NndetConv* c, const Plan& P) {
    const char* e = getenv("NNDET_IG3R");
    if (e && atoi(e) == 0) return false;
    const IgArgs& a = P.a;
    if (!nndet_is16(c->dtype) || P.cfg != 5 || a.Cx != 32 || a.Cy != 32) return false;
    for (int i = 0; i < 3; ++i) if (a.I[i] != a.O[i] || (a.I[i] % 8) != 0) return false;
    const int64_t total = (int64_t)a.N * (a.I[0] / 8) * (a.I[1] / 8) * (a.I[2] / 8);
    const int64_t min_tiles = (e && atoi(e) == 2) ? 1 : 2048;           // 2 = always (tests)
    return total >= min_tiles && total < (1 << 24);
}
static int ig3r_launch(const Plan& P, int dtype, hipStream_t st) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) return (int)hipErrorInvalidValue;
        n_cu = v & ~7;
        int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3r<bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536);
        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3r<bf16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536);
        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3r<f16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536);
        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3r<f16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536);
        if (rc) { n_cu = 0; return rc; }
    }
    const IgArgs& a = P.a;
    auto magic = [](int d) -> uint32_t { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)d + 1ull); };
    Ig3rArgs r;
    memset(&r, 0, sizeof(r));
    r.x = a.x; r.w = a.w; r.bias = a.bias; r.y = a.y; r.stats = a.stats;
    r.N = a.N; r.D = a.I[0]; r.H = a.I[1]; r.W = a.I[2];
    r.nt1 = r.H / 8; r.nt2 = r.W / 8; r.ntiles = (r.D / 8) * r.nt1 * r.nt2;
    r.m_tiles = magic(r.ntiles); r.m_hw = magic(r.nt1 * r.nt2); r.m_w = magic(r.nt2);
    r.rev = a.taps[0].d[0] != 0;
    r.total = r.N * r.ntiles;
    r.per_xcd = ceil_div(r.total, 8);
    int grid = n_cu;
    const char* ge = getenv("NNDET_IG3R_GRID");                        // tests: few workgroups -> many tiles (and images) per workgroup
    if (ge && atoi(ge) >= 8) grid = atoi(ge) & ~7;
    while (grid > 8 && (grid / 8) > r.per_xcd) grid -= 8;
    if (dtype == NNDET_F16) { if (r.stats) k_ig3r<f16_t, true><<<grid, 256, 2 * 65536, st>>>(r); else k_ig3r<f16_t, false><<<grid, 256, 2 * 65536, st>>>(r); }
    else { if (r.stats) k_ig3r<bf16_t, true><<<grid, 256, 2 * 65536, st>>>(r); else k_ig3r<bf16_t, false><<<grid, 256, 2 * 65536, st>>>(r); }
    LAUNCH_CHECK();
    return 0;
}

size_t igemm_splitk_bytes(const NndetConv* c, int kind) {
    if (pw_covers(c, kind)) return 0;
    if (kind == 1 && dgs_covers(c)) return 0;
    Plan P;
    if (build_plan(c, kind, &P)) return 0;
    return P.ksplit > 1 ? P.splitk_bytes : 0;
}

template <typename T>
static int splitk_reduce_launch(const Plan& P, hipStream_t st) {
    const IgArgs& a = P.a;
    const int64_t vox = (int64_t)a.O[0] * a.O[1] * a.O[2];
    dim3 g((unsigned)ceil_div64(vox, 32), a.Cy / 32, a.N);
    k_ig_splitk_reduce<T><<<g, 256, 0, st>>>(a.part, a.ksplit, a.N, vox, a.Cy, a.bias, reinterpret_cast<const T*>(a.res),
                                            reinterpret_cast<T*>(a.y), a.stats);
    LAUNCH_CHECK();
    return 0;
}

static bool ig3r_applicable(const NndetConv* c, const Plan& P);
// Does the data gradient of this convolution run in a k_ig3 launch that can also take the norm-backward sums (igemm_run's `nr`)?
int ig3_fuses_norm_reduce(const NndetConv* c) {
    static const int on = getenv("NNDET_IG3_NORMRED") ? atoi(getenv("NNDET_IG3_NORMRED")) : 1;
    if (!on || !nndet_is16(c->dtype) || c->transposed || c->in_affine || c->cin_p == 1) return 0;
    for (int i = 0; i < 3; ++i) if (c->k[i] != 3 || c->s[i] != 1 || c->p[i] != 1) return 0;
    Plan P;
    if (build_plan(c, 1, &P)) return 0;
    return (P.cfg == 5 || P.cfg == 6 || P.cfg == 7) && P.a.ncls == 1 && !ig3r_applicable(c, P);
}

int igemm_run(const NndetConv* c, int kind, const void* x, const void* w, const float* bias, const void* res, void* y,
              double* stats, hipStream_t st, float* dbias, void* ws, size_t ws_bytes, const DgsNormRed* nr) {
    if (nr) {                                                            // (the caller asked ig3_fuses_norm_reduce)
        if (kind != 1 || bias || stats || res || dbias || !ig3_fuses_norm_reduce(c)) return NNDET_EINVAL;
        Plan P;
        int rc = build_plan(c, 1, &P);
        if (rc) return rc;
        rc = ensure_attrs();
        if (rc) return rc;
        P.a.x = x; P.a.w = w; P.a.y = y;
        P.a.ny = nr->y; P.a.nmr = nr->mean_rstd; P.a.ngamma = nr->gamma; P.a.nbeta = nr->beta; P.a.nred = nr->red_ws;
        P.a.nrelu = nr->relu; P.a.ncout = nr->c;
        return c->dtype == NNDET_F16 ? launch_cfg_nb<f16_t>(P, st) : launch_cfg_nb<bf16_t>(P, st);
    }
    if (!stats && !(kind == 0 && c->in_affine && c->transposed)) {   // pointwise problems (1x1x1, transposed k == s) stream straight from global memory
        const int prc = pw_run(c, kind, x, w, bias, res, y, st, dbias);
        if (prc != 1) return prc;
    }
    if (dbias) return NNDET_EINVAL;     // only the pointwise data-gradient kernels fuse the bias gradient (nndet_conv3d_dgrad_fuses_bias)
    if (kind == 1 && !bias && !stats) {                                 // strided 3x3x3 data gradient: every parity class from ONE staged halo
        const int drc = dgs_run(c, x, w, res, y, st);
        if (drc != 1) return drc;
    }
    if (kind == 0) {                                                    // forward of the 32 -> 64 stride-2 transition: k_ig3s
        const int src = ig3s_run(c, kind, x, w, bias, res, y, stats, st);
        if (src != 1) return src;
    }
    Plan P;
    int rc = build_plan(c, kind, &P);
    if (rc) return rc;
    rc = ensure_attrs();
    if (rc) return rc;
    P.a.x = x; P.a.w = w; P.a.bias = bias; P.a.y = y; P.a.stats = stats; P.a.res = res;
    P.a.ss = nullptr; P.a.ss_relu = 0;
    if (kind == 0 && c->in_affine) {
        if (c->transposed) return NNDET_EINVAL;
        P.a.ss = c->in_affine; P.a.ss_relu = c->in_relu;
    }
    if (!P.a.ss && !res && ig3r_applicable(c, P)) return ig3r_launch(P, c->dtype, st);
    const bool split = P.ksplit > 1 && ws && ws_bytes >= P.splitk_bytes;
    if (split) {
        P.a.ksplit = P.ksplit; P.a.part = reinterpret_cast<float*>(ws);
        P.grid.x *= P.ksplit;
    }
#define IG_CFG_SS(T_) launch_cfg<T_, true>(P, st)
#define IG_CFG_NS(T_) launch_cfg<T_, false>(P, st)
    if (P.a.ss) rc = NNDET_DISPATCH_DTYPE(c->dtype, IG_CFG_SS);
    else rc = NNDET_DISPATCH_DTYPE(c->dtype, IG_CFG_NS);
    if (rc || !split) return rc;
#define IG_SK_RED(T_) splitk_reduce_launch<T_>(P, st)
    return NNDET_DISPATCH_DTYPE(c->dtype, IG_SK_RED);
}

// ------------------------------------------------------------------------------------------------ ragged batches (NndetItems)
// 3x3x3 / stride 1 / pad 1 forward or data gradient over items of different spatial size in ONE k_ig3 launch: grid.z = item,
// grid.x = tiles of the largest item (the others exit early), same compile-time tile, same tap / accumulation order as the per-level
// launches, so every output element is bit-identical to the one the uniform entry point produces for that level.
template <typename T>
static int launch_items(const Plan& P, const IgItems& it, hipStream_t st) {
    switch (P.cfg) {
        case 5: k_ig3<T, 1, 2, 8, 2, false, true><<<P.grid, 256, P.lds, st>>>(P.a, it); break;
        case 6: k_ig3<T, 2, 2, 8, 3, false, true><<<P.grid, 256, P.lds, st>>>(P.a, it); break;
        case 7: k_ig3<T, 2, 2, 16, 2, false, true><<<P.grid, 256, P.lds, st>>>(P.a, it); break;
        case 11: if constexpr (sizeof(T) == 2) { k_ig3<T, 2, 4, 16, 1, false, true, true><<<P.grid, 256, 2 * 65536, st>>>(P.a, it); break; } else return NNDET_EINVAL;
        default: return NNDET_EINVAL;
    }
    LAUNCH_CHECK();
    return 0;
}

int items_check(const NndetConv* c, const NndetItems* it) {
    if (!c || !it || it->n_items < 1 || it->n_items > NNDET_MAX_ITEMS) return NNDET_EINVAL;
    if (c->transposed || c->in_affine || c->cin_p % 32 || c->cout_p % 32) return NNDET_EINVAL;
    for (int i = 0; i < 3; ++i) if (c->k[i] != 3 || c->s[i] != 1 || c->p[i] != 1) return NNDET_EINVAL;
    const int esz = nndet_esize(c->dtype);
    const int cmax = c->cin_p > c->cout_p ? c->cin_p : c->cout_p;
    for (int i = 0; i < it->n_items; ++i) {
        const int32_t* d = it->dims[i];
        if (d[0] <= 0 || d[1] <= 0 || d[2] <= 0 || it->row_off[i] < 0) return NNDET_EINVAL;
        if ((int64_t)d[0] * d[1] * d[2] * cmax * esz >= (1LL << 31)) return NNDET_EINVAL;      // 32-bit buffer offsets per item
    }
    return 0;
}

int igemm_items_run(const NndetConv* c, const NndetItems* it, int kind, const void* x, const void* w, const float* bias, void* y,
                    double* stats, hipStream_t st) {
    int rc = items_check(c, it);
    if (rc) return rc;
    NndetConv cc = *c;                          // the plan of the bounding volume: channels, taps, tile kind, grid.x
    cc.batch = it->n_items;
    int mx[3] = {0, 0, 0};
    for (int i = 0; i < it->n_items; ++i) for (int a = 0; a < 3; ++a) if (it->dims[i][a] > mx[a]) mx[a] = it->dims[i][a];
    cc.in_d = cc.out_d = mx[0]; cc.in_h = cc.out_h = mx[1]; cc.in_w = cc.out_w = mx[2];
    Plan P;
    rc = build_plan(&cc, kind, &P, true);
    if (rc) return rc;
    if ((P.cfg < 5 || P.cfg > 7) && P.cfg != 11) return NNDET_EINVAL;
    rc = ensure_attrs();
    if (rc) return rc;
    P.a.x = x; P.a.w = w; P.a.bias = bias; P.a.y = y; P.a.stats = stats; P.a.res = nullptr; P.a.ss = nullptr; P.a.ss_relu = 0;
    IgItems ig;
    memset(&ig, 0, sizeof(ig));
    ig.n = it->n_items;
    const int TD = P.a.T[0];
    int max_tiles = 0;
    for (int i = 0; i < it->n_items; ++i) {
        for (int a = 0; a < 3; ++a) ig.dims[i][a] = it->dims[i][a];
        ig.row_off[i] = it->row_off[i];
        const int t = ceil_div(it->dims[i][0], TD) * ceil_div(it->dims[i][1], 8) * ceil_div(it->dims[i][2], 8);
        if (t > max_tiles) max_tiles = t;
    }
    P.grid.x = max_tiles;                       // (the bounding volume of items with different aspect ratios could ask for more)
#define IG_ITEMS(T_) launch_items<T_>(P, ig, st)
    return NNDET_DISPATCH_DTYPE(c->dtype, IG_ITEMS);
}

// ------------------------------------------------------------------------------------------------ weight packing
template <typename T>
__global__ void k_pack(const float* __restrict__ w, T* __restrict__ out, int R, int K, int Rp, int Kp, int taps,
                       int64_t sr, int64_t sk, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % Kp);
    int64_t t2 = i / Kp;
    const int r = (int)(t2 % Rp);
    const int t = (int)(t2 / Rp);
    float v = 0.f;
    if (r < R && k < K) v = w[r * sr + k * sk + t];
    out[i] = Elem<T>::st(v);
}

static void pack_dims(const NndetConv* c, int mode, int* R, int* K, int* Rp, int* Kp, int* taps, int64_t* sr, int64_t* sk) {
    *taps = c->k[0] * c->k[1] * c->k[2];
    const int64_t T = *taps;
    if (mode == 0) { *R = c->cout; *Rp = c->cout_p; *K = c->cin; *Kp = c->cin_p; }
    else { *R = c->cin; *Rp = c->cin_p; *K = c->cout; *Kp = c->cout_p; }
    if (!c->transposed) {   // [Cout][Cin][T]
        if (mode == 0) { *sr = (int64_t)c->cin * T; *sk = T; } else { *sr = T; *sk = (int64_t)c->cin * T; }
    } else {                // [Cin][Cout][T]
        if (mode == 0) { *sr = T; *sk = (int64_t)c->cout * T; } else { *sr = (int64_t)c->cout * T; *sk = T; }
    }
}

extern "C" size_t nndet_packed_weight_elems(const NndetConv* c, int32_t mode) {
    int R, K, Rp, Kp, taps; int64_t sr, sk;
    pack_dims(c, mode, &R, &K, &Rp, &Kp, &taps, &sr, &sk);
    return (size_t)taps * Rp * Kp;
}

extern "C" int nndet_pack_weight(const NndetConv* c, int32_t mode, const float* w, void* packed, void* stream) {
    if (!c || !w || !packed || (mode != 0 && mode != 1)) return NNDET_EINVAL;
    int R, K, Rp, Kp, taps; int64_t sr, sk;
    pack_dims(c, mode, &R, &K, &Rp, &Kp, &taps, &sr, &sk);
    const int64_t total = (int64_t)taps * Rp * Kp;
    const unsigned nb = (unsigned)ceil_div64(total, 256);
    if (c->dtype == NNDET_BF16)
        k_pack<bf16_t><<<nb, 256, 0, as_stream(stream)>>>(w, (bf16_t*)packed, R, K, Rp, Kp, taps, sr, sk, total);
    else if (c->dtype == NNDET_F16)
        k_pack<f16_t><<<nb, 256, 0, as_stream(stream)>>>(w, (f16_t*)packed, R, K, Rp, Kp, taps, sr, sk, total);
    else
        k_pack<float><<<nb, 256, 0, as_stream(stream)>>>(w, (float*)packed, R, K, Rp, Kp, taps, sr, sk, total);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ batched weight packing
// All convolutions of a model re-pack their weights after every optimizer step: 66 tiny launches per step (0.4 ms of GPU time and as
// much host time). One launch per up to PACK_MAX_JOBS (layer, mode) pairs instead; the job table travels as a kernel argument.
#define PACK_MAX_JOBS 40
struct PackJob { const float* w; void* out; int64_t sr, sk, total; int32_t R, K, Rp, Kp, dtype, taps; };
struct PackJobs { PackJob j[PACK_MAX_JOBS]; };

// Round 6: tiled through LDS. The first form walked the OUTPUT linearly and gathered w[r * sr + k * sk + t]: consecutive threads 27 (or
// Cin * 27) floats apart, every 128-byte line of the fp32 weights fetched by 27 threads -- 156 us per training step for 76 MB in + 76 MB
// out, at the very start of every forward pass. A tile = 32 k x 4 r x taps: read as contiguous runs of the source (the dimension whose
// stride is `taps` is the inner one: 32 x taps floats per r in mode 0, 4 x taps per k in mode 1), written as 32 consecutive k per (tap, r).
__global__ __launch_bounds__(256) void k_pack_batched(const PackJobs J) {
    __shared__ float tile[4 * 32 * 27];
    const PackJob& job = J.j[blockIdx.y];
    const int taps = job.taps;
    const bool k_inner = job.sk == taps;                 // [.. r ..][k][t] (mode 0 of Conv3d, mode 1 of ConvTranspose3d) or [.. k ..][r][t]
    const int tk = (job.Kp + 31) / 32, tr = (job.Rp + 3) / 4;
    const int ne = 128 * taps;
    for (int tl = blockIdx.x; tl < tk * tr; tl += gridDim.x) {
        const int k0 = (tl % tk) * 32, r0 = (tl / tk) * 4;
        __syncthreads();
        for (int e = threadIdx.x; e < ne; e += 256) {
            float v = 0.f;
            if (k_inner) {
                const int rl = e / (32 * taps), rem = e - rl * 32 * taps;        // rem = k_l * taps + t: contiguous in the source
                const int kl = rem / taps;
                if (r0 + rl < job.R && k0 + kl < job.K) v = job.w[(r0 + rl) * job.sr + (int64_t)k0 * taps + rem];
                tile[e] = v;                                                     // [r_l][k_l][t]
            } else {
                const int kl = e / (4 * taps), rem = e - kl * 4 * taps;          // rem = r_l * taps + t
                const int rl = rem / taps, t = rem - rl * taps;
                if (r0 + rl < job.R && k0 + kl < job.K) v = job.w[(k0 + kl) * job.sk + (int64_t)r0 * taps + rem];
                tile[(rl * 32 + kl) * taps + t] = v;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < ne; e += 256) {
            const int t = e >> 7, rl = (e >> 5) & 3, kl = e & 31;
            if (r0 + rl >= job.Rp || k0 + kl >= job.Kp) continue;
            const float v = tile[(rl * 32 + kl) * taps + t];
            const int64_t o = ((int64_t)t * job.Rp + r0 + rl) * job.Kp + k0 + kl;
            if (job.dtype == NNDET_BF16) reinterpret_cast<bf16_t*>(job.out)[o] = f32_to_bf16(v);
            else if (job.dtype == NNDET_F16) reinterpret_cast<f16_t*>(job.out)[o] = (f16_t)v;
            else reinterpret_cast<float*>(job.out)[o] = v;
        }
    }
}

extern "C" int nndet_pack_weights_batched(const NndetConv* convs, const int32_t* modes, const float* const* w, void* const* out,
                                          int32_t n, void* stream) {
    if (n < 0 || (n > 0 && (!convs || !modes || !w || !out))) return NNDET_EINVAL;
    for (int base = 0; base < n; base += PACK_MAX_JOBS) {
        PackJobs J;
        memset(&J, 0, sizeof(J));
        const int cnt = n - base < PACK_MAX_JOBS ? n - base : PACK_MAX_JOBS;
        int64_t maxtotal = 1;
        for (int q = 0; q < cnt; ++q) {
            const NndetConv* c = convs + base + q;
            const int mode = modes[base + q];
            if ((mode != 0 && mode != 1) || !w[base + q] || !out[base + q]) return NNDET_EINVAL;
            int R, K, Rp, Kp, taps; int64_t sr, sk;
            pack_dims(c, mode, &R, &K, &Rp, &Kp, &taps, &sr, &sk);
            PackJob& j = J.j[q];
            j.w = w[base + q]; j.out = out[base + q]; j.sr = sr; j.sk = sk; j.total = (int64_t)taps * Rp * Kp;
            j.R = R; j.K = K; j.Rp = Rp; j.Kp = Kp; j.dtype = c->dtype; j.taps = taps;
            if (taps > 27 || (sr != taps && sk != taps)) return NNDET_EINVAL;       // (one of the two channel dimensions is the inner one)
            const int64_t tiles = (int64_t)ceil_div(Kp, 32) * ceil_div(Rp, 4);
            if (tiles > maxtotal) maxtotal = tiles;
        }
        int64_t bx = maxtotal;
        if (bx > 512) bx = 512;
        k_pack_batched<<<dim3((unsigned)bx, cnt), 256, 0, as_stream(stream)>>>(J);
        LAUNCH_CHECK();
    }
    return 0;
}
