// Weight gradient of (transposed) 3D convolutions on MFMA for gfx950.
//
//   dW[r][k][tap] = sum_{n, lattice point i}  P[n, i][r] * Q[n, step * i + delta_tap][k]
//     Conv3d:           P = dY (rows r = Cout), Q = X  (k = Cin),  lattice = output grid, step = stride
//     ConvTranspose3d:  P = X  (rows r = Cin),  Q = dY (k = Cout), lattice = input grid,  step = stride (= kernel)
//
// The contraction runs over VOXELS, but NDHWC keeps channels contiguous, so the MFMA operands (which want
// the contraction index contiguous per lane) need a transposition. It happens on the LDS read side: tiles
// are staged [voxel][32 channels] (+ 32/64 B of padding per 8-voxel row so the q-groups of a wave hit disjoint
// banks); bf16 fragments are fetched with gfx950's transpose read ds_read_b64_tr_b16 (2 per fragment, voxel
// offset free per lane, so the 27 tap shifts need no shifted copies), fp32 fragments with scalar reads.
//
// Workgroup = 4 waves (one per SIMD, up to 512 registers each), owns a 32(r) x 32(k) block of dW for ALL taps
// and loops over a slice of the spatial tiles (TD x TH x 8 lattice points, halo of Q staged once and shared by
// all taps). Waves split the taps (>= 4 taps) or the contraction steps (< 4 taps). Accumulators stay in
// registers for the whole slice; the slice result is added to dW (PyTorch layout, fp32) with atomics.
// Software pipeline (bf16): the global loads of tile t+1 are issued into registers BEFORE the MFMA phase of
// tile t and written to LDS after it, so HBM/L2 latency hides behind the matrix work (the kernel was
// wait-bound: SQ_WAIT_ANY 52 %, MFMA busy 6.6 % without it, profiles/round1_pmc_*.txt).
#include "common.h"
#include "conv_common.h"


struct WgTap { int32_t d[3]; int32_t wt; };
struct WgArgs {
    const void* p; const void* q; float* dw;
    const float* qss;     // deferred norm of Q (= X of a Conv3d): [N][Cq][2] (scale, shift), NULL = off (NndetConv.in_affine)
    int32_t qss_relu;
    float* dbias;         // k_wgrad3 only: column sums of P (= dY) accumulated by the otherwise idle 28th tap slot; NULL = off
    float* part;          // partial results [S][rb*kb][ntap][32][32] fp32 (reduced by k_wgrad_reduce)
    int32_t N;
    int32_t PL[3], Cp;
    int32_t QD[3], Cq;
    int32_t step[3], qbase[3];
    int32_t TD, TH;
    int32_t nt[3];
    int32_t H[3];
    int32_t R, K;
    int64_t sr, sk;
    int32_t ntap, total_tiles;
    uint32_t mH2, mH1;    // magic multipliers for / H[2], / H[1] (0 = divisor 1)
    int32_t lTH;          // log2(TH)
    WgTap taps[27];
};

// Ragged batch (NndetItems) for k_wgrad3<..., ITEMS = true>: the persistent tile walk runs over the tiles of ALL items
// (tile_begin = running tile count), so the weight gradient of parameters shared by the items is summed inside the kernel
struct WgItems {
    int32_t n, pad_;
    int32_t dims[NNDET_MAX_ITEMS][3];
    int32_t tile_begin[NNDET_MAX_ITEMS];
    int64_t row_off[NNDET_MAX_ITEMS];
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// One MFMA operand fragment = this lane's 8 contraction values (8 consecutive lattice points of its q-group's run)
// of channel (tile base + li). `run0` points at [first voxel of the run][first channel of the 16-channel tile],
// `vstride` = bytes between consecutive voxels of the run.
template <typename T> struct WF {        // the 16-bit storage types (bf16_t, f16_t)
    u32x4 v;
    // 16-bit types: the LDS transpose read (ds_read_b64_tr_b16, layout verified by tools/probe_mfma.hip): the 16 lanes of a
    // q-group each fetch 8 bytes (4 channels of voxel li>>2) and receive, transposed, 4 voxels of channel li.
    // Two reads (voxels 0-3, 4-7) replace 8 scalar LDS reads; the voxel offset is per lane, so tap shifts cost nothing.
    __device__ __forceinline__ void load(const char* run0, int vstride, int li) {
        const char* p = run0 + (li >> 2) * vstride + (li & 3) * 8;
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
        const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * vstride));
        const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
        v = u32x4{ua.x, ua.y, ub.x, ub.y};
    }
    __device__ __forceinline__ void set_ones() { v = u32x4{H16<T>::ONE2, H16<T>::ONE2, H16<T>::ONE2, H16<T>::ONE2}; }   // 1.0 x 8
    typedef f32x4 acc_t;
    __device__ static __forceinline__ acc_t zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ static __forceinline__ int row(int q, int rr) { return q * 4 + rr; }      // row of register rr of lane row q in a 16 x 16 result
    __device__ static __forceinline__ void mma(const WF& a, const WF& b, f32x4& c) { c = H16<T>::mma(a.v, b.v, c); }
};
template <> struct WF<float> {
    float v[8];
    __device__ __forceinline__ void load(const char* run0, int vstride, int li) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float*>(run0 + j * vstride + li * 4);
    }
    __device__ __forceinline__ void set_ones() {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 1.f;
    }
    // the parity path accumulates in float64 (conv_igemm.hip: Mma<float>): a weight gradient is a sum over up to 4e7 voxels, one fp32
    // accumulator chain per workgroup slice was tens of thousands of roundings long. v_mfma_f64_16x16x4_f64 leaves row q + 4 rr in
    // register rr of lane row q (tools/probe_mfma64.hip): the epilogues index the partial-sum tile through row().
    typedef f64x4_t acc_t;
    __device__ static __forceinline__ acc_t zero() { return f64x4_t{0.0, 0.0, 0.0, 0.0}; }
    __device__ static __forceinline__ int row(int q, int rr) { return q + 4 * rr; }
    __device__ static __forceinline__ void mma(const WF& a, const WF& b, acc_t& c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a.v[j], (double)b.v[j], c, 0, 0, 0);
    }
};

// NTS = tap slots per wave (compile time: the MFMA phase is straight-line code, the compiler software-pipelines the
// LDS transpose reads against the MFMAs); SPLIT = waves split the taps (else: the contraction steps)
// RBK = 32-row blocks of dW per workgroup (1 or 2). The Q halo of a tile is staged ONCE for all of them: with one row block per
// workgroup the strided transitions (stride 2: halo 1.5 x the tile's own voxels) read X once per row block and tile from HBM -- the
// workgroups of different row blocks sit on different XCDs -- i.e. 3 x the tensor for the 32 -> 64 transition, which bounded it.
template <typename T, int KS, int MAXP, int MAXQ, bool PF, int NTS, bool SPLIT, bool AFF = false, int RBK = 1>
__global__ __launch_bounds__(256) void k_wgrad(const WgArgs A) {
    constexpr int RB = 32 * (int)sizeof(T);      // bytes of one voxel's 32-channel block
    constexpr int PPV = RB / 16;                  // 16-byte pieces per voxel
    constexpr int E16 = 16 / (int)sizeof(T);      // elements per piece
    constexpr int POINTS = KS * 32;
    constexpr int PBYTES = POINTS * RB + (POINTS / 8) * (RB / 2);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sp = smem;
    char* const sq = smem + RBK * PBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int r0 = blockIdx.y * (32 * RBK), k0 = blockIdx.z * 32;
    const int H0 = A.H[0], H1 = A.H[1], H2 = A.H[2];
    const int QROW = H2 * RB + RB / 2;            // bytes of one halo row incl. padding
    const int NQP = H0 * H1 * H2 * PPV;           // number of Q pieces
    const int TH = A.TH;

    // ---- per-thread staging descriptors (relative to the tile origin; no divisions in the tile loop)
    int32_t qrel[MAXQ];   // part << 27 | hd << 18 | hh << 9 | hw ; -1 = no piece
    int32_t qdst[MAXQ];
#pragma unroll
    for (int s = 0; s < MAXQ; ++s) {
        const int pp = tid + s * 256;
        int32_t rel = -1, dst = 0;
        if (pp < NQP) {
            const int hv = pp / PPV, part = pp % PPV;
            const int row = A.mH2 ? (int)__umulhi((unsigned)hv, A.mH2) : hv;
            const int hw = hv - row * H2;
            const int hd = A.mH1 ? (int)__umulhi((unsigned)row, A.mH1) : row;
            const int hh = row - hd * H1;
            rel = (part << 27) | (hd << 18) | (hh << 9) | hw;
            dst = row * QROW + hw * RB + part * 16;
        }
        qrel[s] = rel; qdst[s] = dst;
    }
    int32_t prel[MAXP], pdst[MAXP];
#pragma unroll
    for (int s = 0; s < MAXP; ++s) {
        const int pp = tid + s * 256;
        int32_t rel = -1, dst = 0;
        if (pp < POINTS * PPV) {
            const int pt = pp / PPV, part = pp % PPV;
            const int tr = pt >> 3, pw = pt & 7;
            const int pd = tr >> A.lTH, ph = tr & (TH - 1);
            rel = (part << 27) | (pd << 18) | (ph << 9) | pw;
            dst = pt * RB + tr * (RB / 2) + part * 16;
        }
        prel[s] = rel; pdst[s] = dst;
    }
    // ---- per-lane fragment bases
    int qrow0[KS];   // halo row of lattice row (ks*4 + q) at tap offset 0
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int tr = ks * 4 + q;
        const int pd = tr >> A.lTH, ph = tr & (TH - 1);
        qrow0[ks] = (pd * A.step[0]) * H1 + ph * A.step[1];
    }
    // ---- work split between the waves: slot ts of wave wv handles tap (SPLIT ? wv + 4*ts : ts); slots beyond the
    // tap count recompute tap 0 and are discarded (keeps the MFMA phase branch-free)
    int tapoff[NTS];      // LDS byte offset of the tap inside the halo tile
    int tapw[NTS];        // weight tap index, -1 = unused slot
#pragma unroll
    for (int ts = 0; ts < NTS; ++ts) {
        const int t = SPLIT ? wv + ts * 4 : ts;
        const bool valid = t < A.ntap;
        const WgTap& tap = A.taps[valid ? t : 0];
        tapoff[ts] = (tap.d[0] * H1 + tap.d[1]) * QROW + tap.d[2] * RB;
        tapw[ts] = valid ? tap.wt : -1;
    }
    int qrowb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qrowb[ks] = qrow0[ks] * QROW;
    const int wstep = A.step[2] * RB;

    typename WF<T>::acc_t acc[NTS][2 * RBK][2];
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
        for (int i = 0; i < 2 * RBK; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = WF<T>::zero();

    const T* pbase = reinterpret_cast<const T*>(A.p);
    const T* qbase_ptr = reinterpret_cast<const T*>(A.q);
    const int tiles_per_n = A.nt[0] * A.nt[1] * A.nt[2];

    // staging registers: all loads of a tile are UNCONDITIONAL (invalid pieces read a clamped address and are
    // zeroed at commit) so that they are in flight together; a conditional load per piece would serialise
    // into one HBM round trip each.
    u32x4 vp[RBK * MAXP], vq[MAXQ];
    uint32_t okp = 0, okq = 0;
    int n_staged = 0;                             // image of the tile whose pieces sit in vq (deferred input norm)

    auto issue = [&](int tile) {
        const int n = tile / tiles_per_n;
        n_staged = n;
        int tt = tile - n * tiles_per_n;
        const int tw_i = tt % A.nt[2]; tt /= A.nt[2];
        const int th_i = tt % A.nt[1];
        const int td_i = tt / A.nt[1];
        const int l0d = td_i * A.TD, l0h = th_i * TH, l0w = tw_i * 8;
        const int q0d = l0d * A.step[0] + A.qbase[0], q0h = l0h * A.step[1] + A.qbase[1], q0w = l0w * A.step[2] + A.qbase[2];
        const T* pn = pbase + (int64_t)n * A.PL[0] * A.PL[1] * A.PL[2] * A.Cp + r0;
        const T* qn = qbase_ptr + (int64_t)n * A.QD[0] * A.QD[1] * A.QD[2] * A.Cq + k0;
        okp = 0; okq = 0;
#pragma unroll
        for (int s = 0; s < MAXP; ++s) {
            const int r = prel[s] < 0 ? 0 : prel[s];
            const int ld = l0d + ((r >> 18) & 511), lh = l0h + ((r >> 9) & 511), lw = l0w + (r & 511);
            const bool ok = prel[s] >= 0 && ld < A.PL[0] && lh < A.PL[1] && lw < A.PL[2];
            okp |= (uint32_t)ok << s;
            const int64_t off = ok ? ((int64_t)(ld * A.PL[1] + lh) * A.PL[2] + lw) * A.Cp + (r >> 27) * E16 : 0;
#pragma unroll
            for (int b = 0; b < RBK; ++b) vp[b * MAXP + s] = *reinterpret_cast<const u32x4*>(pn + off + (ok ? b * 32 : 0));
        }
#pragma unroll
        for (int s = 0; s < MAXQ; ++s) {
            if (s * 256 >= NQP) break;   // uniform
            const int rel = qrel[s];
            const int r = rel < 0 ? 0 : rel;
            const int qd = q0d + ((r >> 18) & 511), qh = q0h + ((r >> 9) & 511), qw = q0w + (r & 511);
            const bool ok = rel >= 0 && (unsigned)qd < (unsigned)A.QD[0] && (unsigned)qh < (unsigned)A.QD[1] && (unsigned)qw < (unsigned)A.QD[2];
            okq |= (uint32_t)ok << s;
            const int64_t off = ok ? ((int64_t)(qd * A.QD[1] + qh) * A.QD[2] + qw) * A.Cq + (r >> 27) * E16 : 0;
            vq[s] = *reinterpret_cast<const u32x4*>(qn + off);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int s = 0; s < MAXP; ++s)
            if (prel[s] >= 0) {
#pragma unroll
                for (int b = 0; b < RBK; ++b)
                    *reinterpret_cast<u32x4*>(sp + b * PBYTES + pdst[s]) = ((okp >> s) & 1u) ? vp[b * MAXP + s] : u32x4{0u, 0u, 0u, 0u};
            }
        float asc[E16], ash[E16];
        if constexpr (AFF) load_affine<E16>(A.qss, n_staged, A.Cq, k0 + (tid % PPV) * E16, asc, ash);
#pragma unroll
        for (int s = 0; s < MAXQ; ++s) {
            if (s * 256 >= NQP) break;   // uniform
            if (qrel[s] >= 0) {
                u32x4 v = vq[s];
                if constexpr (AFF) v = AffinePiece<T>::apply(v, asc, ash, A.qss_relu);
                *reinterpret_cast<u32x4*>(sq + qdst[s]) = ((okq >> s) & 1u) ? v : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (!SPLIT && (ks & 3) != wv) continue;
            // P fragments of this contraction step: point = ks*32 + q*8 + j, channels it*16 + li
            WF<T> pf[2 * RBK];
            {
                const int tr = ks * 4 + q;
                const char* b0 = sp + (tr * 8) * RB + tr * (RB / 2);
#pragma unroll
                for (int b = 0; b < RBK; ++b) {
                    pf[2 * b].load(b0 + b * PBYTES, RB, li);
                    pf[2 * b + 1].load(b0 + b * PBYTES + 16 * (int)sizeof(T), RB, li);
                }
            }
#pragma unroll
            for (int ts = 0; ts < NTS; ++ts) {
                const char* b0 = sq + qrowb[ks] + tapoff[ts];
                WF<T> qf[2];
                qf[0].load(b0, wstep, li);
                qf[1].load(b0 + 16 * (int)sizeof(T), wstep, li);
#pragma unroll
                for (int i = 0; i < 2 * RBK; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) WF<T>::mma(pf[i], qf[j], acc[ts][i][j]);
            }
        }
    };

    int tile = blockIdx.x;
    if (PF) {
        if (tile < A.total_tiles) { issue(tile); commit(); }
        __syncthreads();
        for (; tile < A.total_tiles; tile += gridDim.x) {
            const int next = tile + gridDim.x;
            if (next < A.total_tiles) issue(next);      // in flight during the MFMA phase
            compute();
            __syncthreads();                            // every wave is done reading the tile
            if (next < A.total_tiles) commit();
            __syncthreads();
        }
    } else {
        for (; tile < A.total_tiles; tile += gridDim.x) {
            __syncthreads();
            issue(tile);
            commit();
            __syncthreads();
            compute();
        }
    }
    // ---- slice result -> partial buffer (plain coalesced stores). Device-scope atomics on dW were the bottleneck of
    // v1/v2: 1536 workgroups x 27.6 K atomics onto the same 27.6 K addresses ran at ~15 G atomics/s (2.2-2.9 ms per call
    // independent of the layer's FLOPs); a two-stage reduction costs ~0.1 ms and is deterministic.
    const int64_t slice = SPLIT ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + wv;   // !SPLIT: every wave owns a full partial
#pragma unroll
    for (int b = 0; b < RBK; ++b) {
        float* part = A.part + (slice * (gridDim.y * RBK * gridDim.z) + (blockIdx.y * RBK + b) * gridDim.z + blockIdx.z) * ((int64_t)A.ntap * 1024);
#pragma unroll
        for (int ts = 0; ts < NTS; ++ts) {
            if (tapw[ts] >= 0) {
                const int t = SPLIT ? wv + ts * 4 : ts;
                float* pt = part + (int64_t)t * 1024;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) pt[(i * 16 + WF<T>::row(q, rr)) * 32 + j * 16 + li] = (float)acc[ts][2 * b + i][j][rr];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3x3, stride 1
// Specialisation of k_wgrad for the 3x3x3 / stride 1 / pad 1 layers (most of the weight-gradient FLOPs), built like k_ig3:
//   * compile-time tile 4 x 8 x 8 lattice points (Q halo 6 x 10 x 10), so every LDS transpose read is lane base + wave tap
//     offset + IMMEDIATE, and the staging descriptors need no registers;
//   * staging through buffer loads (per-image descriptor, 32-bit offsets, out-of-tensor pieces read as zero by the hardware
//     bounds check): no 64-bit address arithmetic, no select on the LDS write;
//   * the fragment reads are software-pipelined two steps (= 8 MFMAs) ahead of their use and pinned with sched_barrier
//     (hipcc otherwise issues them right before the MFMAs: with the generic kernel's 436 registers = one wave per SIMD every
//     LDS round trip was exposed, 25 % MFMA utilisation);
//   * 256 registers per wave -> two workgroups per CU.
// Work split as in k_wgrad: the 4 waves take taps wv, wv + 4, ... (7 slots, 27 of 28 used), all 8 contraction steps.
template <typename T, int MINW, bool AFF = false, int QDEPTH = 2, bool ITEMS = false>
__global__ __launch_bounds__(256, MINW) void k_wgrad3(const WgArgs A, const WgItems IT) {
    constexpr int RB = 32 * (int)sizeof(T), PPV = RB / 16;
    constexpr int KS = 8, NTS = 7, TD = 4, TH = 8, HD = TD + 2, HH = TH + 2, HW = 10;
    constexpr int PROW = 8 * RB + RB / 2, QROW = HW * RB + RB / 2;     // bytes per row of 8 points / per halo row (with bank padding)
    constexpr int PBYTES = 32 * PROW;
    constexpr int QCOLP = HW * PPV, QRPS = 256 / QCOLP, QNROW = HD * HH, QSTEPS = (QNROW + QRPS - 1) / QRPS;   // Q staging: rows per step
    constexpr int PSTEPS = 256 * PPV / 256;                                                                  // P pieces per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sp = smem;
    char* const sq = smem + PBYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int r0 = blockIdx.y * 32, k0 = blockIdx.z * 32;

    // ---- staging geometry (tile independent)
    // P: piece s of a thread = point (pd = s / (PPV/4)..) -- with PPV pieces per voxel: pp = tid + s*256, pt = pp / PPV
    // P piece s2 of a thread = point pt = tid / PPV + s2 * (256 / PPV): row tr = pt >> 3 = tr0 + s2 * TRSTEP, pw = pt & 7. Nothing of
    // this is kept in per-piece register arrays (round 1 did: p_ph / p_dst / p_rel = 12 VGPRs, and the bf16 kernel -- capped at 256
    // registers for two workgroups per CU -- spilled 13 of them to scratch; the scratch RELOADS sat between the staging loads of
    // issue(), and since scratch shares the vector-memory counter each forced s_waitcnt vmcnt(0): the four P loads of every tile
    // were serialised into four memory round trips).
    const int p_part = tid % PPV;
    constexpr int TRSTEP = 256 / PPV / 8;
    const int p_tr0 = (tid / PPV) >> 3, p_pw = (tid / PPV) & 7;
    const int p_dst0 = p_tr0 * PROW + p_pw * RB + p_part * 16;                      // + s2 * TRSTEP * PROW
    const int p_rowb = A.PL[2] * A.Cp * (int)sizeof(T), p_slab = A.PL[1] * p_rowb;  // bytes per lattice row / slice
    const int p_col = p_pw * A.Cp * (int)sizeof(T) + p_part * 16;
    // Q: thread = column piece (hw, part) of QRPS rows per step
    const int q_cp = tid % QCOLP, q_r0 = tid / QCOLP;
    const bool q_active = tid < QRPS * QCOLP;
    const int q_hw = q_cp / PPV;
    const int q_dst0 = q_r0 * QROW + q_hw * RB + (q_cp % PPV) * 16;
    const int q_colb = q_hw * A.Cq * (int)sizeof(T) + (q_cp % PPV) * 16;
    const int q_rowb = A.QD[2] * A.Cq * (int)sizeof(T);

    // ---- fragment bases: P run of lane group q at contraction step ks = row ks*4 + q; Q halo row of that run at tap (0,0,0)
    // per-lane part of a fragment address (WF<T>::load is then called with li = 0): bf16 transpose read / fp32 scalar reads
    const int lanepart = sizeof(T) == 2 ? (li >> 2) * RB + (li & 3) * 8 : li * 4;
    const char* const p_lane = sp + q * PROW + lanepart;
    const int q_lane = PBYTES + q * QROW + lanepart;
    int tapoff[NTS], tapw[NTS];
#pragma unroll
    for (int ts = 0; ts < NTS; ++ts) {
        const int t = wv + ts * 4;
        const bool valid = t < 27;
        const int tt = valid ? t : 0;
        const int a = tt / 9, b = (tt / 3) % 3, c = tt % 3;
        tapoff[ts] = __builtin_amdgcn_readfirstlane((a * HH + b) * QROW + c * RB);      // wave-uniform: keep them in SGPRs
        tapw[ts] = __builtin_amdgcn_readfirstlane(valid ? tt : -1);
    }

    typename WF<T>::acc_t acc[NTS][2][2];
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = WF<T>::zero();

    const int tiles_per_n = A.nt[0] * A.nt[1] * A.nt[2];
    const int p_img = A.PL[0] * A.PL[1] * A.PL[2] * A.Cp * (int)sizeof(T), q_img = A.QD[0] * A.QD[1] * A.QD[2] * A.Cq * (int)sizeof(T);
    u32x4 vp[PSTEPS], vq[QSTEPS];
    uint32_t okq = 0;                             // deferred input norm: which Q pieces are inside the tensor (padding stays zero)
    int n_staged = 0;

    auto issue = [&](int tile) {
        int n, tt, nt1 = A.nt[1], nt2 = A.nt[2];
        int PL0 = A.PL[0], PL1 = A.PL[1], PL2 = A.PL[2], QD0 = A.QD[0], QD1 = A.QD[1], QD2 = A.QD[2];
        int pi_bytes = p_img, qi_bytes = q_img, prb = p_rowb, psl = p_slab, qrb = q_rowb;
        int64_t p_base, q_base;
        if constexpr (ITEMS) {                      // `tile` is workgroup-uniform: a scalar walk over <= 32 running tile counts
            n = 0;
            while (n + 1 < IT.n && tile >= IT.tile_begin[n + 1]) ++n;
            tt = tile - IT.tile_begin[n];
            PL0 = QD0 = IT.dims[n][0]; PL1 = QD1 = IT.dims[n][1]; PL2 = QD2 = IT.dims[n][2];
            nt1 = (PL1 + TH - 1) / TH; nt2 = (PL2 + 7) / 8;
            prb = PL2 * A.Cp * (int)sizeof(T); psl = PL1 * prb; qrb = QD2 * A.Cq * (int)sizeof(T);
            pi_bytes = PL0 * psl; qi_bytes = QD0 * QD1 * qrb;
            p_base = IT.row_off[n] * A.Cp * (int64_t)sizeof(T); q_base = IT.row_off[n] * A.Cq * (int64_t)sizeof(T);
        } else {
            n = tile / tiles_per_n;
            tt = tile - n * tiles_per_n;
            p_base = (int64_t)n * p_img; q_base = (int64_t)n * q_img;
        }
        n_staged = n; okq = 0;
        const int tw_i = tt % nt2; tt /= nt2;
        const int th_i = tt % nt1;
        const int td_i = tt / nt1;
        const int l0d = td_i * TD, l0h = th_i * TH, l0w = tw_i * 8;
        const auto prs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(A.p)) + p_base + r0 * (int)sizeof(T), 0, pi_bytes - r0 * (int)sizeof(T), 0x00020000);
        const auto qrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(A.q)) + q_base + k0 * (int)sizeof(T), 0, qi_bytes - k0 * (int)sizeof(T), 0x00020000);
        const int p_org = ((l0d * PL1 + l0h) * PL2 + l0w) * A.Cp * (int)sizeof(T);
#pragma unroll
        for (int s2 = 0; s2 < PSTEPS; ++s2) {
            const int tr = p_tr0 + s2 * TRSTEP;
            const int pd = tr >> 3, ph = tr & 7;
            const int ld = l0d + pd, lh = l0h + ph, lw = l0w + p_pw;
            const bool ok = ld < PL0 && lh < PL1 && lw < PL2;
            vp[s2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, ok ? pd * psl + ph * prb + p_col : (int)0x80000000, p_org, 0));
        }
        const int q0d = l0d - 1, q0h = l0h - 1, qw = l0w - 1 + q_hw;
        const bool okw = q_active && (unsigned)qw < (unsigned)QD2;
        const int q_org = (l0w - 1) * A.Cq * (int)sizeof(T);
#pragma unroll
        for (int s2 = 0; s2 < QSTEPS; ++s2) {
            const int r = q_r0 + s2 * QRPS;
            const int hd = r / HH, hh = r - hd * HH;
            const int qd = q0d + hd, qh = q0h + hh;
            const bool ok = okw && (unsigned)qd < (unsigned)QD0 && (unsigned)qh < (unsigned)QD1 && (s2 + 1 < QSTEPS || r < QNROW);
            if constexpr (AFF) okq |= (uint32_t)ok << s2;
            vq[s2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                qrs, ok ? (qd * QD1 + qh) * qrb + q_colb + q_org : (int)0x80000000, 0, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int s2 = 0; s2 < PSTEPS; ++s2) *reinterpret_cast<u32x4*>(sp + p_dst0 + s2 * (TRSTEP * PROW)) = vp[s2];
        if (q_active) {
            constexpr int E16 = 16 / (int)sizeof(T);
            float asc[E16], ash[E16];
            if constexpr (AFF) load_affine<E16>(A.qss, n_staged, A.Cq, k0 + (q_cp % PPV) * E16, asc, ash);
#pragma unroll
            for (int s2 = 0; s2 < QSTEPS; ++s2)
                if (s2 + 1 < QSTEPS || q_r0 + s2 * QRPS < QNROW) {
                    if constexpr (AFF) vq[s2] = ((okq >> s2) & 1u) ? AffinePiece<T>::apply(vq[s2], asc, ash, A.qss_relu) : u32x4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4*>(sq + q_dst0 + s2 * (QRPS * QROW)) = vq[s2];
                }
        }
    };
    // wave 3's last slot would recompute tap 0 and discard it (27 taps on 28 slots): it accumulates sum_p dY[p][r] instead
    const bool do_bias = A.dbias != nullptr && blockIdx.z == 0 && wv == 3;
    // pinned software pipeline over the 56 (contraction step, tap slot) pairs
    auto compute = [&]() {
        constexpr int U = KS * NTS, QD_ = QDEPTH;   // LDS fragment prefetch distance (steps of 4 MFMAs)
        WF<T> pf[2][2], qf[QD_ + 1][2];
        auto load_p = [&](int ks, WF<T>* d) {
            const char* b0 = p_lane + ks * 4 * PROW;
            d[0].load(b0, RB, 0); d[1].load(b0 + 16 * (int)sizeof(T), RB, 0);
        };
        auto load_q = [&](int u, WF<T>* d) {
            const int ks = u / NTS, ts = u % NTS;
            const char* b0 = smem + (q_lane + tapoff[ts]) + (((ks >> 1) * HH) + (ks & 1) * 4) * QROW;
            if (ts == NTS - 1 && do_bias) {       // the 28th slot (wave 3): Q = ones -> acc = column sums of P = the bias gradient
                d[0].set_ones(); d[1].set_ones();
            } else {
                d[0].load(b0, RB, 0); d[1].load(b0 + 16 * (int)sizeof(T), RB, 0);
            }
        };
        load_p(0, pf[0]);
#pragma unroll
        for (int u0 = 0; u0 < QD_; ++u0) load_q(u0, qf[u0]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = u / NTS, ts = u % NTS;
            if (u + QD_ < U) load_q(u + QD_, qf[(u + QD_) % (QD_ + 1)]);
            if (ts == 0 && ks + 1 < KS) load_p(ks + 1, pf[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) WF<T>::mma(pf[ks & 1][i], qf[u % (QD_ + 1)][j], acc[ts][i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // round k of the persistent loop covers tiles [k * grid, (k + 1) * grid); within a round the workgroups of one XCD take a
    // contiguous run of tiles (xcd_compact) so that overlapping halos are shared through that XCD's L2
    int base = 0;
    auto tile_of = [&](int b0) { return b0 + xcd_compact(blockIdx.x, min((int)gridDim.x, A.total_tiles - b0), gridDim.x); };
    int tile = tile_of(0);
    const bool first_ok = (int)blockIdx.x < A.total_tiles;
    if (first_ok) { issue(tile); commit(); }
    __syncthreads();
    for (; base + (int)blockIdx.x < A.total_tiles; base += gridDim.x) {
        const int nb = base + gridDim.x;
        const bool has_next = nb + (int)blockIdx.x < A.total_tiles;
        if (has_next) issue(tile_of(nb));           // in flight during the MFMA phase
        compute();
        __syncthreads();                            // every wave is done reading the tile
        if (has_next) commit();
        __syncthreads();
    }
    if (do_bias && li == 0) {                // column 0 of the ones-GEMM: rows r0 + i*16 + 4q + rr
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = r0 + i * 16 + WF<T>::row(q, rr);
                if (r < A.R) atomicAdd(A.dbias + r, (float)acc[NTS - 1][i][0][rr]);
            }
    }
    float* part = A.part + ((int64_t)blockIdx.x * (gridDim.y * gridDim.z) + blockIdx.y * gridDim.z + blockIdx.z) * ((int64_t)27 * 1024);
#pragma unroll
    for (int ts = 0; ts < NTS; ++ts) {
        if (tapw[ts] >= 0) {
            float* pt = part + (int64_t)tapw[ts] * 1024;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) pt[(i * 16 + WF<T>::row(q, rr)) * 32 + j * 16 + li] = (float)acc[ts][i][j][rr];
        }
    }
}

static const WgItems g_wg_no_items = {};

// dW[r][k][tap] += sum_s part[s][pair][tap][r%32][k%32]; one thread per (pair, tap, r, k); consecutive threads = consecutive k.
// With !SPLIT (1-3 taps) every wave holds a partial of every tap: those are summed here too (nsub = 4 sub-slices).
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, int S, int pairs, int kb, int ntap,
                                                      int R, int K, int64_t sr, int64_t sk, float* __restrict__ dw, int64_t total) {
    // grid (ceil(total / 256), ceil(S / 32)): a thread sums 32 slices of one element (coalesced across threads), then
    // one atomic per thread: total * S / 32 atomics (< 1 M) instead of the 42 M of the atomics-only scheme.
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int k = (int)(e & 31), r = (int)((e >> 5) & 31);
    const int64_t pt = e >> 10;               // pair * ntap + tap
    const int t = (int)(pt % ntap), pair = (int)(pt / ntap);
    const int rbi = pair / kb, kbi = pair % kb;
    const int rg = rbi * 32 + r, kg = kbi * 32 + k;
    if (rg >= R || kg >= K) return;
    const int64_t stride = (int64_t)pairs * ntap * 1024;
    const int s0 = blockIdx.y * 32, s1 = min(S, s0 + 32);
    float acc = 0.f;
    const float* src = part + e + (int64_t)s0 * stride;
#pragma unroll 8
    for (int s2 = s0; s2 < s1; ++s2, src += stride) acc += *src;
    float* dst = &dw[rg * sr + kg * sk + t];
    if (gridDim.y == 1) *dst += acc;          // single writer per element (S <= 32): no atomic needed, deterministic
    else atomicAdd(dst, acc);
}

template <typename T, int KS, int MAXP, int MAXQ, bool PF, int NTS, bool SPLIT, int RBK = 1>
static int wg_launch(const WgArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<T, KS, MAXP, MAXQ, PF, NTS, SPLIT, false, RBK>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<T, KS, MAXP, MAXQ, PF, NTS, SPLIT, true, RBK>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        attr = true;
    }
    if (a.qss) k_wgrad<T, KS, MAXP, MAXQ, PF, NTS, SPLIT, true, RBK><<<grid, 256, lds, st>>>(a);      // deferred input norm applied while staging X
    else k_wgrad<T, KS, MAXP, MAXQ, PF, NTS, SPLIT, false, RBK><<<grid, 256, lds, st>>>(a);
    LAUNCH_CHECK();
    return 0;
}

// tap-slot dispatch: >= 4 taps are split over the 4 waves (slots = ceil(taps / 4) rounded up to an instantiated count),
// 1-3 taps are processed by every wave (the waves split the contraction steps instead)
template <typename T, int KS, int MAXP, int MAXQ, bool PF>
static int wg_dispatch(const WgArgs& a, dim3 grid, size_t lds, hipStream_t st, int rbk = 1) {
    if (rbk == 2) {      // strided 3x3x3 transitions in 16 bits (27 taps: 7 slots per wave), two row blocks per workgroup
        if constexpr (sizeof(T) == 2 && KS == 2) return wg_launch<T, KS, MAXP, MAXQ, PF, 7, true, 2>(a, grid, lds, st);
        return NNDET_EINVAL;
    }
    if (a.ntap >= 4) {
        const int need = (a.ntap + 3) / 4;
        if (need <= 1) return wg_launch<T, KS, MAXP, MAXQ, PF, 1, true>(a, grid, lds, st);
        if (need <= 2) return wg_launch<T, KS, MAXP, MAXQ, PF, 2, true>(a, grid, lds, st);
        if (need <= 3) return wg_launch<T, KS, MAXP, MAXQ, PF, 3, true>(a, grid, lds, st);
        if (need <= 5) return wg_launch<T, KS, MAXP, MAXQ, PF, 5, true>(a, grid, lds, st);
        return wg_launch<T, KS, MAXP, MAXQ, PF, 7, true>(a, grid, lds, st);
    }
    if (a.ntap == 1) return wg_launch<T, KS, MAXP, MAXQ, PF, 1, false>(a, grid, lds, st);
    return wg_launch<T, KS, MAXP, MAXQ, PF, 3, false>(a, grid, lds, st);
}

// number of spatial slices (= workgroups per channel-block pair) for a problem with `pairs` block pairs
static int wgrad_slices(int pairs, int total_tiles) {
    // 384 persistent workgroups over all block pairs = 1.5 per CU (two fit): inside the training step these kernels run on the
    // lowest-priority stream NEXT to the data-gradient chain, and with every slot of every CU taken by a long-running weight-gradient
    // workgroup the chain's kernels wait for slots -- 512: 13.84 / 13.86 ms per step, 448: 13.86 / 13.76, 384: 13.74 / 13.77, 352: 14.19,
    // 256: 14.2, 128: 18.1 (profiles/round3_ab_wgrad_wgs.txt; 192x192x128 patches: 23.14 -> 23.02). NNDET_WGRAD_WGS overrides (<= 512:
    // the workspace bound).
    const char* we = getenv("NNDET_WGRAD_WGS");
    int total = we ? atoi(we) : 384;
    if (total < 8 || total > 512) total = 384;
    int S = total / pairs;
    if (S < 1) S = 1;
    if (S > total_tiles) S = total_tiles;
    return S;
}

size_t wgrad_workspace_bytes(const NndetConv* c) {
    const bool tr = c->transposed != 0;
    const int rb = (tr ? c->cin_p : c->cout_p) / 32, kb = (tr ? c->cout_p : c->cin_p) / 32;
    const int ntap = c->k[0] * c->k[1] * c->k[2];
    // upper bound independent of the tile choice: S <= 1024 / pairs (>= 1), x4 sub-slices when the waves split k-steps
    int S = 512 / (rb * kb); if (S < 1) S = 1;
    const int slices = ntap >= 4 ? S : 4 * S;
    return (size_t)slices * rb * kb * ntap * 1024 * sizeof(float);
}

// uniform k_wgrad3 launch for a 16-bit storage type
template <typename T> static int wgrad3_launch16(const WgArgs& b, dim3 g3, size_t lds3, hipStream_t st) {
    static bool at = false;
    if (!at) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<T, 2, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<T, 2, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<T, 2, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        at = true;
    }
    // QDEPTH 1: 248 registers, no spill. With depth 2 the kernel needs > 256 registers at two workgroups per CU and the
    // spill reloads (scratch shares vmcnt) serialise the staging loads of every tile (profiles/round2_wgrad3_spill.txt)
    static const int qd = getenv("NNDET_WGRAD3_QD") ? atoi(getenv("NNDET_WGRAD3_QD")) : 1;
    if (b.qss) k_wgrad3<T, 2, true, 1><<<g3, 256, lds3, st>>>(b, g_wg_no_items);
    else if (qd == 2) k_wgrad3<T, 2, false, 2><<<g3, 256, lds3, st>>>(b, g_wg_no_items);
    else k_wgrad3<T, 2, false, 1><<<g3, 256, lds3, st>>>(b, g_wg_no_items);
    return 0;
}

int wgrad_run(const NndetConv* c, const void* x, const void* dy, float* dw, float* dbias, int* bias_done, void* ws, size_t ws_bytes,
              hipStream_t st) {
    WgArgs a;
    memset(&a, 0, sizeof(a));
    *bias_done = 0;
    const bool tr = c->transposed != 0;
    const bool bf = nndet_is16(c->dtype), hf = c->dtype == NNDET_F16;
    const int esz = bf ? 2 : 4;
    const int RB = 32 * esz, PPV = RB / 16;
    const int in_sp[3] = {c->in_d, c->in_h, c->in_w}, out_sp[3] = {c->out_d, c->out_h, c->out_w};
    const int64_t T = (int64_t)c->k[0] * c->k[1] * c->k[2];
    if (T > 27) return NNDET_EINVAL;
    a.N = c->batch; a.dw = dw;
    if (!tr) {   // P = dY, Q = X ; dW [Cout][Cin][T]
        a.p = dy; a.q = x; a.Cp = c->cout_p; a.Cq = c->cin_p; a.R = c->cout; a.K = c->cin;
        a.sr = (int64_t)c->cin * T; a.sk = T;
        for (int i = 0; i < 3; ++i) { a.PL[i] = out_sp[i]; a.QD[i] = in_sp[i]; a.step[i] = c->s[i]; a.qbase[i] = -c->p[i]; }
        a.qss = c->in_affine; a.qss_relu = c->in_relu;       // deferred input norm: X is read as relu?(x * scale + shift)
    } else {     // P = X, Q = dY ; dW [Cin][Cout][T]
        if (c->in_affine) return NNDET_EINVAL;
        for (int i = 0; i < 3; ++i) if (c->k[i] != c->s[i] || c->p[i] != 0) return NNDET_EINVAL;
        a.p = x; a.q = dy; a.Cp = c->cin_p; a.Cq = c->cout_p; a.R = c->cin; a.K = c->cout;
        a.sr = (int64_t)c->cout * T; a.sk = T;
        for (int i = 0; i < 3; ++i) { a.PL[i] = in_sp[i]; a.QD[i] = out_sp[i]; a.step[i] = c->s[i]; a.qbase[i] = 0; }
    }
    if (a.Cp % 32 || a.Cq % 32) return NNDET_EINVAL;
    a.ntap = (int)T;
    int nt = 0;
    for (int td = 0; td < c->k[0]; ++td) for (int th = 0; th < c->k[1]; ++th) for (int tw = 0; tw < c->k[2]; ++tw) {
        WgTap& t = a.taps[nt++];
        t.d[0] = td; t.d[1] = th; t.d[2] = tw; t.wt = (td * c->k[1] + th) * c->k[2] + tw;
    }
    const bool strided = a.step[0] > 1 || a.step[1] > 1 || a.step[2] > 1;
    const int KS = strided ? 2 : 8;
    const int rows = KS * 4;
    // Q pieces per thread supported by the instantiation chosen below (registers of the software pipeline)
    const int maxq = bf ? (strided ? 12 : 10) : (strided ? 24 : 20);
    // tile (TD, TH, 8): minimise padded volume, respect LDS and piece limits
    double best = 1e300; int bTD = 0, bTH = 0;
    for (int td = 1; td <= rows; td *= 2) {
        const int th = rows / td;
        const int t3[3] = {td, th, 8};
        int h[3]; int64_t hv = 1; double padded = 1.0;
        for (int i = 0; i < 3; ++i) {
            h[i] = (t3[i] - 1) * a.step[i] + c->k[i];
            hv *= h[i];
            padded *= (double)ceil_div(a.PL[i], t3[i]) * t3[i];
        }
        if (hv * PPV > 256 * maxq) continue;
        const size_t lds = (size_t)(KS * 32) * RB * 17 / 16 + (size_t)h[0] * h[1] * (h[2] * RB + RB / 2);
        if (lds > 150 * 1024) continue;
        const double cost = padded * (1.0 + 0.1 * (double)hv / (KS * 32));
        if (cost < best) { best = cost; bTD = td; bTH = th; }
    }
    if (!bTD) return NNDET_EINVAL;
    a.TD = bTD; a.TH = bTH;
    const int t3[3] = {bTD, bTH, 8};
    for (int i = 0; i < 3; ++i) { a.H[i] = (t3[i] - 1) * a.step[i] + c->k[i]; a.nt[i] = ceil_div(a.PL[i], t3[i]); }
    if (a.H[0] > 511 || a.H[1] > 511 || a.H[2] > 511) return NNDET_EINVAL;
    a.total_tiles = a.N * a.nt[0] * a.nt[1] * a.nt[2];
    auto magic = [](int d) -> uint32_t { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)d + 1ull); };
    a.mH2 = magic(a.H[2]); a.mH1 = magic(a.H[1]);
    a.lTH = 0; while ((1 << a.lTH) < a.TH) ++a.lTH;
    size_t lds = (size_t)(KS * 32) * RB + (size_t)(KS * 4) * (RB / 2) + (size_t)a.H[0] * a.H[1] * (a.H[2] * RB + RB / 2) + 64;
    const int rb = a.Cp / 32, kb = a.Cq / 32;
    const int S = wgrad_slices(rb * kb, a.total_tiles);
    dim3 grid(S, rb, kb);
    // two row blocks per workgroup (k_wgrad<..., RBK = 2>): the strided 3x3x3 transitions in 16 bits; NNDET_WGRAD_RBK=1 disables it
    const char* rbk_env = getenv("NNDET_WGRAD_RBK");
    const int rbk = (strided && bf && !tr && T == 27 && rb % 2 == 0 && (rbk_env ? atoi(rbk_env) : 2) == 2) ? 2 : 1;
    if (rbk == 2) {
        grid.y = rb / 2;
        lds += (size_t)(KS * 32) * RB + (size_t)(KS * 4) * (RB / 2);
    }
    const int slices = (a.ntap >= 4) ? S : 4 * S;
    const size_t need = (size_t)slices * rb * kb * a.ntap * 1024 * sizeof(float);
    if (!ws || ws_bytes < need) return NNDET_EWORKSPACE;
    a.part = reinterpret_cast<float*>(ws);
    int rc;
    // 3x3x3 / stride 1 / pad 1: compile-time-tile kernel k_wgrad3 unless the fixed (4,8,8) tile pads the volume noticeably
    // more than the tile found above (NNDET_WGRAD_SPEC: 0 never, 1 default, 2 always)
    const char* spec_env = getenv("NNDET_WGRAD_SPEC");
    const int spec_on = spec_env ? atoi(spec_env) : 1;
    if (spec_on && !tr && !strided && T == 27 && c->k[0] == 3 && c->k[1] == 3 && c->k[2] == 3 && c->p[0] == 1 && c->p[1] == 1 && c->p[2] == 1) {
        const int st3[3] = {4, 8, 8};
        double pg = 1.0, ps = 1.0;
        for (int i = 0; i < 3; ++i) { pg *= (double)a.nt[i] * t3[i]; ps *= (double)ceil_div(a.PL[i], st3[i]) * st3[i]; }
        const int64_t pb = (int64_t)a.PL[0] * a.PL[1] * a.PL[2] * a.Cp * esz, qb = (int64_t)a.QD[0] * a.QD[1] * a.QD[2] * a.Cq * esz;
        if ((ps <= 1.05 * pg || spec_on == 2) && pb < (1LL << 31) && qb < (1LL << 31)) {
            WgArgs b = a;
            b.TD = 4; b.TH = 8;
            b.dbias = dbias;                  // P = dY here (not transposed): the kernel also produces the bias gradient
            *bias_done = dbias != nullptr;
            for (int i = 0; i < 3; ++i) { b.H[i] = st3[i] + 2; b.nt[i] = ceil_div(a.PL[i], st3[i]); }
            b.total_tiles = b.N * b.nt[0] * b.nt[1] * b.nt[2];
            const int S3 = wgrad_slices(rb * kb, b.total_tiles);
            const size_t need3 = (size_t)S3 * rb * kb * 27 * 1024 * sizeof(float);
            if (ws_bytes < need3) return NNDET_EWORKSPACE;
            const size_t lds3 = (size_t)32 * (8 * RB + RB / 2) + (size_t)60 * (10 * RB + RB / 2);
            dim3 g3(S3, rb, kb);
            if (bf) {
                const int rc3 = hf ? wgrad3_launch16<f16_t>(b, g3, lds3, st) : wgrad3_launch16<bf16_t>(b, g3, lds3, st);
                if (rc3) return rc3;
            } else {
                static bool at = false;
                if (!at) {
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<float, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<float, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
                    at = true;
                }
                if (b.qss) k_wgrad3<float, 1, true><<<g3, 256, lds3, st>>>(b, g_wg_no_items);
                else k_wgrad3<float, 1, false><<<g3, 256, lds3, st>>>(b, g_wg_no_items);
            }
            LAUNCH_CHECK();
            const int64_t total3 = (int64_t)rb * kb * 27 * 1024;
            k_wgrad_reduce<<<dim3((unsigned)ceil_div64(total3, 256), ceil_div(S3, 32)), 256, 0, st>>>(b.part, S3, rb * kb, kb, 27, b.R, b.K, b.sr, b.sk, dw, total3);
            LAUNCH_CHECK();
            return 0;
        }
    }
    if (hf) rc = KS == 8 ? wg_dispatch<f16_t, 8, 4, 10, true>(a, grid, lds, st) : wg_dispatch<f16_t, 2, 1, 12, true>(a, grid, lds, st, rbk);
    else if (bf) rc = KS == 8 ? wg_dispatch<bf16_t, 8, 4, 10, true>(a, grid, lds, st) : wg_dispatch<bf16_t, 2, 1, 12, true>(a, grid, lds, st, rbk);
    else rc = KS == 8 ? wg_dispatch<float, 8, 8, 20, false>(a, grid, lds, st) : wg_dispatch<float, 2, 2, 24, false>(a, grid, lds, st);
    if (rc) return rc;
    const int64_t total = (int64_t)rb * kb * a.ntap * 1024;
    k_wgrad_reduce<<<dim3((unsigned)ceil_div64(total, 256), ceil_div(slices, 32)), 256, 0, st>>>(a.part, slices, rb * kb, kb, a.ntap, a.R, a.K, a.sr, a.sk, dw, total);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ ragged batches (NndetItems)
// Weight (+ bias) gradient of a 3x3x3 / stride 1 / pad 1 convolution over items of different spatial size in ONE k_wgrad3 launch:
// the persistent workgroups walk the tiles of all items, so dW comes out summed over the items (the pyramid levels that share the
// detection-head parameters) and goes through the same two-stage reduction as the uniform entry point.
int wgrad_items_run(const NndetConv* c, const NndetItems* it, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                    size_t ws_bytes, hipStream_t st) {
    int rc = items_check(c, it);
    if (rc) return rc;
    const bool bf = nndet_is16(c->dtype), hf = c->dtype == NNDET_F16;
    const int esz = bf ? 2 : 4;
    const int RB = 32 * esz;
    WgArgs b;
    memset(&b, 0, sizeof(b));
    b.N = it->n_items; b.dw = dw;
    b.p = dy; b.q = x; b.Cp = c->cout_p; b.Cq = c->cin_p; b.R = c->cout; b.K = c->cin;
    b.sr = (int64_t)c->cin * 27; b.sk = 27;
    b.ntap = 27;
    int nt = 0;
    for (int td = 0; td < 3; ++td) for (int th = 0; th < 3; ++th) for (int tw = 0; tw < 3; ++tw) {
        WgTap& t = b.taps[nt++];
        t.d[0] = td; t.d[1] = th; t.d[2] = tw; t.wt = (td * 3 + th) * 3 + tw;
    }
    b.TD = 4; b.TH = 8; b.lTH = 3;
    for (int i = 0; i < 3; ++i) { b.step[i] = 1; b.qbase[i] = -1; }
    b.H[0] = 6; b.H[1] = 10; b.H[2] = 10;
    b.dbias = dbias;
    WgItems wi;
    memset(&wi, 0, sizeof(wi));
    wi.n = it->n_items;
    int total = 0;
    for (int i = 0; i < it->n_items; ++i) {
        for (int a = 0; a < 3; ++a) wi.dims[i][a] = it->dims[i][a];
        wi.row_off[i] = it->row_off[i];
        wi.tile_begin[i] = total;
        total += ceil_div(it->dims[i][0], 4) * ceil_div(it->dims[i][1], 8) * ceil_div(it->dims[i][2], 8);
    }
    b.total_tiles = total;
    // (PL / QD / nt of the uniform kernel are unused in ITEMS mode; keep them at the first item's values for the dead code)
    for (int i = 0; i < 3; ++i) { b.PL[i] = b.QD[i] = it->dims[0][i]; }
    b.nt[0] = ceil_div(b.PL[0], 4); b.nt[1] = ceil_div(b.PL[1], 8); b.nt[2] = ceil_div(b.PL[2], 8);
    const int rb = b.Cp / 32, kb = b.Cq / 32;
    const int S3 = wgrad_slices(rb * kb, b.total_tiles);
    const size_t need3 = (size_t)S3 * rb * kb * 27 * 1024 * sizeof(float);
    if (!ws || ws_bytes < need3) return NNDET_EWORKSPACE;
    b.part = reinterpret_cast<float*>(ws);
    const size_t lds3 = (size_t)32 * (8 * RB + RB / 2) + (size_t)60 * (10 * RB + RB / 2);
    dim3 g3(S3, rb, kb);
    static bool at = false;
    if (!at) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<bf16_t, 2, false, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<f16_t, 2, false, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<float, 1, false, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        at = true;
    }
    if (hf) k_wgrad3<f16_t, 2, false, 1, true><<<g3, 256, lds3, st>>>(b, wi);
    else if (bf) k_wgrad3<bf16_t, 2, false, 1, true><<<g3, 256, lds3, st>>>(b, wi);
    else k_wgrad3<float, 1, false, 2, true><<<g3, 256, lds3, st>>>(b, wi);
    LAUNCH_CHECK();
    const int64_t total3 = (int64_t)rb * kb * 27 * 1024;
    k_wgrad_reduce<<<dim3((unsigned)ceil_div64(total3, 256), ceil_div(S3, 32)), 256, 0, st>>>(b.part, S3, rb * kb, kb, 27, b.R, b.K, b.sr, b.sk, dw, total3);
    LAUNCH_CHECK();
    return 0;
}
