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
#include <atomic>


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
    uint32_t m_tpn, m_nt2, m_nt12;   // k_wgrad3d: / (tiles per image), / nt[2], / (nt[1] * nt[2])
    int32_t lTH;          // log2(TH)
    int32_t dbg;          // timing experiments only (NNDET_WGRAD3_DBG; wrong results): 1 = stage the first tile only, 2 = no MFMA phase, 4 = no commit (k_wgrad3), 16 = no MFMA phase in the second point half (k_wgrad3d)
    WgTap taps[27];
};

// Ragged batch (NndetItems) for k_wgrad3<..., ITEMS = true>: the persistent tile walk runs over the tiles of ALL items
// (tile_begin = running tile count), so the weight gradient of parameters shared by the items is summed inside the kernel
struct WgItems {
    int32_t n, pad_;
    int32_t dims[NNDET_MAX_ITEMS][3];
    int32_t tile_begin[NNDET_MAX_ITEMS];
    int64_t row_off[NNDET_MAX_ITEMS];
    uint32_t m_nt2[NNDET_MAX_ITEMS], m_nt12[NNDET_MAX_ITEMS];   // k_wgrad3d: magic multipliers for / nt2, / (nt1 * nt2) of the item's (4, 8, 8) tiling
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
//
// M32 (round 4, 16-bit types): the same tiles and tap split on v_mfma_f32_32x32x16 instead of four 16x16x32 per (step, tap). The SQ
// counters of the 16x16x32 form (profiles/round4_wgrad3_pmc.txt): MFMA pipe busy 41 % of the kernel, 44 % of the wave cycles are
// issue stalls, 4 instructions issued per MFMA -- with 16-cycle MFMAs a wave has 4 issue slots per MFMA and spends them on the two
// transpose reads, their waits and the pipeline's bookkeeping. One 32x32x16 does the work of two of them from the SAME 2 + 2 operand
// registers (a contraction step is 16 points = two lattice rows; lane groups 0 / 1 fetch channel blocks 0 / 1 of the first row,
// 2 / 3 of the second: 256 contiguous bytes per half wave, conflict-free without any padding rule), so the instruction count per
// FLOP halves and the accumulators stay 7 x 16 registers. (Halving the LDS reads instead -- one 12-point read per halo row feeding
// the three W taps through v_alignbit -- was measured first: +2 %, the reads were never the bound; 256 B / clock, not 128.)
template <typename T, int MINW, bool AFF = false, int QDEPTH = 2, bool ITEMS = false, bool M32 = false>
__global__ __launch_bounds__(256, MINW) void k_wgrad3(const WgArgs A, const WgItems IT) {
    static_assert(!M32 || sizeof(T) == 2, "the 32x32x16 variant is for the 16-bit storage types");
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

    typename WF<T>::acc_t acc[M32 ? 1 : NTS][2][2];
#pragma unroll
    for (int t = 0; t < (M32 ? 1 : NTS); ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = WF<T>::zero();
    f32x16_t acc32[M32 ? NTS : 1];
#pragma unroll
    for (int t = 0; t < (M32 ? NTS : 1); ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[t][r] = 0.f;
    // M32 fragment bases: lane group q = (row q >> 1 of the step's two lattice rows, channel block q & 1)
    const int lane32 = (q & 1) * 16 * (int)sizeof(T) + (li >> 2) * RB + (li & 3) * 8;
    const char* const p_lane32 = sp + (q >> 1) * PROW + lane32;
    const int q_lane32 = PBYTES + (q >> 1) * QROW + lane32;

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
    auto compute32 = [&]() {
        if constexpr (M32) {
            constexpr int KS2 = 16, U = KS2 * NTS, QD_ = 3;       // 16 steps of 16 points; prefetch distance in MFMAs
            auto tr2 = [&](const char* b0) -> u32x4 {
                const uint2 a = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0)));
                const uint2 b = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0 + 4 * RB)));
                return u32x4{a.x, a.y, b.x, b.y};
            };
            u32x4 pf[2], qf[QD_ + 1];
            auto load_q = [&](int u) -> u32x4 {
                const int ks = u / NTS, ts = u % NTS;             // lattice rows 2 ks, 2 ks + 1: slice ks >> 2, rows (ks & 3) * 2 + (q >> 1)
                if (ts == NTS - 1 && do_bias) return u32x4{H16<T>::ONE2, H16<T>::ONE2, H16<T>::ONE2, H16<T>::ONE2};
                return tr2(smem + (q_lane32 + tapoff[ts]) + (((ks >> 2) * HH) + (ks & 3) * 2) * QROW);
            };
            pf[0] = tr2(p_lane32);
#pragma unroll
            for (int u0 = 0; u0 < QD_; ++u0) qf[u0] = load_q(u0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ks = u / NTS, ts = u % NTS;
                if (u + QD_ < U) qf[(u + QD_) % (QD_ + 1)] = load_q(u + QD_);
                if (ts == 0 && ks + 1 < KS2) pf[(ks + 1) & 1] = tr2(p_lane32 + (ks + 1) * 2 * PROW);
                __builtin_amdgcn_sched_barrier(0);
                acc32[ts] = H16<T>::mma32(pf[ks & 1], qf[u % (QD_ + 1)], acc32[ts]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto compute = [&]() {
        if constexpr (M32) { compute32(); return; }
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
        if (has_next && !(A.dbg & 1)) issue(tile_of(nb));           // in flight during the MFMA phase
        if (!(A.dbg & 2)) compute();
        __syncthreads();                            // every wave is done reading the tile
        if (has_next && !(A.dbg & 5)) commit();
        __syncthreads();
    }
    float* part = A.part + ((int64_t)blockIdx.x * (gridDim.y * gridDim.z) + blockIdx.y * gridDim.z + blockIdx.z) * ((int64_t)27 * 1024);
    if constexpr (M32) {
        const int n = lane & 31, mh = (lane >> 5) * 4;
        if (do_bias && n == 0) {             // column 0 of the ones-GEMM
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = r0 + (r >> 2) * 8 + mh + (r & 3);
                if (m < A.R) atomicAdd(A.dbias + m, acc32[NTS - 1][r]);
            }
        }
#pragma unroll
        for (int ts = 0; ts < NTS; ++ts) {
            if (tapw[ts] >= 0) {
                float* pt = part + (int64_t)tapw[ts] * 1024;
#pragma unroll
                for (int r = 0; r < 16; ++r) pt[((r >> 2) * 8 + mh + (r & 3)) * 32 + n] = acc32[ts][r];
            }
        }
        return;
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

// ------------------------------------------------------------------------------------------------ 3x3x3, stride 1: LDS-DMA form
// k_wgrad3d (round 4, 16-bit types): k_wgrad3's tiles and tap split with the staging taken off the critical path.
// Timing k_wgrad3 with phases switched off (NNDET_WGRAD3_DBG, profiles/round4_wgrad3_phases.txt): MFMA phase alone 0.45 ms, staging
// alone 0.30 ms, together 0.65 ms for the full-resolution 32 -> 32 layer -- the two add up although two workgroups share a CU:
// every tile pays ~250 instructions of address arithmetic, 14 buffer loads into 56 staging registers, a commit pass of 14
// ds_write_b128 and two barriers. Here
//   * the NEXT tile goes global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, as in k_ig3r) into a second buffer while the MFMAs
//     of the current one run: no staging registers, no commit pass, ONE barrier per tile; 54 pieces of 1 KB per tile, 6-7 per wave,
//     one issued every 8th MFMA. Lanes outside the tensor use an out-of-range offset and the hardware writes the zero padding;
//   * 8 waves: wave = (tap class wv & 3 as before, point half wv >> 2): 7 tap slots x 8 steps of 16 points = 56 v_mfma_32x32x16
//     per wave and tile, 112 accumulator registers, two waves per SIMD at 256 registers. Each half writes its own partial slice;
//   * LDS rows are unpadded (the destination of a DMA piece is lane-linear): the half wave of a transpose read fetches channel
//     blocks 0 / 1 of four consecutive voxels = 256 contiguous bytes, conflict-free at any row pitch.
// 2 x (16 KB dY tile + 38 KB X halo) = 108 KB of LDS: one workgroup per CU. No deferred input norm (the data never passes
// through registers): those launches stay on k_wgrad3.
__device__ __forceinline__ void wg_dma16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, uint32_t lds_dst) {
    // M0 is written in the same statement that reads it; hipcc does not count this load: waited for by hand (vmcnt(0) before the barrier)
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_uniform_rsrc(const char* p, int num_records) {
    // the descriptor must sit in SGPRs: pin the (workgroup-uniform) address there whatever the divergence analysis concluded
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(((uint64_t)hi << 32) | lo), 0, num_records, 0x00020000);
}

#ifndef WG3D_GAP
#define WG3D_GAP 1
#endif
#ifndef WG3D_QD
#define WG3D_QD 3      // LDS fragment prefetch distance in MFMAs
#endif
template <typename T, bool ITEMS>
__global__ __launch_bounds__(512, 2) void k_wgrad3d(const WgArgs A, const WgItems IT) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    constexpr int RB = 64, NTS = 7, TD = 4, TH = 8, HD = TD + 2, HH = TH + 2, HW = 10;
    constexpr int PROW = 8 * RB, QROW = HW * RB;                   // unpadded rows
    constexpr int PBYTES = 32 * PROW, QPIECES = (HD * HH * HW * 4 + 63) / 64, BUF = PBYTES + QPIECES * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int tq = wv & 3, half = wv >> 2;
    const int r0 = blockIdx.y * 32, k0 = blockIdx.z * 32;

    int tapoff[NTS], tapw[NTS];
#pragma unroll
    for (int ts = 0; ts < NTS; ++ts) {
        const int t = tq + ts * 4;
        const bool valid = t < 27;
        const int tt = valid ? t : 0;
        const int a = tt / 9, b = (tt / 3) % 3, c = tt % 3;
        tapoff[ts] = __builtin_amdgcn_readfirstlane((a * HH + b) * QROW + c * RB);
        tapw[ts] = __builtin_amdgcn_readfirstlane(valid ? tt : -1);
    }
    f32x16_t acc[NTS];
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // fragment bases: lane group q = (row q >> 1 of the step's two lattice rows, channel block q & 1)
    const int lane32 = (q & 1) * 32 + (li >> 2) * RB + (li & 3) * 8;
    const int p_lane = (q >> 1) * PROW + lane32;
    const int q_lane = PBYTES + (q >> 1) * QROW + lane32;

    // ---- DMA geometry. The waves of the FIRST point half issue all 54 pieces (the SIMD favours its older wave: they finish their MFMAs
    // ~1500 cycles before their SIMD partners and would idle at the barrier; measured with s_memtime, profiles/round4_wgrad3d_phases.txt).
    // P piece pp = tq * 4 + i: voxels pp * 16 + (lane >> 2) = slice pp >> 2 = tq, rows i * 2 + (lane >> 5), column (lane >> 2) & 7.
    // Q piece qp = tq + 4 j: granule G = qp * 64 + lane = halo voxel G >> 2 (row-major over 6 x 10 x 10), part G & 3.
    constexpr int NPP = 4, NQ = (QPIECES + 3) / 4, NPC = NPP + NQ;
    const int p_pw = (lane >> 2) & 7, p_hb = lane >> 5, part = lane & 3;
    uint32_t qsel[NQ];          // bit hd | bit 6 + hh | bit 16 + hw; bit 31 = no such granule
    int qcoord[NQ];             // hd << 16 | hh << 8 | hw
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int G = (tq + 4 * j) * 64 + lane;
        const int vox = G >> 2;
        const int row = vox / HW, hw = vox - row * HW;
        const int hd = row / HH, hh = row - hd * HH;
        const bool valid = vox < HD * HH * HW;
        qsel[j] = valid ? (1u << hd) | (1u << (6 + hh)) | (1u << (16 + hw)) : 0x80000000u;
        qcoord[j] = (hd << 16) | (hh << 8) | hw;
    }
    // per-item strides (bytes) and the lane parts of the offsets that depend on them
    int PL0 = A.PL[0], PL1 = A.PL[1], PL2 = A.PL[2];
    int p_rowb = 0, p_slab = 0, q_rowb = 0, q_slab = 0, p_voff = 0, n_cur = -1;
    int qrel[NQ];
    auto set_dims = [&](int d0, int d1, int d2) {
        PL0 = d0; PL1 = d1; PL2 = d2;
        p_rowb = d2 * A.Cp * 2; p_slab = d1 * p_rowb;
        q_rowb = d2 * A.Cq * 2; q_slab = d1 * q_rowb;
        p_voff = p_hb * p_rowb + p_pw * A.Cp * 2 + part * 16;
#pragma unroll
        for (int j = 0; j < NQ; ++j)
            qrel[j] = (qcoord[j] >> 16) * q_slab + ((qcoord[j] >> 8) & 255) * q_rowb + (qcoord[j] & 255) * A.Cq * 2 + part * 16;
    };
    if constexpr (!ITEMS) set_dims(A.PL[0], A.PL[1], A.PL[2]);

    // scalars of the tile being staged (no divisions: magic multipliers from the host)
    __amdgpu_buffer_rsrc_t prs, qrs;
    int l0d = 0, l0h = 0, l0w = 0;
    uint32_t qmask = 0;
    bool p_okw = false;
    auto mdivu = [](uint32_t n, uint32_t m) -> uint32_t { return m ? __umulhi(n, m) : n; };
    auto decode = [&](int tile) {
        int n, tt, nt1 = A.nt[1], nt2 = A.nt[2];
        uint32_t m2 = A.m_nt2, m12 = A.m_nt12;
        int64_t p_base, q_base;
        if constexpr (ITEMS) {
            n = 0;
            while (n + 1 < IT.n && tile >= IT.tile_begin[n + 1]) ++n;
            tt = tile - IT.tile_begin[n];
            if (n != n_cur) { set_dims(IT.dims[n][0], IT.dims[n][1], IT.dims[n][2]); n_cur = n; }
            nt1 = (PL1 + TH - 1) / TH; nt2 = (PL2 + 7) / 8;
            m2 = IT.m_nt2[n]; m12 = IT.m_nt12[n];
            p_base = IT.row_off[n] * A.Cp * (int64_t)2; q_base = IT.row_off[n] * A.Cq * (int64_t)2;
        } else {
            n = (int)mdivu((uint32_t)tile, A.m_tpn);
            tt = tile - n * (A.nt[0] * nt1 * nt2);
            p_base = (int64_t)n * PL0 * p_slab; q_base = (int64_t)n * PL0 * q_slab;
        }
        const int td_i = (int)mdivu((uint32_t)tt, m12);
        tt -= td_i * nt1 * nt2;
        const int th_i = (int)mdivu((uint32_t)tt, m2);
        const int tw_i = tt - th_i * nt2;
        l0d = td_i * TD; l0h = th_i * TH; l0w = tw_i * 8;
        // descriptors based at the tile origin (dY) / the halo origin (X; may lie before the image: only lanes inside the tensor use it)
        const int64_t p_org = (int64_t)l0d * p_slab + l0h * p_rowb + l0w * A.Cp * 2 + r0 * 2;
        const int64_t q_org = (int64_t)(l0d - 1) * q_slab + (l0h - 1) * q_rowb + (l0w - 1) * A.Cq * 2 + k0 * 2;
        prs = wg_uniform_rsrc(reinterpret_cast<const char*>(A.p) + p_base + p_org, 0x7ffffff0);
        qrs = wg_uniform_rsrc(reinterpret_cast<const char*>(A.q) + q_base + q_org, 0x7ffffff0);
        p_okw = l0w + p_pw < PL2;
        auto rng = [](int lo, int hi) -> uint32_t { return ((1u << hi) - 1u) & ~((1u << lo) - 1u); };   // bits [lo, hi)
        const uint32_t md = rng(l0d == 0 ? 1 : 0, min(HD, PL0 - l0d + 1));
        const uint32_t mh = rng(l0h == 0 ? 1 : 0, min(HH, PL1 - l0h + 1));
        const uint32_t mw = rng(l0w == 0 ? 1 : 0, min(HW, PL2 - l0w + 1));
        qmask = md | (mh << 6) | (mw << 16);
    };
    // piece i of this wave (0..3: dY; 4..: X halo) of the decoded tile -> buffer `buf`
    auto dma_piece = [&](int i, int buf) {
        if (i < NPP) {
            const int pp = tq * NPP + i;
            const int pd = tq, ph0 = i * 2;
            const bool ok = p_okw && (l0d + pd < PL0) && (l0h + ph0 + p_hb < PL1);
            wg_dma16(prs, ok ? p_voff : (int)0x80000000, __builtin_amdgcn_readfirstlane(pd * p_slab + ph0 * p_rowb),
                     (uint32_t)__builtin_amdgcn_readfirstlane(buf * BUF + pp * 1024));
        } else {
            const int j = i - NPP;
            if (tq + 4 * j < QPIECES) {
                const bool ok = (qsel[j] & qmask) == qsel[j];
                wg_dma16(qrs, ok ? qrel[j] : (int)0x80000000, 0, (uint32_t)__builtin_amdgcn_readfirstlane(buf * BUF + PBYTES + (tq + 4 * j) * 1024));
            }
        }
    };
    const bool do_bias = A.dbias != nullptr && blockIdx.z == 0 && tq == 3;     // the 28th tap slot of both point halves
    auto tr2 = [&](const char* b0) -> u32x4 {
        const uint2 a = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0)));
        const uint2 b = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0 + 4 * RB)));
        return u32x4{a.x, a.y, b.x, b.y};
    };
    // MFMA phase on buffer `buf`; with `stage` the pieces of the decoded (next) tile are issued into the other buffer, one per 8 MFMAs
    auto compute = [&](int buf, bool stage) {
        // the pieces of the next tile are issued EARLY in the phase (one per DMA_GAP MFMAs): they must have landed at its end, and a
        // piece issued with 4 MFMAs to go exposes its whole HBM latency at the barrier (measured: one per 8 MFMAs 0.642 ms, ...)
        constexpr int KH = 8, U = KH * NTS, QD_ = WG3D_QD, DMA_GAP = WG3D_GAP;
        // step ks of this half = lattice rows 2 ks, 2 ks + 1 of slices 2 half, 2 half + 1: slice ks >> 2, rows (ks & 3) * 2 + (q >> 1)
        const char* const pb = smem + buf * BUF + p_lane + half * KH * 2 * PROW;
        const char* const qb = smem + buf * BUF + q_lane + half * 2 * HH * QROW;
        u32x4 pf[2], qf[QD_ + 1];
        auto load_q = [&](int u) -> u32x4 {
            const int ks = u / NTS, ts = u % NTS;
            if (ts == NTS - 1 && do_bias) return u32x4{H16<T>::ONE2, H16<T>::ONE2, H16<T>::ONE2, H16<T>::ONE2};
            return tr2(qb + tapoff[ts] + (((ks >> 2) * HH) + (ks & 3) * 2) * QROW);
        };
        pf[0] = tr2(pb);
#pragma unroll
        for (int u0 = 0; u0 < QD_; ++u0) qf[u0] = load_q(u0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = u / NTS, ts = u % NTS;
            if (u + QD_ < U) qf[(u + QD_) % (QD_ + 1)] = load_q(u + QD_);
            if (ts == 0 && ks + 1 < KH) pf[(ks + 1) & 1] = tr2(pb + (ks + 1) * 2 * PROW);
            if (u < DMA_GAP * NPC && u % DMA_GAP == 0 && stage) dma_piece(u / DMA_GAP, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[ts] = H16<T>::mma32(pf[ks & 1], qf[u % (QD_ + 1)], acc[ts]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int G = gridDim.x, bx = blockIdx.x;
    // round k covers tiles [k G, (k + 1) G); within a round the workgroups of one XCD take a contiguous run (xcd_compact, written out:
    // bx < G, so no division)
    auto tile_of = [&](int b0) {
        const int cnt = min(G, A.total_tiles - b0);
        return b0 + ((cnt & 7) ? bx : (bx & 7) * (cnt >> 3) + (bx >> 3));
    };
    const bool stager = half == 0;
    if (bx < A.total_tiles && stager) {
        decode(tile_of(0));
#pragma unroll
        for (int i = 0; i < NPC; ++i) dma_piece(i, 0);
        if (G + bx < A.total_tiles) decode(tile_of(G));       // the tile staged during the first MFMA phase
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
#ifdef WG3D_TIMING
    long long t_comp = 0, t_dec = 0, t_bar = 0; int t_n = 0;
#endif
    for (int base = 0; base + bx < A.total_tiles; base += G) {
        const bool has_next = base + G + bx < A.total_tiles;
#ifdef WG3D_TIMING
        const long long t0 = __builtin_readcyclecounter();
#endif
        if (A.dbg & 2) {
            if (has_next && stager && !(A.dbg & 1))
#pragma unroll
                for (int i = 0; i < NPC; ++i) dma_piece(i, buf ^ 1);
        } else if (!((A.dbg & 16) && half)) compute(buf, has_next && stager && !(A.dbg & 1));
#ifdef WG3D_TIMING
        const long long t1 = __builtin_readcyclecounter();
#endif
        // the scalars of the tile after the next one, computed BEFORE the barrier by the waves that staged (their pieces of the next tile
        // are on their way; their SIMD partners are still in their MFMA phase). ~100 dependent scalar instructions = 780 cycles; placing
        // them in the middle of the MFMA phase instead was a draw for uniform launches and 50 % slower for the ragged ones.
        if (stager && base + 2 * G + bx < A.total_tiles) decode(tile_of(base + 2 * G));
#ifdef WG3D_TIMING
        const long long t2 = __builtin_readcyclecounter();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                            // every wave is done reading `buf`, the next tile has landed in the other buffer
        buf ^= 1;
#ifdef WG3D_TIMING
        const long long t3 = __builtin_readcyclecounter();
        t_comp += t1 - t0; t_dec += t2 - t1; t_bar += t3 - t2; ++t_n;
#endif
    }
#ifdef WG3D_TIMING
    if ((bx == 0 || bx == 100) && lane == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        printf("wg %d wave %d tiles %d: compute %lld decode %lld wait+barrier %lld cycles per tile\n", bx, wv, t_n, t_comp / t_n, t_dec / t_n, t_bar / t_n);
#endif
    // the second point half hands its sums to the first through LDS (both buffers are free now; two rounds of 4 + 3 tap slots, 64 KB):
    // ONE partial slice per workgroup -- half the partial traffic and half the work of k_wgrad_reduce
    float* const xch = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        if (half) {
#pragma unroll
            for (int ts = rd * 4; ts < (rd ? NTS : 4); ++ts)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[((tq * 4 + (ts - rd * 4)) * 16 + r) * 64 + lane] = acc[ts][r];
        }
        __syncthreads();
        if (!half) {
#pragma unroll
            for (int ts = rd * 4; ts < (rd ? NTS : 4); ++ts)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ts][r] += xch[((tq * 4 + (ts - rd * 4)) * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (half) return;
    const int n = lane & 31, mh = (lane >> 5) * 4;
    if (do_bias && n == 0) {                        // column 0 of the ones-GEMM
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = r0 + (r >> 2) * 8 + mh + (r & 3);
            if (m < A.R) atomicAdd(A.dbias + m, acc[NTS - 1][r]);
        }
    }
    float* part_out = A.part + ((int64_t)blockIdx.x * (gridDim.y * gridDim.z) + blockIdx.y * gridDim.z + blockIdx.z) * ((int64_t)27 * 1024);
#pragma unroll
    for (int ts = 0; ts < NTS; ++ts) {
        if (tapw[ts] >= 0) {
            float* pt = part_out + (int64_t)tapw[ts] * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) pt[((r >> 2) * 8 + mh + (r & 3)) * 32 + n] = acc[ts][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3x3, stride 1: LDS-DMA form, lean
// k_wgrad3e (round 6): k_wgrad3d's tiles, LDS layout, DMA geometry and epilogue with the instruction stream between two MFMAs cut
// down to what the matrix pipe can hide. The ISA of k_wgrad3d spent, per MFMA and wave, a v_readlane (the tap offsets lived in
// spilled SGPRs) + s_nop + v_add for every fragment address, a branch around every DMA slot, and ~250 scalar instructions of tile
// decode per tile with 61 SGPR spills: 9+ issue slots per 32-cycle MFMA -- a lone wave ran the phase at 47 cycles per MFMA and the
// younger wave of each SIMD finished 1500 cycles after its partner (profiles/round4_wgrad_investigation.txt). Here
//   * the two roles are two straight-line code paths chosen ONCE per wave: the waves of the first point half stage (DMA + tile
//     walk), the others only compute -- no per-slot branches;
//   * every fragment address is a VGPR base (one per tap slot, re-based by the buffer offset once per tile) + an immediate: per
//     MFMA a wave issues 2 transpose reads, 2/7 of a dY read, one s_waitcnt and the MFMA;
//   * the tile walk is incremental (mixed-radix add of the grid size as in k_wgrad3s; ragged batches: flat index + item search);
//     buffer descriptors are per IMAGE / ITEM, the tile origin travels in the scalar offset of the DMA instruction, validity is
//     two bit masks per tile (dY: column | row, X halo: slab | row | column).
struct Wg3eWalk { int32_t gw, gh, gd, gn; };      // grid size = gw + nt2 * (gh + nt1 * (gd + nt0 * gn))

typedef __attribute__((address_space(3))) char* lds_char_ptr;
__device__ __forceinline__ u32x4 wg_tr2(uint32_t a) {      // 16 points x 32 channels fragment: two transpose reads 4 voxels apart
    const uint2 x = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds_char_ptr)(uintptr_t)a));
    const uint2 y = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds_char_ptr)(uintptr_t)(a + 256)));
    return u32x4{x.x, x.y, y.x, y.y};
}

template <typename T, bool ITEMS>
__global__ __launch_bounds__(512, 2) void k_wgrad3e(const WgArgs A, const WgItems IT, const Wg3eWalk WK) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    constexpr int RB = 64, NTS = 7, TD = 4, TH = 8, HD = TD + 2, HH = TH + 2, HW = 10;
    constexpr int PROW = 8 * RB, QROW = HW * RB;
    constexpr int PBYTES = 32 * PROW, QPIECES = (HD * HH * HW * 4 + 63) / 64, BUF = PBYTES + QPIECES * 1024;
    constexpr int NPP = 4, NQ = (QPIECES + 3) / 4, NPC = NPP + NQ;
    constexpr int KH = 8, U = KH * NTS, QD_ = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int tq = wv & 3, half = wv >> 2;
    const int r0 = blockIdx.y * 32, k0 = blockIdx.z * 32;
    const int G = gridDim.x, bx = blockIdx.x;

    f32x16_t acc[NTS];
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- fragment bases in buffer 0 (LDS byte addresses; lane group q = (row q >> 1 of the step's two lattice rows, channel block q & 1))
    const uint32_t sb = (uint32_t)(uintptr_t)(lds_char_ptr)smem;
    const int lane32 = (q & 1) * 32 + (li >> 2) * RB + (li & 3) * 8;
    const uint32_t pA0 = sb + (q >> 1) * PROW + lane32 + half * KH * 2 * PROW;
    uint32_t qA0[NTS];
#pragma unroll
    for (int ts = 0; ts < NTS; ++ts) {
        const int t = tq + ts * 4;
        const int tt = t < 27 ? t : 0;
        const int a = tt / 9, b = (tt / 3) % 3, c = tt % 3;
        qA0[ts] = sb + PBYTES + (q >> 1) * QROW + lane32 + half * 2 * HH * QROW + (a * HH + b) * QROW + c * RB;
    }
    const bool do_bias = A.dbias != nullptr && blockIdx.z == 0 && tq == 3;     // the 28th tap slot of both point halves
    const u32x4 ones = u32x4{H16<T>::ONE2, H16<T>::ONE2, H16<T>::ONE2, H16<T>::ONE2};

    // ---- this workgroup's tiles: t_k = k G + p, p = its place in the XCD-compact order of a round (a bijection on [0, G))
    const int pstart = (G & 7) ? bx : (bx & 7) * (G >> 3) + (bx >> 3);
    const int ntile = pstart < A.total_tiles ? (A.total_tiles - pstart + G - 1) / G : 0;

    // one MFMA phase on the buffer at byte offset `boff`; STG: the 14 DMA slots of the next tile ride in the first 14 MFMA gaps
    auto phase = [&](uint32_t boff, auto dma) {
        uint32_t pA = pA0 + boff, qA[NTS];
#pragma unroll
        for (int ts = 0; ts < NTS; ++ts) qA[ts] = qA0[ts] + boff;
        asm volatile("" : "+v"(pA));                                   // (bases live in VGPRs: everything below is base + immediate)
#pragma unroll
        for (int ts = 0; ts < NTS; ++ts) asm volatile("" : "+v"(qA[ts]));
        u32x4 pf[2], qf[QD_ + 1];
        auto load_q = [&](int u) -> u32x4 {
            const int ks = u / NTS, ts = u % NTS;
            // step ks of this half = lattice rows 2 ks, 2 ks + 1 of slices 2 half, 2 half + 1: slice ks >> 2, rows (ks & 3) * 2 + (q >> 1)
            return wg_tr2(qA[ts] + (((ks >> 2) * HH) + (ks & 3) * 2) * QROW);
        };
        pf[0] = wg_tr2(pA);
#pragma unroll
        for (int u0 = 0; u0 < QD_; ++u0) qf[u0] = load_q(u0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = u / NTS, ts = u % NTS;
            if (u + QD_ < U) qf[(u + QD_) % (QD_ + 1)] = load_q(u + QD_);
            if (ts == 0 && ks + 1 < KH) pf[(ks + 1) & 1] = wg_tr2(pA + (ks + 1) * 2 * PROW);
            if (u < NPC) dma(u);
            __builtin_amdgcn_sched_barrier(0);
            // (the ones of the bias slot are selected at USE time: selecting at load time waits for the read that was just issued)
            const u32x4 bq = (ts == NTS - 1 && do_bias) ? ones : qf[u % (QD_ + 1)];
            acc[ts] = H16<T>::mma32(pf[ks & 1], bq, acc[ts]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (half == 0) {
        // ================================================================ staging waves: DMA + tile walk + their share of the MFMAs
        // P piece pp = tq * 4 + i: voxels pp * 16 + (lane >> 2) = slice tq, rows i * 2 + (lane >> 5), column (lane >> 2) & 7.
        // Q piece qp = tq + 4 j: granule Gq = qp * 64 + lane = halo voxel Gq >> 2 (row-major over 6 x 10 x 10), part Gq & 3.
        const int p_pw = (lane >> 2) & 7, p_hb = lane >> 5, part = lane & 3;
        uint32_t psel[NPP];          // bit column | bit 8 + row
        uint32_t qsel[NQ];           // bit hd | bit 6 + hh | bit 16 + hw; bit 31 = no such granule
        int qcoord[NQ];
#pragma unroll
        for (int i = 0; i < NPP; ++i) psel[i] = (1u << p_pw) | (1u << (8 + i * 2 + p_hb));
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int Gq = (tq + 4 * j) * 64 + lane;
            const int vox = Gq >> 2;
            const int row = vox / HW, hw = vox - row * HW;
            const int hd = row / HH, hh = row - hd * HH;
            qsel[j] = vox < HD * HH * HW ? (1u << hd) | (1u << (6 + hh)) | (1u << (16 + hw)) : 0x80000000u;
            qcoord[j] = (hd << 16) | (hh << 8) | hw;
        }
        // geometry of the current image / item (bytes)
        int PL0 = A.PL[0], PL1 = A.PL[1], PL2 = A.PL[2];
        int p_rowb = 0, p_slab = 0, q_rowb = 0, q_slab = 0, p_voff = 0, sp_tq = 0, p_row2 = 0;
        int qrel[NQ];
        auto set_dims = [&](int d0, int d1, int d2) {
            PL0 = d0; PL1 = d1; PL2 = d2;
            p_rowb = d2 * A.Cp * 2; p_slab = d1 * p_rowb;
            q_rowb = d2 * A.Cq * 2; q_slab = d1 * q_rowb;
            p_voff = p_hb * p_rowb + p_pw * A.Cp * 2 + part * 16;
            sp_tq = tq * p_slab; p_row2 = 2 * p_rowb;
#pragma unroll
            for (int j = 0; j < NQ; ++j)
                qrel[j] = (qcoord[j] >> 16) * q_slab + ((qcoord[j] >> 8) & 255) * q_rowb + (qcoord[j] & 255) * A.Cq * 2 + part * 16;
        };
        if constexpr (!ITEMS) set_dims(A.PL[0], A.PL[1], A.PL[2]);
        // walk state: image / item, tile coordinates (uniform), flat tile index (ragged)
        int c_n = 0, c_d = 0, c_h = 0, c_w = 0, flat = pstart, n_dims = -1;
        int nt1 = A.nt[1], nt2 = A.nt[2];
        // per tile: descriptors (base = image / item origin of this channel block; X: moved back by one halo voxel in every axis, only
        // lanes inside the tensor use it), scalar offsets of the tile origin, validity masks
        __amdgpu_buffer_rsrc_t prs, qrs;
        int soff_p = 0, soff_q = 0;
        uint32_t pmask = 0, qmask = 0;
        auto setup = [&](bool live) {
            int64_t p_img, q_img;
            if constexpr (ITEMS) {
                if (c_n != n_dims) { set_dims(IT.dims[c_n][0], IT.dims[c_n][1], IT.dims[c_n][2]); n_dims = c_n; }
                p_img = IT.row_off[c_n] * A.Cp * (int64_t)2; q_img = IT.row_off[c_n] * A.Cq * (int64_t)2;
            } else {
                p_img = (int64_t)c_n * PL0 * p_slab; q_img = (int64_t)c_n * PL0 * q_slab;
            }
            const int l0d = c_d * TD, l0h = c_h * TH, l0w = c_w * 8;
            prs = wg_uniform_rsrc(reinterpret_cast<const char*>(A.p) + p_img + r0 * 2, 0x7ffffff0);
            qrs = wg_uniform_rsrc(reinterpret_cast<const char*>(A.q) + q_img + k0 * 2 - (int64_t)(q_slab + q_rowb + A.Cq * 2), 0x7ffffff0);
            soff_p = l0d * p_slab + l0h * p_rowb + l0w * A.Cp * 2;
            soff_q = l0d * q_slab + l0h * q_rowb + l0w * A.Cq * 2;
            auto rng = [](int lo, int hi) -> uint32_t { return ((1u << hi) - 1u) & ~((1u << lo) - 1u); };   // bits [lo, hi)
            const uint32_t md = rng(l0d == 0 ? 1 : 0, min(HD, PL0 - l0d + 1));
            const uint32_t mh = rng(l0h == 0 ? 1 : 0, min(HH, PL1 - l0h + 1));
            const uint32_t mw = rng(l0w == 0 ? 1 : 0, min(HW, PL2 - l0w + 1));
            qmask = live ? md | (mh << 6) | (mw << 16) : 0u;
            const uint32_t pw = (1u << min(8, PL2 - l0w)) - 1u, ph = (1u << min(TH, PL1 - l0h)) - 1u;
            pmask = (live && l0d + tq < PL0) ? pw | (ph << 8) : 0u;
        };
        auto locate = [&]() {                          // state <- tile `flat` (full decode: the first tile; ragged batches: every tile)
            if constexpr (ITEMS) {
                int n = c_n;
                while (n + 1 < IT.n && flat >= IT.tile_begin[n + 1]) ++n;
                c_n = n;
                const int d1 = IT.dims[n][1], d2 = IT.dims[n][2];
                nt1 = (d1 + TH - 1) / TH; nt2 = (d2 + 7) / 8;
                int tt = flat - IT.tile_begin[n];
                const uint32_t m12 = IT.m_nt12[n], m2 = IT.m_nt2[n];
                c_d = m12 ? (int)__umulhi((uint32_t)tt, m12) : tt;
                tt -= c_d * nt1 * nt2;
                c_h = m2 ? (int)__umulhi((uint32_t)tt, m2) : tt;
                c_w = tt - c_h * nt2;
            } else {
                const int tpn = A.nt[0] * nt1 * nt2;
                c_n = flat / tpn;
                int tt = flat - c_n * tpn;
                c_w = tt % nt2; tt /= nt2;
                c_h = tt % nt1;
                c_d = tt / nt1;
            }
        };
        auto advance = [&]() {                         // tile += grid size
            flat += G;
            if constexpr (ITEMS) {
                locate();
            } else {
                c_w += WK.gw; int cy = c_w >= nt2; c_w -= cy ? nt2 : 0;
                c_h += WK.gh + cy; cy = c_h >= nt1; c_h -= cy ? nt1 : 0;
                c_d += WK.gd + cy; cy = c_d >= A.nt[0]; c_d -= cy ? A.nt[0] : 0;
                c_n += WK.gn + cy;
            }
        };
        uint32_t dst_p = 0, dst_q = 0;                 // LDS destinations of this wave's pieces in the buffer being filled
        auto dma = [&](int i) {
            if (i < NPP) {
                const bool ok = (psel[i] & pmask) == psel[i];
                wg_dma16(prs, ok ? p_voff : (int)0x80000000, __builtin_amdgcn_readfirstlane(soff_p + sp_tq + i * p_row2), dst_p + i * 1024);
            } else {
                const int j = i - NPP;
                const bool ok = (qsel[j] & qmask) == qsel[j];
                if ((j + 1) * 4 <= QPIECES) {
                    wg_dma16(qrs, ok ? qrel[j] : (int)0x80000000, soff_q, dst_q + j * 4096);
                } else if (tq + 4 * j < QPIECES) {     // the last, partial group of four pieces
                    wg_dma16(qrs, ok ? qrel[j] : (int)0x80000000, soff_q, dst_q + j * 4096);
                }
            }
        };
        auto aim = [&](uint32_t boff) {
            dst_p = (uint32_t)__builtin_amdgcn_readfirstlane(sb + boff + tq * NPP * 1024);
            dst_q = (uint32_t)__builtin_amdgcn_readfirstlane(sb + boff + PBYTES + tq * 1024);
        };
        if (ntile > 0) {
            locate(); setup(true); aim(0);
#pragma unroll
            for (int i = 0; i < NPC; ++i) dma(i);
            advance(); setup(ntile > 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t boff = 0;
#ifdef WG3E_TIMING
        long long t_ph = 0, t_walk = 0, t_bar = 0;
#endif
        for (int k = 0; k < ntile; ++k) {
#ifdef WG3E_TIMING
            const long long t0 = __builtin_readcyclecounter();
#endif
            aim(boff ^ (uint32_t)BUF);
            phase(boff, dma);                           // (past the last tile the masks are empty: the slots write zeros, read nothing)
#ifdef WG3E_TIMING
            const long long t1 = __builtin_readcyclecounter();
#endif
            advance(); setup(k + 2 < ntile);
#ifdef WG3E_TIMING
            const long long t2 = __builtin_readcyclecounter();
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                            // every wave is done reading this buffer, the next tile has landed in the other one
            boff ^= (uint32_t)BUF;
#ifdef WG3E_TIMING
            const long long t3 = __builtin_readcyclecounter();
            t_ph += t1 - t0; t_walk += t2 - t1; t_bar += t3 - t2;
#endif
        }
#ifdef WG3E_TIMING
        if ((bx == 0 || bx == 100) && lane == 0 && blockIdx.y == 0 && blockIdx.z == 0 && ntile > 0)
            printf("wg %d wave %d (stager) tiles %d: phase %lld walk %lld wait+barrier %lld cycles per tile\n", bx, wv, ntile, t_ph / ntile, t_walk / ntile, t_bar / ntile);
#endif
    } else {
        // ================================================================ computing waves
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t boff = 0;
#ifdef WG3E_TIMING
        long long t_ph = 0, t_bar = 0;
#endif
        for (int k = 0; k < ntile; ++k) {
#ifdef WG3E_TIMING
            const long long t0 = __builtin_readcyclecounter();
#endif
            phase(boff, [](int) {});
#ifdef WG3E_TIMING
            const long long t1 = __builtin_readcyclecounter();
#endif
            __syncthreads();
            boff ^= (uint32_t)BUF;
#ifdef WG3E_TIMING
            t_ph += t1 - t0; t_bar += __builtin_readcyclecounter() - t1;
#endif
        }
#ifdef WG3E_TIMING
        if ((bx == 0 || bx == 100) && lane == 0 && blockIdx.y == 0 && blockIdx.z == 0 && ntile > 0)
            printf("wg %d wave %d (compute) tiles %d: phase %lld wait+barrier %lld cycles per tile\n", bx, wv, ntile, t_ph / ntile, t_bar / ntile);
#endif
    }

    // the second point half hands its sums to the first through LDS (both buffers are free now; two rounds of 4 + 3 tap slots, 64 KB):
    // ONE partial slice per workgroup -- half the partial traffic and half the work of k_wgrad_reduce
    float* const xch = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        if (half) {
#pragma unroll
            for (int ts = rd * 4; ts < (rd ? NTS : 4); ++ts)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[((tq * 4 + (ts - rd * 4)) * 16 + r) * 64 + lane] = acc[ts][r];
        }
        __syncthreads();
        if (!half) {
#pragma unroll
            for (int ts = rd * 4; ts < (rd ? NTS : 4); ++ts)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ts][r] += xch[((tq * 4 + (ts - rd * 4)) * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (half) return;
    const int n = lane & 31, mh = (lane >> 5) * 4;
    if (do_bias && n == 0) {                        // column 0 of the ones-GEMM
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = r0 + (r >> 2) * 8 + mh + (r & 3);
            if (m < A.R) atomicAdd(A.dbias + m, acc[NTS - 1][r]);
        }
    }
    float* part_out = A.part + ((int64_t)blockIdx.x * (gridDim.y * gridDim.z) + blockIdx.y * gridDim.z + blockIdx.z) * ((int64_t)27 * 1024);
#pragma unroll
    for (int ts = 0; ts < NTS; ++ts) {
        const int tap = tq + ts * 4;
        if (tap < 27) {
            float* pt = part_out + (int64_t)tap * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) pt[((r >> 2) * 8 + mh + (r & 3)) * 32 + n] = acc[ts][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3x3, stride 2: LDS-DMA form
// k_wgrad3s (round 4, 16-bit types): the stride-(2, 2, 2) / pad 1 transitions on k_wgrad3d's recipe. The generic k_wgrad ran them at
// 0.17-0.19 PFLOP/s alone and 1.3 + 0.6 + 0.4 ms inside the step -- skipping them altogether made the step 0.98 ms shorter
// (profiles/round4_wgrad_strided.txt): one 64-point tile in flight per workgroup behind a 49 KB halo (12 staging pieces per thread,
// 2 barriers), transpose reads with a 2-voxel stride (2-way bank conflicts), 192 workgroups for 256 CUs.
//   * tile = 2 x 4 x 8 lattice points of dY (BOTH 32-channel row blocks of the workgroup's pair: the X halo is staged once for them),
//     halo = 5 x 9 x 17 voxels of one 32-channel block of X; double-buffered by LDS-DMA, one barrier per tile;
//   * the halo rows are stored DE-INTERLEAVED along W: the 9 even-offset voxels, then the 8 odd ones (the DMA destination is
//     lane-linear, the SOURCE address per lane is free). Tap c = 1 reads the odd run, c = 0 / 2 the even run at slot 0 / 1: every
//     fragment is 8 consecutive voxels again, the half wave of a transpose read covers 256 contiguous bytes -- conflict-free;
//   * 8 waves = (tap class wv & 3: 7 slots) x (row block wv >> 2); v_mfma_32x32x16, 4 steps of 16 points: 28 MFMAs per wave and tile;
//   * the tile walk is incremental (the workgroup's tile index advances by the grid size: a mixed-radix add with the host's
//     decomposition of the grid size) -- the ~100 dependent scalar instructions of k_wgrad3d's decode would be a third of a tile here.
// Waves 0..3 issue all 56 pieces of the next tile in the first 14 MFMA slots.
struct Wg3sInc { int32_t gw, gh, gd, gn; };      // grid size = gw + nt2 * (gh + nt1 * (gd + nt0 * gn))

template <typename T>
__global__ __launch_bounds__(512, 2) void k_wgrad3s(const WgArgs A, const Wg3sInc INC) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    constexpr int RB = 64, NTS = 7, TD = 2, TH = 4, HD = 2 * TD + 1, HH = 2 * TH + 1, HW = 17, NEV = 9;
    constexpr int PROW = 8 * RB, QROW = HW * RB;
    constexpr int PBLK = TD * TH * PROW, PBYTES = 2 * PBLK;                        // two row blocks of dY
    constexpr int QVOX = HD * HH * HW, QPIECES = (QVOX * 4 + 63) / 64, BUF = PBYTES + QPIECES * 1024;
    constexpr int NPP = 2, NQ = (QPIECES + 3) / 4, NPC = NPP + NQ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int tq = wv & 3, rbw = wv >> 2;
    const int r0 = blockIdx.y * 64, k0 = blockIdx.z * 32;

    int tapoff[NTS], tapw[NTS];
#pragma unroll
    for (int ts = 0; ts < NTS; ++ts) {
        const int t = tq + ts * 4;
        const bool valid = t < 27;
        const int tt = valid ? t : 0;
        const int a = tt / 9, b = (tt / 3) % 3, c = tt % 3;
        tapoff[ts] = __builtin_amdgcn_readfirstlane((a * HH + b) * QROW + (c == 1 ? NEV * RB : c == 2 ? RB : 0));
        tapw[ts] = __builtin_amdgcn_readfirstlane(valid ? tt : -1);
    }
    f32x16_t acc[NTS];
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // fragment bases: lane group q = (row q >> 1 of the step's two lattice rows, channel block q & 1); lattice rows are 2 halo rows apart
    const int lane32 = (q & 1) * 32 + (li >> 2) * RB + (li & 3) * 8;
    const int p_lane = rbw * PBLK + (q >> 1) * PROW + lane32;
    const int q_lane = PBYTES + (q >> 1) * 2 * QROW + lane32;

    // ---- DMA geometry (waves 0..3). dY piece pp = tq * 2 + i = row block pp >> 2, rows (pp & 3) * 2 + (lane >> 5) of the 8 lattice rows,
    // column (lane >> 2) & 7. X piece qp = tq + 4 j: granule G = qp * 64 + lane = halo slot G >> 2 (row-major over 5 x 9 x 17 slots,
    // slot s of a row = W offset 2 s (s < 9) or 2 (s - 9) + 1), part G & 3.
    const int p_pw = (lane >> 2) & 7, p_hb = lane >> 5, part = lane & 3;
    const int p_rowb = A.PL[2] * A.Cp * 2, p_slab = A.PL[1] * p_rowb;
    const int q_rowb = A.QD[2] * A.Cq * 2, q_slab = A.QD[1] * q_rowb;
    const int p_voff = p_hb * p_rowb + p_pw * A.Cp * 2 + part * 16;
    uint32_t qsel[NQ];          // bit hd | bit 5 + hh | bit 14 + W offset; bit 31 = no such granule
    int qrel[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int G = (tq + 4 * j) * 64 + lane;
        const int vox = G >> 2;
        const int row = vox / HW, sl = vox - row * HW;
        const int hd = row / HH, hh = row - hd * HH;
        const int jw = sl < NEV ? 2 * sl : 2 * (sl - NEV) + 1;
        const bool valid = vox < QVOX;
        qsel[j] = valid ? (1u << hd) | (1u << (5 + hh)) | (1u << (14 + jw)) : 0x80000000u;
        qrel[j] = hd * q_slab + hh * q_rowb + jw * A.Cq * 2 + part * 16;
    }

    // scalars of the tile being staged
    __amdgpu_buffer_rsrc_t prs, qrs;
    int l0d = 0, l0h = 0, l0w = 0;
    uint32_t qmask = 0;
    bool p_okw = false;
    int c_n = 0, c_d = 0, c_h = 0, c_w = 0;               // mixed-radix coordinates of the tile last decoded
    auto setup = [&]() {
        l0d = c_d * TD; l0h = c_h * TH; l0w = c_w * 8;
        const int64_t p_org = (int64_t)c_n * A.PL[0] * p_slab + (int64_t)l0d * p_slab + l0h * p_rowb + l0w * A.Cp * 2 + r0 * 2;
        const int64_t q_org = (int64_t)c_n * A.QD[0] * q_slab + (int64_t)(2 * l0d - 1) * q_slab + (2 * l0h - 1) * q_rowb + (2 * l0w - 1) * A.Cq * 2 + k0 * 2;
        prs = wg_uniform_rsrc(reinterpret_cast<const char*>(A.p) + p_org, 0x7ffffff0);
        qrs = wg_uniform_rsrc(reinterpret_cast<const char*>(A.q) + q_org, 0x7ffffff0);
        p_okw = l0w + p_pw < A.PL[2];
        auto rng = [](int lo, int hi) -> uint32_t { return (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u); };   // bits [lo, hi)
        const uint32_t md = rng(l0d == 0 ? 1 : 0, min(HD, A.QD[0] - 2 * l0d + 1));
        const uint32_t mh = rng(l0h == 0 ? 1 : 0, min(HH, A.QD[1] - 2 * l0h + 1));
        const uint32_t mw = rng(l0w == 0 ? 1 : 0, min(HW, A.QD[2] - 2 * l0w + 1));
        qmask = md | (mh << 5) | (mw << 14);
    };
    auto decode = [&](int tile) {                          // full decode (first tile, ragged last round)
        const int tpn = A.nt[0] * A.nt[1] * A.nt[2];
        c_n = tile / tpn;
        int tt = tile - c_n * tpn;
        c_w = tt % A.nt[2]; tt /= A.nt[2];
        c_h = tt % A.nt[1];
        c_d = tt / A.nt[1];
        setup();
    };
    auto advance = [&]() {                                 // tile += grid size
        c_w += INC.gw; int cy = c_w >= A.nt[2]; c_w -= cy ? A.nt[2] : 0;
        c_h += INC.gh + cy; cy = c_h >= A.nt[1]; c_h -= cy ? A.nt[1] : 0;
        c_d += INC.gd + cy; cy = c_d >= A.nt[0]; c_d -= cy ? A.nt[0] : 0;
        c_n += INC.gn + cy;
        setup();
    };
    auto dma_piece = [&](int i, int buf) {
        if (i < NPP) {
            const int pp = tq * NPP + i;
            const int rb = pp >> 2, pi = pp & 3;
            const int pd = pi >> 1, ph0 = (pi & 1) * 2;
            const bool ok = p_okw && (l0d + pd < A.PL[0]) && (l0h + ph0 + p_hb < A.PL[1]);
            wg_dma16(prs, ok ? p_voff : (int)0x80000000, __builtin_amdgcn_readfirstlane(pd * p_slab + ph0 * p_rowb + rb * 64),
                     (uint32_t)__builtin_amdgcn_readfirstlane(buf * BUF + pp * 1024));
        } else {
            const int j = i - NPP;
            if (tq + 4 * j < QPIECES) {
                const bool ok = (qsel[j] & qmask) == qsel[j];
                wg_dma16(qrs, ok ? qrel[j] : (int)0x80000000, 0, (uint32_t)__builtin_amdgcn_readfirstlane(buf * BUF + PBYTES + (tq + 4 * j) * 1024));
            }
        }
    };
    auto tr2 = [&](const char* b0) -> u32x4 {
        const uint2 a = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0)));
        const uint2 b = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0 + 4 * RB)));
        return u32x4{a.x, a.y, b.x, b.y};
    };
    auto compute = [&](int buf, bool stage) {
        constexpr int KS2 = TD * TH / 2, U = KS2 * NTS, QD_ = 3;
        // step ks = lattice rows 2 ks, 2 ks + 1 = slice ks >> 1, rows (ks & 1) * 2 + (q >> 1)
        const char* const pb = smem + buf * BUF + p_lane;
        const char* const qb = smem + buf * BUF + q_lane;
        u32x4 pf[2], qf[QD_ + 1];
        auto load_q = [&](int u) -> u32x4 {
            const int ks = u / NTS, ts = u % NTS;
            return tr2(qb + tapoff[ts] + ((ks >> 1) * 2 * HH + (ks & 1) * 4) * QROW);
        };
        pf[0] = tr2(pb);
#pragma unroll
        for (int u0 = 0; u0 < QD_; ++u0) qf[u0] = load_q(u0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = u / NTS, ts = u % NTS;
            if (u + QD_ < U) qf[(u + QD_) % (QD_ + 1)] = load_q(u + QD_);
            if (ts == 0 && ks + 1 < KS2) pf[(ks + 1) & 1] = tr2(pb + (ks + 1) * 2 * PROW);
            if (u < NPC && stage) dma_piece(u, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[ts] = H16<T>::mma32(pf[ks & 1], qf[u % (QD_ + 1)], acc[ts]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int G = gridDim.x, bx = blockIdx.x;
    auto perm = [&](int cnt) { return (cnt & 7) ? bx : (bx & 7) * (cnt >> 3) + (bx >> 3); };   // xcd_compact within a round
    const bool stager = rbw == 0;
    const int full = A.total_tiles / G;                       // rounds in which every workgroup has a tile
    const int pfull = perm(G);
    // tile of round k: k < full: k G + pfull (advance); the ragged last round: full G + perm(rest)
    auto to_round = [&](int k) {                               // -> true when this workgroup has a tile in round k (and decodes it)
        if (k < full) { if (k == 0) decode(pfull); else advance(); return true; }
        const int rest = A.total_tiles - full * G;
        if (k == full && bx < rest) { decode(full * G + perm(rest)); return true; }
        return false;
    };
    bool have = false, have_next = false;
    if (stager) {
        have = to_round(0);
        if (have) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) dma_piece(i, 0);
        }
        have_next = to_round(1);
    }
    have = 0 < full || bx < A.total_tiles - full * G;          // (every wave: does round 0 exist for this workgroup?)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int k = 0; have; ++k) {
        const bool next_exists = (k + 1 < full) || (k + 1 == full && bx < A.total_tiles - full * G);
        compute(buf, next_exists && stager);
        if (stager && next_exists) have_next = to_round(k + 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
        have = next_exists;
    }
    (void)have_next;
    const int n = lane & 31, mh = (lane >> 5) * 4;
    float* part_out = A.part + ((int64_t)blockIdx.x * (gridDim.y * 2 * gridDim.z) + (blockIdx.y * 2 + rbw) * gridDim.z + blockIdx.z) * ((int64_t)27 * 1024);
#pragma unroll
    for (int ts = 0; ts < NTS; ++ts) {
        if (tapw[ts] >= 0) {
            float* pt = part_out + (int64_t)tapw[ts] * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) pt[((r >> 2) * 8 + mh + (r & 3)) * 32 + n] = acc[ts][r];
        }
    }
}

static const WgItems g_wg_no_items = {};

// dW[r][k][tap] += sum_s part[s][pair][tap][r%32][k%32].
// Round 6: one workgroup per (pair, row r) and block of 32 slices. The first form had one thread per element in the PARTIAL layout
// ([tap][r][k], k fastest) and wrote dW[r][k][tap] from there: consecutive threads 27 floats apart, every 128-byte line of dW touched by
// 27 different threads (read-modify-write) -- 34 us for the 22 MB of a 320 x 320 layer, 0.65 ms of kernel time per step on the stream that
// ends the step. Here the (tap, k) tile of one output row is summed with coalesced reads (32 k = 128 bytes per tap and slice),
// transposed through LDS and written as ONE contiguous run of 32 x ntap floats of dW (PyTorch layout: [r][k][tap]). Same order of
// additions over the slices: same values.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, int S, int pairs, int kb, int ntap,
                                                      int R, int K, int64_t sr, int64_t sk, float* __restrict__ dw) {
    __shared__ float tile[27 * 32 + 32];
    const int pair = blockIdx.x >> 5, r = blockIdx.x & 31;
    const int rbi = pair / kb, kbi = pair % kb;
    const int rg = rbi * 32 + r;
    if (rg >= R) return;                                  // uniform
    const int ne = ntap * 32;
    const int64_t stride = (int64_t)pairs * ntap * 1024;
    const int s0 = blockIdx.y * 32, s1 = min(S, s0 + 32);
    for (int e = threadIdx.x; e < ne; e += 256) {
        const int t = e >> 5, k = e & 31;
        const float* src = part + ((int64_t)pair * ntap + t) * 1024 + r * 32 + k + (int64_t)s0 * stride;
        float acc = 0.f;
#pragma unroll 8
        for (int s2 = s0; s2 < s1; ++s2, src += stride) acc += *src;
        tile[k * ntap + t] = acc;                         // (27 is odd: consecutive k land on different banks)
    }
    __syncthreads();
    for (int w = threadIdx.x; w < ne; w += 256) {
        const int k = w / ntap, t = w - k * ntap;
        const int kg = kbi * 32 + k;
        if (kg >= K) continue;
        float* dst = &dw[rg * sr + kg * sk + t];          // sk == ntap: w runs through 32 x ntap CONSECUTIVE floats
        if (gridDim.y == 1) *dst += tile[w];              // single writer per element (S <= 32): no atomic needed, deterministic
        else atomicAdd(dst, tile[w]);
    }
}
static inline void wgrad_reduce_launch(const float* part, int S, int pairs, int kb, int ntap, int R, int K, int64_t sr, int64_t sk, float* dw,
                                       hipStream_t st) {
    k_wgrad_reduce<<<dim3((unsigned)pairs * 32, (unsigned)ceil_div(S, 32)), 256, 0, st>>>(part, S, pairs, kb, ntap, R, K, sr, sk, dw);
}

template <typename T, int KS, int MAXP, int MAXQ, bool PF, int NTS, bool SPLIT, int RBK = 1>
static int wg_launch(const WgArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    static NndetDevOnce attr;
    if (attr.need()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<T, KS, MAXP, MAXQ, PF, NTS, SPLIT, false, RBK>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<T, KS, MAXP, MAXQ, PF, NTS, SPLIT, true, RBK>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        attr.done();
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

// the 32x32x16 variant of k_wgrad3 (16-bit types) unless NNDET_WGRAD3_M32=0
static bool wgrad3_m32() {
    static const int w = getenv("NNDET_WGRAD3_M32") ? atoi(getenv("NNDET_WGRAD3_M32")) : 1;
    return w != 0;
}
#define WG3_LDS_ATTR(K) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048))

// k_wgrad3d (LDS-DMA form) for the 16-bit launches without a deferred input norm unless NNDET_WGRAD3D=0; its persistent grid:
// NNDET_WGRAD3D_WGS workgroups of 8 waves over all block pairs (one fits per CU)
static int wgrad3d_mode() {                        // 0: k_wgrad3 (register staging), 1: k_wgrad3d, 2 (default): k_wgrad3e
    const char* e = getenv("NNDET_WGRAD3D");       // (read per call: the tests compare the forms)
    return e ? atoi(e) : 2;
}
static bool wgrad3d_on() { return wgrad3d_mode() != 0; }
static int wgrad3d_slices(int pairs, int total_tiles) {
    const int total = getenv("NNDET_WGRAD3D_WGS") ? atoi(getenv("NNDET_WGRAD3D_WGS")) : 256;       // (read per call: the tests vary it)
    int S = (total < 1 || total > 256 ? 256 : total) / pairs;
    if (S < 1) S = 1;
    if (S > total_tiles) S = total_tiles;
    return S;
}
static constexpr size_t WG3D_LDS = 2 * (32 * 512 + 38 * 1024);
// one-shot event probe around ONE uniform k_wgrad3d launch (include/nndet_amd.h: nndet_probe_wgrad3d; bench.py times the kernel inside
// the training step on the stream it runs on -- the weight-gradient stream, which torch.cuda.Event on the current stream does not see)
static std::atomic<int64_t> g_probe_tiles{0};
static hipEvent_t g_probe_ev[2] = {nullptr, nullptr};
extern "C" int nndet_probe_wgrad3d(int64_t total_tiles, void* ev_start, void* ev_stop) {
    g_probe_tiles.store(0);
    if (total_tiles > 0 && ev_start && ev_stop) {
        g_probe_ev[0] = reinterpret_cast<hipEvent_t>(ev_start); g_probe_ev[1] = reinterpret_cast<hipEvent_t>(ev_stop);
        g_probe_tiles.store(total_tiles);
    }
    return 0;
}
template <typename T, bool ITEMS> static int wgrad3d_launch(const WgArgs& b, const WgItems& wi, dim3 g, hipStream_t st) {
    static NndetDevOnce at;
    if (at.need()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3d<T, ITEMS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG3D_LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3e<T, ITEMS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG3D_LDS));
        at.done();
    }
    int64_t want = ITEMS ? 0 : g_probe_tiles.load();
    const bool probe = want > 0 && want == b.total_tiles && g_probe_tiles.compare_exchange_strong(want, 0);
    if (probe) HIP_TRY(hipEventRecord(g_probe_ev[0], st));
    if (wgrad3d_mode() >= 2) {                    // the lean form (round 6); NNDET_WGRAD3D=1: k_wgrad3d
        Wg3eWalk wk;
        int gg = (int)g.x;
        wk.gw = gg % b.nt[2]; gg /= b.nt[2];
        wk.gh = gg % b.nt[1]; gg /= b.nt[1];
        wk.gd = gg % b.nt[0]; wk.gn = gg / b.nt[0];
        k_wgrad3e<T, ITEMS><<<g, 512, WG3D_LDS, st>>>(b, wi, wk);
    } else {
        k_wgrad3d<T, ITEMS><<<g, 512, WG3D_LDS, st>>>(b, wi);
    }
    LAUNCH_CHECK();
    if (probe) HIP_TRY(hipEventRecord(g_probe_ev[1], st));
    return 0;
}

// uniform k_wgrad3 launch for a 16-bit storage type
template <typename T> static int wgrad3_launch16(const WgArgs& b, dim3 g3, size_t lds3, hipStream_t st) {
    static NndetDevOnce at;
    if (at.need()) {
        WG3_LDS_ATTR((k_wgrad3<T, 2, false, 2>));
        WG3_LDS_ATTR((k_wgrad3<T, 2, false, 1>));
        WG3_LDS_ATTR((k_wgrad3<T, 2, true, 1>));
        WG3_LDS_ATTR((k_wgrad3<T, 2, false, 1, false, true>));
        WG3_LDS_ATTR((k_wgrad3<T, 2, true, 1, false, true>));
        at.done();
    }
    // QDEPTH 1: 248 registers, no spill. With depth 2 the kernel needs > 256 registers at two workgroups per CU and the
    // spill reloads (scratch shares vmcnt) serialise the staging loads of every tile (profiles/round2_wgrad3_spill.txt)
    static const int qd = getenv("NNDET_WGRAD3_QD") ? atoi(getenv("NNDET_WGRAD3_QD")) : 1;
    if (wgrad3_m32()) {
        if (b.qss) k_wgrad3<T, 2, true, 1, false, true><<<g3, 256, lds3, st>>>(b, g_wg_no_items);
        else k_wgrad3<T, 2, false, 1, false, true><<<g3, 256, lds3, st>>>(b, g_wg_no_items);
    } else if (b.qss) k_wgrad3<T, 2, true, 1><<<g3, 256, lds3, st>>>(b, g_wg_no_items);
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
            static const int dbg = nndet_timing_experiment("NNDET_WGRAD3_DBG");
            b.dbg = dbg;
            b.dbias = dbias;                  // P = dY here (not transposed): the kernel also produces the bias gradient
            *bias_done = dbias != nullptr;
            for (int i = 0; i < 3; ++i) { b.H[i] = st3[i] + 2; b.nt[i] = ceil_div(a.PL[i], st3[i]); }
            b.total_tiles = b.N * b.nt[0] * b.nt[1] * b.nt[2];
            b.m_tpn = magic(b.nt[0] * b.nt[1] * b.nt[2]); b.m_nt2 = magic(b.nt[2]); b.m_nt12 = magic(b.nt[1] * b.nt[2]);
            if (bf && !b.qss && wgrad3d_on()) {          // LDS-DMA form
                const int Sd = wgrad3d_slices(rb * kb, b.total_tiles);
                if (ws_bytes < (size_t)Sd * rb * kb * 27 * 1024 * sizeof(float)) return NNDET_EWORKSPACE;
                const int rcd = hf ? wgrad3d_launch<f16_t, false>(b, g_wg_no_items, dim3(Sd, rb, kb), st)
                                   : wgrad3d_launch<bf16_t, false>(b, g_wg_no_items, dim3(Sd, rb, kb), st);
                if (rcd) return rcd;
                wgrad_reduce_launch(b.part, Sd, rb * kb, kb, 27, b.R, b.K, b.sr, b.sk, dw, st);
                LAUNCH_CHECK();
                return 0;
            }
            const int S3 = wgrad_slices(rb * kb, b.total_tiles);
            const size_t need3 = (size_t)S3 * rb * kb * 27 * 1024 * sizeof(float);
            if (ws_bytes < need3) return NNDET_EWORKSPACE;
            const size_t lds3 = (size_t)32 * (8 * RB + RB / 2) + (size_t)60 * (10 * RB + RB / 2);
            dim3 g3(S3, rb, kb);
            if (bf) {
                const int rc3 = hf ? wgrad3_launch16<f16_t>(b, g3, lds3, st) : wgrad3_launch16<bf16_t>(b, g3, lds3, st);
                if (rc3) return rc3;
            } else {
                static NndetDevOnce at;
                if (at.need()) {
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<float, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<float, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
                    at.done();
                }
                if (b.qss) k_wgrad3<float, 1, true><<<g3, 256, lds3, st>>>(b, g_wg_no_items);
                else k_wgrad3<float, 1, false><<<g3, 256, lds3, st>>>(b, g_wg_no_items);
            }
            LAUNCH_CHECK();
            wgrad_reduce_launch(b.part, S3, rb * kb, kb, 27, b.R, b.K, b.sr, b.sk, dw, st);
            LAUNCH_CHECK();
            return 0;
        }
    }
    static const int dbg_skip_s = nndet_timing_experiment("NNDET_WGRAD_DBG_SKIP_STRIDED");
    if (dbg_skip_s && strided && T == 27) return 0;
    // stride (2, 2, 2) / 3x3x3 / pad 1 in 16 bits, an even number of row blocks, no deferred input norm: k_wgrad3s (NNDET_WGRAD3S=0: off)
    static const int s3 = getenv("NNDET_WGRAD3S") ? atoi(getenv("NNDET_WGRAD3S")) : 1;
    if (s3 && bf && !tr && T == 27 && !a.qss && rb % 2 == 0 && c->s[0] == 2 && c->s[1] == 2 && c->s[2] == 2 && c->k[0] == 3 && c->k[1] == 3 &&
        c->k[2] == 3 && c->p[0] == 1 && c->p[1] == 1 && c->p[2] == 1 &&
        (int64_t)a.PL[0] * a.PL[1] * a.PL[2] * a.Cp * 2 < (1LL << 31) && (int64_t)a.QD[0] * a.QD[1] * a.QD[2] * a.Cq * 2 < (1LL << 31)) {
        WgArgs b = a;
        b.TD = 2; b.TH = 4;
        const int st3[3] = {2, 4, 8};
        for (int i = 0; i < 3; ++i) b.nt[i] = ceil_div(a.PL[i], st3[i]);
        b.total_tiles = b.N * b.nt[0] * b.nt[1] * b.nt[2];
        const int wgs = getenv("NNDET_WGRAD3S_WGS") ? atoi(getenv("NNDET_WGRAD3S_WGS")) : 256;     // (read per call: the tests vary it)
        int Ss = (wgs < 2 || wgs > 256 ? 256 : wgs) * 2 / (rb * kb);
        if (Ss < 1) Ss = 1;
        if (Ss > b.total_tiles) Ss = b.total_tiles;
        if (ws_bytes < (size_t)Ss * rb * kb * 27 * 1024 * sizeof(float)) return NNDET_EWORKSPACE;
        Wg3sInc inc;
        int g = Ss;
        inc.gw = g % b.nt[2]; g /= b.nt[2];
        inc.gh = g % b.nt[1]; g /= b.nt[1];
        inc.gd = g % b.nt[0]; inc.gn = g / b.nt[0];
        constexpr size_t lds_s = 2 * (2 * 2 * 4 * 512 + 48 * 1024);
        static NndetDevOnce at;
        if (at.need()) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3s<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3s<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
            at.done();
        }
        dim3 gs(Ss, rb / 2, kb);
        if (hf) k_wgrad3s<f16_t><<<gs, 512, lds_s, st>>>(b, inc);
        else k_wgrad3s<bf16_t><<<gs, 512, lds_s, st>>>(b, inc);
        LAUNCH_CHECK();
        wgrad_reduce_launch(b.part, Ss, rb * kb, kb, 27, b.R, b.K, b.sr, b.sk, dw, st);
        LAUNCH_CHECK();
        return 0;
    }
    if (hf) rc = KS == 8 ? wg_dispatch<f16_t, 8, 4, 10, true>(a, grid, lds, st) : wg_dispatch<f16_t, 2, 1, 12, true>(a, grid, lds, st, rbk);
    else if (bf) rc = KS == 8 ? wg_dispatch<bf16_t, 8, 4, 10, true>(a, grid, lds, st) : wg_dispatch<bf16_t, 2, 1, 12, true>(a, grid, lds, st, rbk);
    else rc = KS == 8 ? wg_dispatch<float, 8, 8, 20, false>(a, grid, lds, st) : wg_dispatch<float, 2, 2, 24, false>(a, grid, lds, st);
    if (rc) return rc;
    wgrad_reduce_launch(a.part, slices, rb * kb, kb, a.ntap, a.R, a.K, a.sr, a.sk, dw, st);
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
        const int n1 = ceil_div(it->dims[i][1], 8), n2 = ceil_div(it->dims[i][2], 8);
        wi.m_nt2[i] = n2 <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)n2 + 1ull);
        wi.m_nt12[i] = n1 * n2 <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)(n1 * n2) + 1ull);
        total += ceil_div(it->dims[i][0], 4) * ceil_div(it->dims[i][1], 8) * ceil_div(it->dims[i][2], 8);
    }
    b.total_tiles = total;
    // (PL / QD / nt of the uniform kernel are unused in ITEMS mode; keep them at the first item's values for the dead code)
    for (int i = 0; i < 3; ++i) { b.PL[i] = b.QD[i] = it->dims[0][i]; }
    b.nt[0] = ceil_div(b.PL[0], 4); b.nt[1] = ceil_div(b.PL[1], 8); b.nt[2] = ceil_div(b.PL[2], 8);
    const int rb = b.Cp / 32, kb = b.Cq / 32;
    if (bf && wgrad3d_on()) {
        const int Sd = wgrad3d_slices(rb * kb, b.total_tiles);
        if (!ws || ws_bytes < (size_t)Sd * rb * kb * 27 * 1024 * sizeof(float)) return NNDET_EWORKSPACE;
        b.part = reinterpret_cast<float*>(ws);
        const int rcd = hf ? wgrad3d_launch<f16_t, true>(b, wi, dim3(Sd, rb, kb), st) : wgrad3d_launch<bf16_t, true>(b, wi, dim3(Sd, rb, kb), st);
        if (rcd) return rcd;
        wgrad_reduce_launch(b.part, Sd, rb * kb, kb, 27, b.R, b.K, b.sr, b.sk, dw, st);
        LAUNCH_CHECK();
        return 0;
    }
    const int S3 = wgrad_slices(rb * kb, b.total_tiles);
    const size_t need3 = (size_t)S3 * rb * kb * 27 * 1024 * sizeof(float);
    if (!ws || ws_bytes < need3) return NNDET_EWORKSPACE;
    b.part = reinterpret_cast<float*>(ws);
    const size_t lds3 = (size_t)32 * (8 * RB + RB / 2) + (size_t)60 * (10 * RB + RB / 2);
    dim3 g3(S3, rb, kb);
    static NndetDevOnce at;
    if (at.need()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<bf16_t, 2, false, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<f16_t, 2, false, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<float, 1, false, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        at.done();
    }
    static NndetDevOnce atw;
    if (atw.need()) {
        WG3_LDS_ATTR((k_wgrad3<bf16_t, 2, false, 1, true, true>));
        WG3_LDS_ATTR((k_wgrad3<f16_t, 2, false, 1, true, true>));
        atw.done();
    }
    if (bf && wgrad3_m32()) {
        if (hf) k_wgrad3<f16_t, 2, false, 1, true, true><<<g3, 256, lds3, st>>>(b, wi);
        else k_wgrad3<bf16_t, 2, false, 1, true, true><<<g3, 256, lds3, st>>>(b, wi);
    } else if (hf) k_wgrad3<f16_t, 2, false, 1, true><<<g3, 256, lds3, st>>>(b, wi);
    else if (bf) k_wgrad3<bf16_t, 2, false, 1, true><<<g3, 256, lds3, st>>>(b, wi);
    else k_wgrad3<float, 1, false, 2, true><<<g3, 256, lds3, st>>>(b, wi);
    LAUNCH_CHECK();
    wgrad_reduce_launch(b.part, S3, rb * kb, kb, 27, b.R, b.K, b.sr, b.sk, dw, st);
    LAUNCH_CHECK();
    return 0;
}
