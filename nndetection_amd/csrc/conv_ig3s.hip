// Forward pass of the FIRST stride-2 stage transition (3x3x3, stride (2, 2, 2), pad 1, 32 -> 64 padded channels, 16-bit) for gfx950:
// k_ig3s, the k_ig3r / k_wgrad3s recipe applied to a strided forward convolution (round 4).
//
// k_igemm runs this layer at 0.38 ms (135.9 GFLOP, 786 MB algorithmic = 0.14 of the MFMA peak, 0.26 of HBM) on the SERIAL forward
// chain: its waves wait on memory 51 % of their cycles (register-staged halo, one tile in flight per workgroup; MFMA pipe busy 15 %,
// profiles/round4_strided_fwd_pmc.txt). Here
//   * persistent workgroups (one per CU, 4 waves, one per SIMD) walk 2 x 4 x 8-point output tiles; the 5 x 9 x 17-voxel halo of the
//     NEXT tile goes global -> LDS by LDS-DMA (48 pieces of 1 KB, 12 per wave, issued under the first MFMAs of the current tile) into
//     the second 48 KB buffer; lanes outside the tensor use an out-of-range offset and the hardware writes the zero padding;
//   * halo rows are stored de-interleaved along W (9 even offsets, then 8 odd ones), so the 8 W-points of a row read CONSECUTIVE
//     voxels for every tap (c = 0 / 2: the even run at slot 0 / 1, c = 1: the odd run);
//   * ALL weights of a wave's 32 output channels live in registers: 27 taps x 2 input-channel halves x 4 VGPRs = 216 per lane, the
//     A operands of v_mfma_f32_32x32x16; wave = (output-channel half wv & 1, d-plane of the tile wv >> 1): 54 MFMAs per tile on 4
//     rotating accumulators, B operands = one ds_read_b128 per MFMA (8 input channels of a voxel), prefetched 3 MFMAs ahead;
//   * the tile walk is a mixed-radix increment by the grid size (k_wgrad3s);
//   * epilogue after the tile barrier: sum of the 4 accumulators (+ bias), 16-bit pack, 8-byte stores (each lane holds 4 x 4
//     consecutive channels of one point); the InstanceNorm statistics of the ROUNDED outputs stay in registers over the tiles of an
//     image and are flushed once per image (32-lane shuffle reduction, fp64 atomics into the replica table).
// Other channel counts keep k_igemm: 64 -> 128 and 128 -> 256 need 432 / 1728 weight registers per lane in this form.
#include "common.h"
#include "conv_common.h"

typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));

// D (VGPRs) = A (ACCUMULATION registers) x B (VGPRs) + C for the kernels whose weights live in registers: with 216 weight registers per
// lane hipcc keeps them in AGPRs and copies every A operand back with 4 v_accvgpr_read before its MFMA -- 216 VALU instructions per
// tile and wave, ALL the VALU work of k_ig3s and the reason its waves spent half their cycles issuing (profiles/round6_ig3s_pre_pmc.txt).
// gfx90a+ MFMAs read srcA from an AGPR directly; only inline assembly can say so. Hazards hipcc no longer sees (it does not look into the
// assembly) are covered by construction: an accumulator is reused by every 4th MFMA (no back-to-back dependency), the first MFMA of a
// tile takes C = 0 (no VALU-written accumulator), wait states in front of it (the epilogue's reads of the previous tile) and behind
// the last one (VALU reads of the results) are explicit; the B operands are ordinary register inputs (hipcc waits for their LDS loads).
template <typename T> struct MfmaA;
template <> struct MfmaA<bf16_t> {
    __device__ static __forceinline__ void first32(f32x16_t& d, const u32x4& a, const u32x4& b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "a"(a), "v"(b)); }
    __device__ static __forceinline__ void acc32(f32x16_t& d, const u32x4& a, const u32x4& b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "v"(b)); }
};
template <> struct MfmaA<f16_t> {
    __device__ static __forceinline__ void first32(f32x16_t& d, const u32x4& a, const u32x4& b) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "a"(a), "v"(b)); }
    __device__ static __forceinline__ void acc32(f32x16_t& d, const u32x4& a, const u32x4& b) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "v"(b)); }
};
#ifndef IG3S_ASM_MFMA
#define IG3S_ASM_MFMA 1
#endif

struct Ig3sArgs {
    const void* x; const void* w; const float* bias; void* y; double* stats;
    int32_t N;
    int32_t I[3], O[3];          // input / output spatial dims
    int32_t nt[3], total_tiles;  // (2, 4, 8)-point tiles per axis
    int32_t gw, gh, gd, gn;      // grid size = gw + nt[2] * (gh + nt[1] * (gd + nt[0] * gn))
    // k_ig3s<.., PRE = true>: x is the PRE-norm output of the previous convolution; the kernel reads relu?(x * scale + shift) and
    // writes that normalised tensor to xn on the way (each voxel by the ONE tile whose 4 x 8 x 16 input core holds it)
    const float* ss;             // [N][32][2] (scale, shift) fp32 (nndet_norm_finalize)
    void* xn;
    int32_t ss_relu;
};

__device__ __forceinline__ void ig3s_dma16(__amdgpu_buffer_rsrc_t rs, int voff, uint32_t lds_dst) {
    // M0 is written in the same statement that reads it; hipcc does not count this load: waited for by hand (vmcnt(0) before the barrier)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ig3s_rsrc(const char* p, int num_records) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(((uint64_t)hi << 32) | lo), 0, num_records, 0x00020000);
}

template <typename T, bool STATS, bool PRE>
__global__ __launch_bounds__(256, 1) void k_ig3s(const Ig3sArgs A) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    constexpr int RB = 64, CY = 64, TD = 2, TH = 4, HD = 2 * TD + 1, HH = 2 * TH + 1, HW = 17, NEV = 9;
    constexpr int QROW = HW * RB, QVOX = HD * HH * HW, QPIECES = (QVOX * 4 + 63) / 64, BUF = QPIECES * 1024, NQ = QPIECES / 4;
    static_assert(QPIECES % 4 == 0, "pieces are dealt to 4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int coh = wv & 1, pth = wv >> 1;                 // output-channel half, d-plane of the tile
    const int n_pt = lane & 31, kg = lane >> 5;            // MFMA column (point) / 8-channel group of this lane
    const int rr = n_pt >> 3, pw = n_pt & 7;               // the wave's 32 points = 4 rows x 8 W-points of plane pth

    // ---- weights -> registers: packed mode 0 = [tap][64 rows][32 k] (2-byte elements)
    u32x4 wr[27][2];
    {
        const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A.w), 0, 27 * 64 * 32 * 2, 0x00020000);
#pragma unroll
        for (int tp = 0; tp < 27; ++tp)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
                wr[tp][kh] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, ((coh * 32 + n_pt) * 32 + kh * 16 + kg * 8) * 2, tp * (64 * 32 * 2), 0));
    }
    float bia[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bia[r] = A.bias ? A.bias[coh * 32 + (r >> 2) * 8 + kg * 4 + (r & 3)] : 0.f;
    if constexpr (PRE) {       // (consumed here, see t_coef: the epilogue must not wait for them by hipcc's count)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(bia[r]));
    }

    // ---- DMA geometry: piece qp = wv + 4 j: granule G = qp * 64 + lane = halo slot G >> 2 (row-major over 5 x 9 x 17 slots, slot s of a
    // row = W offset 2 s (s < 9) or 2 (s - 9) + 1), 16-byte part G & 3
    const int q_rowb = A.I[2] * RB, q_slab = A.I[1] * q_rowb;
    uint32_t qsel[NQ];          // bit hd | bit 5 + hh | bit 14 + W offset; bit 31 = no such granule
    int qrel[NQ];
    uint32_t qcls = 0, qown = 0;   // PRE: 2 bits per piece = swizzle class of the granule's halo row; 1 bit per piece = the voxel is in the tile's core
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int G = (wv + 4 * j) * 64 + lane;
        const int vox = G >> 2;
        const int row = vox / HW, sl = vox - row * HW;
        const int hd = row / HH, hh = row - hd * HH;
        const int jw = sl < NEV ? 2 * sl : 2 * (sl - NEV) + 1;
        qsel[j] = vox < QVOX ? (1u << hd) | (1u << (5 + hh)) | (1u << (14 + jw)) : 0x80000000u;
        // Bank swizzle (round 5): the 16-byte part at position G & 3 of a halo voxel holds the voxel's part (G & 3) ^ ((hh >> 1) & 3). A lane of
        // the MFMA phase reads ONE part of its voxel; with the parts in place, the 16 lanes of a ds_read_b128 group (4 rows x 4 columns, rows
        // 2 x 1088 B apart, columns 64 B) met on 4 of the 16 part slots: a 4-way conflict on every B-operand read -- SQ_LDS_BANK_CONFLICT was
        // 72 % of the LDS cycles of this kernel, the LDS 49 % busy (profiles/round5_step_pmc_survey.txt). The source address of a DMA lane is free.
        qrel[j] = hd * q_slab + hh * q_rowb + jw * RB + (((G & 3) ^ ((hh >> 1) & 3)) * 16);
        if constexpr (PRE) {
            qcls |= (uint32_t)((hh >> 1) & 3) << (2 * j);
            qown |= (uint32_t)(vox < QVOX && hd >= 1 && hd <= 2 * TD && hh >= 1 && hh <= 2 * TH && jw >= 1 && jw <= 16) << j;
        }
    }
    int64_t c_org = 0;                                     // byte offset of the decoded tile's halo origin in x (and in xn)
    __amdgpu_buffer_rsrc_t qrs;
    uint32_t qmask = 0;
    int c_n = 0, c_d = 0, c_h = 0, c_w = 0;               // mixed-radix coordinates of the tile last decoded (= the one being staged)
    auto setup = [&]() {
        const int l0d = c_d * TD, l0h = c_h * TH, l0w = c_w * 8;
        const int64_t q_org = (int64_t)c_n * A.I[0] * q_slab + (int64_t)(2 * l0d - 1) * q_slab + (2 * l0h - 1) * q_rowb + (2 * l0w - 1) * RB;
        qrs = ig3s_rsrc(reinterpret_cast<const char*>(A.x) + q_org, 0x7ffffff0);
        c_org = q_org;
        auto rng = [](int lo, int hi) -> uint32_t { return (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u); };   // bits [lo, hi)
        const uint32_t md = rng(l0d == 0 ? 1 : 0, min(HD, A.I[0] - 2 * l0d + 1));
        const uint32_t mh = rng(l0h == 0 ? 1 : 0, min(HH, A.I[1] - 2 * l0h + 1));
        const uint32_t mw = rng(l0w == 0 ? 1 : 0, min(HW, A.I[2] - 2 * l0w + 1));
        qmask = md | (mh << 5) | (mw << 14);
    };
    auto decode = [&](int tile) {
        const int tpn = A.nt[0] * A.nt[1] * A.nt[2];
        c_n = tile / tpn;
        int tt = tile - c_n * tpn;
        c_w = tt % A.nt[2]; tt /= A.nt[2];
        c_h = tt % A.nt[1];
        c_d = tt / A.nt[1];
        setup();
    };
    auto advance = [&]() {                                 // tile += grid size
        c_w += A.gw; int cy = c_w >= A.nt[2]; c_w -= cy ? A.nt[2] : 0;
        c_h += A.gh + cy; cy = c_h >= A.nt[1]; c_h -= cy ? A.nt[1] : 0;
        c_d += A.gd + cy; cy = c_d >= A.nt[0]; c_d -= cy ? A.nt[0] : 0;
        c_n += A.gn + cy;
        setup();
    };
    auto dma_piece = [&](int j, int buf) {
        const bool ok = (qsel[j] & qmask) == qsel[j];
        ig3s_dma16(qrs, ok ? qrel[j] : (int)0x80000000, (uint32_t)__builtin_amdgcn_readfirstlane(buf * BUF + (wv + 4 * j) * 1024));
    };

    // ---- PRE: the transform of a landed halo, relu?(x * scale + shift) in place + the core voxels out to xn. A lane works on channel
    // part lane & 3 of its voxels (ONE set of 8 + 8 coefficients in registers); with the bank swizzle that part sits at position
    // (lane & 3) ^ class(row) of the voxel: written by another lane of the SAME wave's piece, so the wave's own vmcnt wait orders it.
    __amdgpu_buffer_rsrc_t trs = qrs;                      // xn at the origin of the tile being transformed
    uint32_t tmask = 0;
    int t_n = -1, ss_n = -1;
    f32x2_t scp[4], shp[4];                                // (pairs: the operands of v_pk_fma_f32)
    const uint32_t floor2 = A.ss_relu ? 0u : 0x80008000u;  // v_pk_max_i16 against (0, 0) = ReLU on the packed pair (AffinePiece); against INT16_MIN = identity
    u32x4 tv = {0, 0, 0, 0}, tr = {0, 0, 0, 0};
    auto take_t = [&]() {                                  // the decoded tile becomes the one to transform
        if constexpr (PRE) {
            tmask = qmask; t_n = c_n;
            trs = ig3s_rsrc(reinterpret_cast<const char*>(A.xn) + c_org, 0x7ffffff0);
        }
    };
    auto t_coef = [&]() {
        if constexpr (PRE) {
            if (t_n != ss_n) {
                float sc[8], sh[8];
                load_affine<8>(A.ss, t_n, 32, (lane & 3) * 8, sc, sh); ss_n = t_n;
#pragma unroll
                for (int i = 0; i < 4; ++i) { scp[i] = f32x2_t{sc[2 * i], sc[2 * i + 1]}; shp[i] = f32x2_t{sh[2 * i], sh[2 * i + 1]}; }
                // consumed HERE: otherwise hipcc waits for these loads where the transform first uses them -- by ITS count of what is
                // outstanding, a vmcnt(0) in every tile that drains the halo requests it knows nothing about
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(scp[i]), "+v"(shp[i]));
            }
        }
    };
    auto t_addr = [&](int j, int buf) -> int { return buf * BUF + (wv + 4 * j) * 1024 + ((lane * 16) ^ (int)(((qcls >> (2 * j)) & 3u) << 4)); };
    auto t_read = [&](int j, int buf) { tv = *reinterpret_cast<const u32x4*>(smem + t_addr(j, buf)); };
    auto t_half = [&](int h) {                             // AffinePiece<T>::apply on dwords 2 h, 2 h + 1 (the same operations in the same order)
        typedef short s16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 2 * h; i < 2 * h + 2; ++i) {
            const f32x2_t x = {H16<T>::lo(tv[i]), H16<T>::hi(tv[i])};
            const f32x2_t r = __builtin_elementwise_fma(x, scp[i], shp[i]);
            const uint32_t pk = H16<T>::pack2(r[0], r[1]);
            tr[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), __builtin_bit_cast(s16x2, floor2)));
        }
    };
    auto t_write = [&](int j, int buf) {
        const bool ok = (qsel[j] & tmask) == qsel[j];      // (the zero padding stays zero)
        const u32x4 o = {ok ? tr[0] : 0u, ok ? tr[1] : 0u, ok ? tr[2] : 0u, ok ? tr[3] : 0u};
        *reinterpret_cast<u32x4*>(smem + t_addr(j, buf)) = o;
        const bool own = ok && ((qown >> j) & 1u);
        // (no SGPR offset on a 16-byte store: tests/test_isa_hazards.py; lanes that do not own their voxel use an out-of-range offset, dropped by the hardware)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, o), trs, own ? (qrel[j] ^ (int)(((qcls >> (2 * j)) & 3u) << 4)) : (int)0x80000000, 0, 0);
    };

    // ---- fragment base: point (plane pth, row rr, column pw) at tap (0, 0, 0) = halo row (2 pth, 2 rr), even slot pw; 8 channels kg
    const int q_lane = ((2 * pth) * HH + 2 * rr) * QROW + pw * RB;
    // part kh * 2 + kg of the voxel in halo row 2 rr + b sits at position (kh * 2 + kg) ^ ((2 rr + b) >> 1): rr for b = 0 / 1, rr + 1 for b = 2
    int q_part[2][2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        q_part[kh][0] = q_lane + (((kh * 2 + kg) ^ (rr & 3)) * 16);
        q_part[kh][1] = q_lane + (((kh * 2 + kg) ^ ((rr + 1) & 3)) * 16);
    }
    f32x16_t acc[4];
    float ssum[16], ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ssum[r] = 0.f; ssq[r] = 0.f; }

    // PRE: while tile k runs out of `buf`, the halo of tile k + 2 is requested into `sbuf` (MFMA slots 0 .. 11) and the one of tile k + 1, landed
    // in `tbuf`, is transformed (4 slots per piece from slot T0 on: read | dwords 0, 1 | dwords 2, 3 | write + store)
    auto compute = [&](int buf, auto stage, int sbuf, auto trans, int tbuf) {
        constexpr int U = 54, QD_ = 3;
        const char* const qb = smem + buf * BUF;
#if !IG3S_ASM_MFMA
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#endif
        u32x4 bf[QD_ + 1];
        auto load_b = [&](int u) -> u32x4 {
            const int tp = u >> 1, kh = u & 1;
            const int a = tp / 9, b = (tp / 3) % 3, c = tp % 3;
            return *reinterpret_cast<const u32x4*>(qb + q_part[kh][b == 2] + (a * HH + b) * QROW + (c == 1 ? NEV * RB : c == 2 ? RB : 0));
        };
#pragma unroll
        for (int u0 = 0; u0 < QD_; ++u0) bf[u0] = load_b(u0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u + QD_ < U) bf[(u + QD_) % (QD_ + 1)] = load_b(u + QD_);
            if constexpr (PRE) {
                constexpr int T0 = 4;
                if (u < NQ && stage) dma_piece(u, sbuf);
                if (trans && u >= T0 && u < T0 + 4 * NQ) {
                    const int pj = (u - T0) >> 2, ps = (u - T0) & 3;
                    if (u == T0) {
                        // the halo in `tbuf` was requested one tile ago. Issued after it, by this wave, ALWAYS (masked lanes use out-of-range
                        // offsets, the instructions are not predicated): NQ xn stores + 2 output stores of the previous tile (+ the T0
                        // requests of this one); a statistics flush in between only makes the wait longer than needed
                        if (stage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ + 2 + T0) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ + 2) : "memory");
                    }
                    if (ps == 0) t_read(pj, tbuf);
                    else if (ps == 1) t_half(0);
                    else if (ps == 2) t_half(1);
                    else t_write(pj, tbuf);
                }
            } else {
                if (u < NQ && stage) dma_piece(u, buf ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#if IG3S_ASM_MFMA
            if (u == 0) asm volatile("s_nop 4");              // (the epilogue of the previous tile read these registers)
            if (u < 4) MfmaA<T>::first32(acc[u & 3], wr[u >> 1][u & 1], bf[u % (QD_ + 1)]);
            else MfmaA<T>::acc32(acc[u & 3], wr[u >> 1][u & 1], bf[u % (QD_ + 1)]);
            if (u == U - 1) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");   // 20 wait states before anything reads the results
#else
            acc[u & 3] = H16<T>::mma32(wr[u >> 1][u & 1], bf[u % (QD_ + 1)], acc[u & 3]);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // statistics of image n: sum over the 32 points (lanes with the same kg), fp64 atomics into this workgroup's replica
    auto flush_stats = [&](int n) {
        if constexpr (STATS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = ssum[r], s2 = ssq[r];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); s2 += __shfl_xor(s2, o, 64); }
                if (n_pt == 0) {
                    const int co = coh * 32 + (r >> 2) * 8 + kg * 4 + (r & 3);
                    double* dst = A.stats + (((int64_t)(blockIdx.x % NNDET_STATS_REPLICAS) * A.N + n) * CY + co) * 2;
                    atomicAdd(dst, (double)s); atomicAdd(dst + 1, (double)s2);
                }
                ssum[r] = 0.f; ssq[r] = 0.f;
            }
        }
    };
    int st_n = -1;                                          // image the register statistics belong to
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(A.y, 0, PRE ? (int)((int64_t)A.N * A.O[0] * A.O[1] * A.O[2] * CY * 2) : 0, 0x00020000);
    auto epilogue = [&](int n, int td, int th, int tw) {
        if (STATS && n != st_n) { if (st_n >= 0) flush_stats(st_n); st_n = n; }
        const int od = td * TD + pth, oh = th * TH + rr, ow = tw * 8 + pw;
        const bool valid = od < A.O[0] && oh < A.O[1] && ow < A.O[2];
        T* const yp = reinterpret_cast<T*>(A.y) + ((((int64_t)n * A.O[0] + od) * A.O[1] + oh) * A.O[2] + ow) * CY + coh * 32;
        uint2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (acc[0][g * 4 + i] + acc[1][g * 4 + i]) + (acc[2][g * 4 + i] + acc[3][g * 4 + i]) + bia[g * 4 + i];
            pk[g].x = H16<T>::pack2(v[0], v[1]); pk[g].y = H16<T>::pack2(v[2], v[3]);
            if constexpr (STATS) {                            // of the ROUNDED values (what the norm kernels will read)
                if (valid) {
                    const float r0 = H16<T>::lo(pk[g].x), r1 = H16<T>::hi(pk[g].x), r2 = H16<T>::lo(pk[g].y), r3 = H16<T>::hi(pk[g].y);
                    ssum[g * 4 + 0] += r0; ssum[g * 4 + 1] += r1; ssum[g * 4 + 2] += r2; ssum[g * 4 + 3] += r3;
                    ssq[g * 4 + 0] += r0 * r0; ssq[g * 4 + 1] += r1 * r1; ssq[g * 4 + 2] += r2 * r2; ssq[g * 4 + 3] += r3 * r3;
                }
            }
        }
        // lanes l and l + 32 hold the channel quads 8 g + {0..3} / {4..7} of the SAME point: v_permlane32_swap hands the lower lane the
        // upper one's quads of groups 0 / 1 and the upper lane the lower one's of groups 2 / 3 -> two 16-byte stores per lane instead of
        // four 8-byte ones (lower lane: channels 0..15, upper lane: 16..31 of the wave's 32). Both lanes belong to one point: same `valid`.
        typedef unsigned int ig3s_v2u __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const ig3s_v2u sx = __builtin_amdgcn_permlane32_swap(pk[h].x, pk[h + 2].x, false, false);
            const ig3s_v2u sy = __builtin_amdgcn_permlane32_swap(pk[h].y, pk[h + 2].y, false, false);
            if constexpr (PRE) {   // (always issued: the vmcnt arithmetic of the transform counts on two stores per tile)
                const int64_t off = (reinterpret_cast<const char*>(yp + kg * 16 + h * 8) - reinterpret_cast<const char*>(A.y));
                __builtin_amdgcn_raw_buffer_store_b128(v4u_t{sx[0], sy[0], sx[1], sy[1]}, yrs, valid ? (int)off : (int)0x80000000, 0, 0);
            } else {
                if (valid) *reinterpret_cast<u32x4*>(yp + kg * 16 + h * 8) = u32x4{sx[0], sy[0], sx[1], sy[1]};
            }
        }
    };

    const int G = gridDim.x, bx = blockIdx.x;
    auto perm = [&](int cnt) { return (cnt & 7) ? bx : (bx & 7) * (cnt >> 3) + (bx >> 3); };   // xcd_compact within a round
    const int full = A.total_tiles / G, rest = A.total_tiles - full * G;
    const int pfull = perm(G);
    auto exists = [&](int k) { return k < full || (k == full && bx < rest); };
    auto to_round = [&](int k) {                               // decode round k's tile of this workgroup (if it has one)
        if (k < full) { if (k == 0) decode(pfull); else advance(); }
        else if (k == full && bx < rest) decode(full * G + perm(rest));
    };
    if (!exists(0)) return;
    if constexpr (PRE) {
        // three halo buffers: tile k in the MFMAs | tile k + 1 landed, being transformed | tile k + 2 in flight. No vmcnt(0) in the steady
        // state: a wave waits for ITS pieces of the halo it is about to transform (in-order counter, see compute), the barrier per tile
        // orders the LDS traffic (transform -> MFMA reads of the next tile, MFMA reads -> the next request into that buffer).
        auto lds_barrier = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        to_round(0);
#pragma unroll
        for (int j = 0; j < NQ; ++j) dma_piece(j, 0);
        take_t();
        int e_n = c_n, e_d = c_d, e_h = c_h, e_w = c_w;       // tile k (epilogue)
        int n_n = 0, n_d = 0, n_h = 0, n_w = 0;               // tile k + 1
        t_coef();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < NQ; ++j) { t_read(j, 0); t_half(0); t_half(1); t_write(j, 0); }
        if (exists(1)) {
            to_round(1);
#pragma unroll
            for (int j = 0; j < NQ; ++j) dma_piece(j, 1);
            take_t();
            n_n = c_n; n_d = c_d; n_h = c_h; n_w = c_w;
            if (exists(2)) to_round(2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        int buf = 0;
        for (int k = 0; exists(k); ++k) {
            const bool nx = exists(k + 1), nx2 = exists(k + 2);
            const int b1 = buf == 2 ? 0 : buf + 1, b2 = b1 == 2 ? 0 : b1 + 1;
            if (nx) t_coef();
            // (compile-time flags: the steady state is one branch-free body)
            if (nx2) compute(buf, std::true_type{}, b2, std::true_type{}, b1);
            else if (nx) compute(buf, std::false_type{}, b2, std::true_type{}, b1);
            else compute(buf, std::false_type{}, b2, std::false_type{}, b1);
            const int p_n = e_n, p_d = e_d, p_h = e_h, p_w = e_w;
            e_n = n_n; e_d = n_d; e_h = n_h; e_w = n_w;
            if (nx2) {
                take_t();
                n_n = c_n; n_d = c_d; n_h = c_h; n_w = c_w;
                if (exists(k + 3)) to_round(k + 3);
            }
            lds_barrier();
            epilogue(p_n, p_d, p_h, p_w);
            buf = b1;
        }
        if (STATS && st_n >= 0) flush_stats(st_n);
        return;
    }
    to_round(0);
#pragma unroll
    for (int j = 0; j < NQ; ++j) dma_piece(j, 0);
    int cur_n = c_n, cur_d = c_d, cur_h = c_h, cur_w = c_w;   // coordinates of the tile in the MFMAs
    to_round(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int k = 0; exists(k); ++k) {
        const bool nx = exists(k + 1);
        compute(buf, nx, 0, false, 0);
        const int e_n = cur_n, e_d = cur_d, e_h = cur_h, e_w = cur_w;
        cur_n = c_n; cur_d = c_d; cur_h = c_h; cur_w = c_w;   // (the scalars of round k + 1, decoded before this phase)
        if (nx) to_round(k + 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                      // every wave is done reading `buf`, the next halo has landed in the other buffer
        epilogue(e_n, e_d, e_h, e_w);                         // (after the barrier: its stores are not waited for by the next one's vmcnt(0) ... they are, 54 MFMAs later)
        buf ^= 1;
    }
    if (STATS && st_n >= 0) flush_stats(st_n);
}

// ------------------------------------------------------------------------------------------------ 64 input channels
// k_ig3s2: the SECOND stride-2 transition (64 -> 128 padded channels: 0.23 ms in k_igemm on the serial forward chain). Two 32-channel
// chunks of the input: the weights of a wave's 16 output channels for both chunks are 27 x 2 x 4 = 216 registers (A operands of
// v_mfma_f32_16x16x32), a workgroup = 4 waves = 64 output channels (blockIdx.y picks the block of 64; the input is read once per
// block). The pipeline unit is (tile, chunk): while the 108 MFMAs of chunk c of a tile run out of one 48 KB buffer, the halo of the next
// unit -- chunk 1 of the same tile, then chunk 0 of the next tile -- lands in the other one; the accumulators (4 point tiles of 16)
// live across the two chunks, the epilogue follows the second. Everything else as k_ig3s.
template <typename T, bool STATS>
__global__ __launch_bounds__(256, 1) void k_ig3s2(const Ig3sArgs A, int cout_p) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    constexpr int RB = 64, CX = 64, TD = 2, TH = 4, HD = 2 * TD + 1, HH = 2 * TH + 1, HW = 17, NEV = 9;
    constexpr int QROW = HW * RB, QVOX = HD * HH * HW, QPIECES = (QVOX * 4 + 63) / 64, BUF = QPIECES * 1024, NQ = QPIECES / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.y * 64 + wv * 16;             // this wave's 16 output channels

    // ---- weights -> registers: packed mode 0 = [tap][cout_p rows][64 k]
    u32x4 wr[27][2];
    {
        const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A.w), 0, 27 * cout_p * CX * 2, 0x00020000);
#pragma unroll
        for (int tp = 0; tp < 27; ++tp)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                wr[tp][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, ((row0 + li) * CX + c * 32 + q * 8) * 2, tp * (cout_p * CX * 2), 0));
    }
    float bia[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bia[r] = A.bias ? A.bias[row0 + q * 4 + r] : 0.f;

    const int q_rowb = A.I[2] * CX * 2, q_slab = A.I[1] * q_rowb;
    uint32_t qsel[NQ];
    int qrel[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int G = (wv + 4 * j) * 64 + lane;
        const int vox = G >> 2;
        const int row = vox / HW, sl = vox - row * HW;
        const int hd = row / HH, hh = row - hd * HH;
        const int jw = sl < NEV ? 2 * sl : 2 * (sl - NEV) + 1;
        qsel[j] = vox < QVOX ? (1u << hd) | (1u << (5 + hh)) | (1u << (14 + jw)) : 0x80000000u;
        qrel[j] = hd * q_slab + hh * q_rowb + jw * (CX * 2) + (((G & 3) ^ (((hh >> 1) & 1) * 2)) * 16);    // bank swizzle, see k_ig3s (here: a 2-way conflict
        //                                                    between the two lattice rows of a 16-point MFMA tile, 2 x 1088 B apart)
    }
    __amdgpu_buffer_rsrc_t qrs0, qrs1;                      // the two channel chunks of the decoded tile's halo
    uint32_t qmask = 0;
    int c_n = 0, c_d = 0, c_h = 0, c_w = 0;
    auto setup = [&]() {
        const int l0d = c_d * TD, l0h = c_h * TH, l0w = c_w * 8;
        const int64_t q_org = (int64_t)c_n * A.I[0] * q_slab + (int64_t)(2 * l0d - 1) * q_slab + (2 * l0h - 1) * q_rowb + (2 * l0w - 1) * (CX * 2);
        qrs0 = ig3s_rsrc(reinterpret_cast<const char*>(A.x) + q_org, 0x7ffffff0);
        qrs1 = ig3s_rsrc(reinterpret_cast<const char*>(A.x) + q_org + 64, 0x7ffffff0);
        auto rng = [](int lo, int hi) -> uint32_t { return (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u); };
        const uint32_t md = rng(l0d == 0 ? 1 : 0, min(HD, A.I[0] - 2 * l0d + 1));
        const uint32_t mh = rng(l0h == 0 ? 1 : 0, min(HH, A.I[1] - 2 * l0h + 1));
        const uint32_t mw = rng(l0w == 0 ? 1 : 0, min(HW, A.I[2] - 2 * l0w + 1));
        qmask = md | (mh << 5) | (mw << 14);
    };
    auto decode = [&](int tile) {
        const int tpn = A.nt[0] * A.nt[1] * A.nt[2];
        c_n = tile / tpn;
        int tt = tile - c_n * tpn;
        c_w = tt % A.nt[2]; tt /= A.nt[2];
        c_h = tt % A.nt[1];
        c_d = tt / A.nt[1];
        setup();
    };
    auto advance = [&]() {
        c_w += A.gw; int cy = c_w >= A.nt[2]; c_w -= cy ? A.nt[2] : 0;
        c_h += A.gh + cy; cy = c_h >= A.nt[1]; c_h -= cy ? A.nt[1] : 0;
        c_d += A.gd + cy; cy = c_d >= A.nt[0]; c_d -= cy ? A.nt[0] : 0;
        c_n += A.gn + cy;
        setup();
    };
    auto dma_piece = [&](int j, int buf, int chunk) {
        const bool ok = (qsel[j] & qmask) == qsel[j];
        ig3s_dma16(chunk ? qrs1 : qrs0, ok ? qrel[j] : (int)0x80000000, (uint32_t)__builtin_amdgcn_readfirstlane(buf * BUF + (wv + 4 * j) * 1024));
    };

    // point tile j = lattice rows 2 j, 2 j + 1 of the tile's 8 rows (plane j >> 1, rows 2 (j & 1) + (li >> 3)), column li & 7
    const int q_lane = ((li >> 3) * 2) * QROW + (li & 7) * RB;
    // part q of the voxel in halo row hh = 2 (li >> 3) + 4 (j & 1) + b sits at position q ^ 2 ((hh >> 1) & 1): the row parity for b = 0 / 1, flipped for b = 2
    const int q_part[2] = {q_lane + ((q ^ (((li >> 3) & 1) * 2)) * 16), q_lane + ((q ^ ((((li >> 3) + 1) & 1) * 2)) * 16)};
    f32x4 acc[4];
    float ssum[4], ssq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[r] = 0.f; ssq[r] = 0.f; }

    // MFMAs of chunk `chunk` out of buffer `buf`; `stage`: the pieces of chunk `nchunk` of the decoded tile go into the other buffer
    auto compute = [&](int buf, int chunk, bool stage, int nchunk) {
        constexpr int U = 27 * 4, QD_ = 4;
        const char* const qb = smem + buf * BUF;
        u32x4 bf[QD_ + 1];
        auto load_b = [&](int u) -> u32x4 {
            const int tp = u >> 2, j = u & 3;
            const int a = tp / 9, b = (tp / 3) % 3, c = tp % 3;
            return *reinterpret_cast<const u32x4*>(qb + q_part[b == 2] + ((2 * (j >> 1) + a) * HH + 4 * (j & 1) + b) * QROW + (c == 1 ? NEV * RB : c == 2 ? RB : 0));
        };
#pragma unroll
        for (int u0 = 0; u0 < QD_; ++u0) bf[u0] = load_b(u0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u + QD_ < U) bf[(u + QD_) % (QD_ + 1)] = load_b(u + QD_);
            if (u < NQ && stage) dma_piece(u, buf ^ 1, nchunk);
            __builtin_amdgcn_sched_barrier(0);
            acc[u & 3] = H16<T>::mma(chunk ? wr[u >> 2][1] : wr[u >> 2][0], bf[u % (QD_ + 1)], acc[u & 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto flush_stats = [&](int n) {
        if constexpr (STATS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = ssum[r], s2 = ssq[r];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); s2 += __shfl_xor(s2, o, 64); }
                if (li == 0) {
                    double* dst = A.stats + (((int64_t)(blockIdx.x % NNDET_STATS_REPLICAS) * A.N + n) * cout_p + row0 + q * 4 + r) * 2;
                    atomicAdd(dst, (double)s); atomicAdd(dst + 1, (double)s2);
                }
                ssum[r] = 0.f; ssq[r] = 0.f;
            }
        }
    };
    int st_n = -1;
    auto epilogue = [&](int n, int td, int th, int tw) {
        if (STATS && n != st_n) { if (st_n >= 0) flush_stats(st_n); st_n = n; }
        uint2 pk[4];
        bool valid[4];
        int64_t vox[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int od = td * TD + (j >> 1), oh = th * TH + 2 * (j & 1) + (li >> 3), ow = tw * 8 + (li & 7);
            valid[j] = od < A.O[0] && oh < A.O[1] && ow < A.O[2];
            vox[j] = (((int64_t)n * A.O[0] + od) * A.O[1] + oh) * A.O[2] + ow;
            pk[j].x = H16<T>::pack2(acc[j][0] + bia[0], acc[j][1] + bia[1]); pk[j].y = H16<T>::pack2(acc[j][2] + bia[2], acc[j][3] + bia[3]);
            if constexpr (STATS) {
                if (valid[j]) {
                    const float r0 = H16<T>::lo(pk[j].x), r1 = H16<T>::hi(pk[j].x), r2 = H16<T>::lo(pk[j].y), r3 = H16<T>::hi(pk[j].y);
                    ssum[0] += r0; ssum[1] += r1; ssum[2] += r2; ssum[3] += r3;
                    ssq[0] += r0 * r0; ssq[1] += r1 * r1; ssq[2] += r2 * r2; ssq[3] += r3 * r3;
                }
            }
        }
        // lane rows q and q ^ 1 hold neighbouring channel quads of the same points: v_permlane16_swap over a PAIR of point tiles gives the
        // even rows the 8 consecutive channels 4 q .. 4 q + 7 of tile j and the odd rows the channels 4 (q - 1) .. 4 q + 3 of tile j + 1:
        // one 16-byte store per lane and tile pair instead of two 8-byte ones
        typedef unsigned int ig3s_v2u __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            const ig3s_v2u sx = __builtin_amdgcn_permlane16_swap(pk[j].x, pk[j + 1].x, false, false);
            const ig3s_v2u sy = __builtin_amdgcn_permlane16_swap(pk[j].y, pk[j + 1].y, false, false);
            const bool odd = q & 1;
            const bool ok = odd ? valid[j + 1] : valid[j];
            T* const yp = reinterpret_cast<T*>(A.y) + (odd ? vox[j + 1] : vox[j]) * cout_p + row0 + (q & 2) * 4;
            if (ok) *reinterpret_cast<u32x4*>(yp) = u32x4{sx[0], sy[0], sx[1], sy[1]};
        }
    };

    const int G = gridDim.x, bx = blockIdx.x;
    auto perm = [&](int cnt) { return (cnt & 7) ? bx : (bx & 7) * (cnt >> 3) + (bx >> 3); };
    const int full = A.total_tiles / G, rest = A.total_tiles - full * G;
    const int pfull = perm(G);
    auto exists = [&](int k) { return k < full || (k == full && bx < rest); };
    auto to_round = [&](int k) {
        if (k < full) { if (k == 0) decode(pfull); else advance(); }
        else if (k == full && bx < rest) decode(full * G + perm(rest));
    };
    if (!exists(0)) return;
    to_round(0);
#pragma unroll
    for (int j = 0; j < NQ; ++j) dma_piece(j, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; exists(k); ++k) {
        const bool nx = exists(k + 1);
        const int e_n = c_n, e_d = c_d, e_h = c_h, e_w = c_w;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        compute(0, 0, true, 1);                               // chunk 0 from buffer 0; chunk 1 of this tile -> buffer 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nx) to_round(k + 1);                              // the next tile's scalars
        compute(1, 1, nx, 0);                                 // chunk 1 from buffer 1; chunk 0 of the next tile -> buffer 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        epilogue(e_n, e_d, e_h, e_w);
    }
    if (STATS && st_n >= 0) flush_stats(st_n);
}

// returns 1 = not covered (the caller goes on to k_igemm)
static int ig3s_on() { return getenv("NNDET_IG3S") ? atoi(getenv("NNDET_IG3S")) : 1; }   // (read per call: tests compare routes that must share their kernels)
int ig3s_covers_pre(const NndetConv* c) {
    if (!ig3s_on() || !c->in_affine || c->transposed || !nndet_is16(c->dtype) || c->cin_p != 32 || c->cout_p != 64 || c->batch <= 0) return 0;
    for (int i = 0; i < 3; ++i) if (c->k[i] != 3 || c->s[i] != 2 || c->p[i] != 1) return 0;
    if ((int64_t)c->in_d * c->in_h * c->in_w * c->cin_p * 2 >= (1LL << 31)) return 0;
    return (int64_t)c->batch * c->out_d * c->out_h * c->out_w * 64 * 2 < (1LL << 31);
}

// xn != NULL (with c->in_affine): x is the pre-norm tensor, the normalised one is written to xn by the same launch (k_ig3s<.., PRE>)
int ig3s_run(const NndetConv* c, int kind, const void* x, const void* w, const float* bias, const void* res, void* y, double* stats, hipStream_t st,
             void* xn) {
    const int on = ig3s_on();
    const bool v1 = c->cin_p == 32 && c->cout_p == 64, v2 = c->cin_p == 64 && c->cout_p % 64 == 0 && c->cout_p <= 256 && on != 3;   // (NNDET_IG3S=3: the 32 -> 64 form only)
    const bool pre = c->in_affine && xn;
    if (!on || kind != 0 || res || c->transposed || (c->in_affine && !pre) || !nndet_is16(c->dtype) || !(v1 || v2)) return 1;
    if (pre && !ig3s_covers_pre(c)) return 1;
    for (int i = 0; i < 3; ++i) if (c->k[i] != 3 || c->s[i] != 2 || c->p[i] != 1) return 1;
    const int64_t xb = (int64_t)c->in_d * c->in_h * c->in_w * c->cin_p * 2;
    if (xb >= (1LL << 31) || c->batch <= 0) return 1;
    Ig3sArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stats; a.N = c->batch;
    a.I[0] = c->in_d; a.I[1] = c->in_h; a.I[2] = c->in_w;
    a.O[0] = c->out_d; a.O[1] = c->out_h; a.O[2] = c->out_w;
    a.nt[0] = ceil_div(a.O[0], 2); a.nt[1] = ceil_div(a.O[1], 4); a.nt[2] = ceil_div(a.O[2], 8);
    a.total_tiles = a.N * a.nt[0] * a.nt[1] * a.nt[2];
    const int wgs = getenv("NNDET_IG3S_WGS") ? atoi(getenv("NNDET_IG3S_WGS")) : 256;      // (read per call: the tests vary it)
    const int ny = v2 ? c->cout_p / 64 : 1;                 // blocks of 64 output channels (k_ig3s2)
    int G = (wgs < 1 || wgs > 256 ? 256 : wgs) / ny;
    if (G < 1) G = 1;
    if (G > a.total_tiles) G = a.total_tiles;
    int g = G;
    a.gw = g % a.nt[2]; g /= a.nt[2];
    a.gh = g % a.nt[1]; g /= a.nt[1];
    a.gd = g % a.nt[0]; a.gn = g / a.nt[0];
    constexpr size_t lds = 2 * 48 * 1024, lds3 = 3 * 48 * 1024;
    static NndetDevOnce at;
    if (at.need()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3s<bf16_t, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3s<bf16_t, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3s<f16_t, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3s<f16_t, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3s<bf16_t, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3s<bf16_t, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3s<f16_t, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3s<f16_t, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3s2<bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3s2<bf16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3s2<f16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig3s2<f16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        at.done();
    }
    if (v2) {
        const dim3 g2(G, ny);
        if (c->dtype == NNDET_F16) { if (stats) k_ig3s2<f16_t, true><<<g2, 256, lds, st>>>(a, c->cout_p); else k_ig3s2<f16_t, false><<<g2, 256, lds, st>>>(a, c->cout_p); }
        else { if (stats) k_ig3s2<bf16_t, true><<<g2, 256, lds, st>>>(a, c->cout_p); else k_ig3s2<bf16_t, false><<<g2, 256, lds, st>>>(a, c->cout_p); }
        LAUNCH_CHECK();
        return 0;
    }
    if (pre) {
        a.ss = c->in_affine; a.xn = xn; a.ss_relu = c->in_relu;
        if (c->dtype == NNDET_F16) { if (stats) k_ig3s<f16_t, true, true><<<G, 256, lds3, st>>>(a); else k_ig3s<f16_t, false, true><<<G, 256, lds3, st>>>(a); }
        else { if (stats) k_ig3s<bf16_t, true, true><<<G, 256, lds3, st>>>(a); else k_ig3s<bf16_t, false, true><<<G, 256, lds3, st>>>(a); }
        LAUNCH_CHECK();
        return 0;
    }
    if (c->dtype == NNDET_F16) { if (stats) k_ig3s<f16_t, true, false><<<G, 256, lds, st>>>(a); else k_ig3s<f16_t, false, false><<<G, 256, lds, st>>>(a); }
    else { if (stats) k_ig3s<bf16_t, true, false><<<G, 256, lds, st>>>(a); else k_ig3s<bf16_t, false, false><<<G, 256, lds, st>>>(a); }
    LAUNCH_CHECK();
    return 0;
}
