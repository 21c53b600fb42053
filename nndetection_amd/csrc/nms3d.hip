// 3D NMS for gfx950 -- replaces nndet._C.nms (nndet/csrc/cuda/nms.cu:99-221).
//
// Pipeline (all on `stream`, no host synchronisation, no D2H of the mask):
//   1. stable descending radix sort of the scores (rocPRIM device primitive; index payload)
//   2. gather boxes in score order
//   3. k_nms_mask:   one wave per 64x64 tile of the UPPER triangle; lane = row box, the 64 column
//                    boxes sit in LDS and are read as broadcasts; word (i, c) bit j <=> IoU(i, 64c+j) > thr
//                    (same bit-matrix as the reference kernel, lower-triangle tiles are never touched)
//   4. greedy scan, blocked in "super chunks" of 64 x 64 = 4096 boxes:
//        k_nms_scan_super  (1 workgroup): resolves the 64 chunks of the super chunk sequentially; the
//                          64-row dependency chain of a chunk is resolved inside ONE wave with readlane
//                          (wave = 64 = bits of a mask word), rows OR-reduced into the removed-words in LDS
//        k_nms_propagate   (whole GPU): ORs the kept rows of the finished super chunk into the removed-words
//                          of all later columns
//   5. k_nms_compact: ordered compaction of the keep bits -> indices into the input order.
// IoU arithmetic is the reference's devIoU_3d (nms.cu:36-51) operation for operation; compile with
// -ffp-contract=off (no FMA contraction) and correctly rounded fp32 division (hipcc default).
#include "common.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

typedef unsigned long long u64;

__device__ __forceinline__ float iou3d(const float* a, const float* b) {
    // (x1, y1, x2, y2, z1, z2). width/height naming of the reference is irrelevant: the first product commutes.
    float bottom = fmaxf(a[0], b[0]), top = fminf(a[2], b[2]);
    float left = fmaxf(a[1], b[1]), right = fminf(a[3], b[3]);
    float front = fmaxf(a[4], b[4]), back = fminf(a[5], b[5]);
    float width = fmaxf(right - left, 0.f), height = fmaxf(top - bottom, 0.f);
    float depth = fmaxf(back - front, 0.f);
    float inter = width * height * depth;
    float sa = (a[2] - a[0]) * (a[3] - a[1]) * (a[5] - a[4]);
    float sb = (b[2] - b[0]) * (b[3] - b[1]) * (b[5] - b[4]);
    return inter / (sa + sb - inter);
}

// iou3d(a, b) > thr without the IEEE division in all but ~1e-6 of the cases, and with EXACTLY its result in all of them:
// inter / un > thr is decided by comparing inter with un * thr * (1 +- 1e-6) -- the two products are rounded (relative error
// 6e-8 each), far inside the 1e-6 margin, while RN(inter / un) can only differ from inter / un by 6e-8 relative, so outside the
// band the rounded quotient is on the same side of thr as the exact one. Inside the band, and for NaN / inf / negative volumes
// (both comparisons false), the division itself decides. The mask kernel was issue-stalled on the dependent division sequence
// for 64 % of its wave cycles (profiles/round2_nms_pmc.txt).
__device__ __forceinline__ bool iou3d_gt(const float* a, const float* b, float thr, float thr_hi, float thr_lo) {
    float bottom = fmaxf(a[0], b[0]), top = fminf(a[2], b[2]);
    float left = fmaxf(a[1], b[1]), right = fminf(a[3], b[3]);
    float front = fmaxf(a[4], b[4]), back = fminf(a[5], b[5]);
    float width = fmaxf(right - left, 0.f), height = fmaxf(top - bottom, 0.f);
    float depth = fmaxf(back - front, 0.f);
    float inter = width * height * depth;
    float sa = (a[2] - a[0]) * (a[3] - a[1]) * (a[5] - a[4]);
    float sb = (b[2] - b[0]) * (b[3] - b[1]) * (b[5] - b[4]);
    const float un = sa + sb - inter;
    if (un > 1e-30f && thr > 0.f) {                  // the bounds below assume positive, normal operands
        if (inter > un * thr_hi) return true;
        if (inter < un * thr_lo) return false;
    }
    return inter / un > thr;
}

__global__ void k_iota(int32_t* v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}

__global__ void k_gather_boxes(const float* __restrict__ boxes, const int32_t* __restrict__ order, int64_t n,
                               float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 6) return;
    int64_t r = i / 6;
    int k = (int)(i - r * 6);
    out[i] = boxes[(int64_t)order[r] * 6 + k];
}

// 2D boxes (x1, y1, x2, y2) -> (x1, y1, x2, y2, 0, 1): the 3D IoU of two such boxes IS devIoU of nndet/csrc/cuda/nms.cu:22-34 bit for bit
// (the third factor of the intersection and of both volumes is exactly 1.0f, and the products keep the reference's order), so the 2D
// kernel of the reference (nms.cu:54-96) needs no second mask kernel here.
__global__ void k_gather_boxes2d(const float* __restrict__ boxes, const int32_t* __restrict__ order, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 b = *reinterpret_cast<const float4*>(boxes + (int64_t)order[i] * 4);
    float* o = out + i * 6;
    o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = 0.f; o[5] = 1.f;
}

// grid (col_blocks, ceil(col_blocks/4)), block 256 = 4 waves; wave w handles row block 4*by + w.
// labels (may be NULL): bit only between boxes of the same label (per-class clustering of wbc3d.hip); nan_hits: a NaN IoU sets the
// bit as well (`!(iou <= thr)`: the box leaves the pool, wbc.py:122,141) -- NMS itself uses `iou > thr` (nms.cu:126).
// blockIdx.z = image of a batch (round 4: the fused post-processing runs the NMS of all images of a batch in the same launches instead
// of one image after the other -- every launch of this file is latency-bound at N <= 10 000): boxes / mask advance by the strides.
__global__ __launch_bounds__(256) void k_nms_mask(const float* __restrict__ boxes, int64_t n, float thr,
                                                  u64* __restrict__ mask, int col_blocks,
                                                  const int32_t* __restrict__ labels = nullptr, int nan_hits = 0,
                                                  int64_t bs_boxes = 0, int64_t bs_mask = 0) {
    __shared__ float cbox[64 * 6];
    __shared__ int32_t clab[64];
    boxes += (int64_t)blockIdx.z * bs_boxes;
    mask += (int64_t)blockIdx.z * bs_mask;
    const int cb = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int rb = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (cb < (int)blockIdx.y * 4) return;  // the whole block is below the diagonal (uniform exit)
    const int col_size = (int)min((int64_t)64, n - (int64_t)cb * 64);
    for (int i = threadIdx.x; i < col_size * 6; i += 256) cbox[i] = boxes[(int64_t)cb * 64 * 6 + i];
    if (labels && (int)threadIdx.x < col_size) clab[threadIdx.x] = labels[(int64_t)cb * 64 + threadIdx.x];
    __syncthreads();
    if (rb > cb || rb >= col_blocks) return;
    const int64_t row = (int64_t)rb * 64 + lane;
    if (row >= n) return;
    float a[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k] = boxes[row * 6 + k];
    u64 t = 0;
    // Diagonal tiles carry BOTH directions (bit j of row i for every j != i): the in-wave resolution of k_nms_scan_super needs
    // the incoming edges of a box (higher-scored boxes of its chunk that overlap it) = the low bits of its own row, because
    // iou3d(a, b) == iou3d(b, a) bit for bit (min / max / the two additions commute, the products have the same order).
    const int skip = (rb == cb) ? lane : -1;
    if (!labels && !nan_hits) {
        const float thr_hi = (float)((double)thr * (1.0 + 1e-6)), thr_lo = (float)((double)thr * (1.0 - 1e-6));
        for (int j = 0; j < col_size; ++j) {
            if (j != skip && iou3d_gt(a, cbox + j * 6, thr, thr_hi, thr_lo)) t |= 1ULL << j;
        }
    } else {
        const int32_t la = labels ? labels[row] : 0;
        for (int j = 0; j < col_size; ++j) {
            const float v = iou3d(a, cbox + j * 6);
            const bool hit = nan_hits ? !(v <= thr) : (v > thr);
            if (j != skip && hit && (!labels || clab[j] == la)) t |= 1ULL << j;
        }
    }
    mask[row * col_blocks + cb] = t;
}

// One workgroup of 1024 threads resolves the chunks [c0, c1) (c1 - c0 <= 64) of a super chunk sequentially.
// Round 1 fetched the mask rows of the KEPT boxes of a chunk from global memory after its keep bits were known: one exposed L2
// round trip per chunk, 2.6 us x 64 chunks = 168 us per super chunk = 75 % of the whole NMS at N = 10 000
// (profiles/round2_nms_kernel_stats.txt). Now the rows of ALL 64 boxes of a chunk (64 rows x <= 64 words of this super chunk =
// 32 KiB) are prefetched speculatively into a ring of 4 LDS buffers three chunks ahead (global -> registers another three
// iterations before the LDS write), so the dependent part of an iteration only touches LDS: the 64-row readlane chain on the diagonal word
// (wave 0), then the OR of the kept rows into the removed-words of the later chunks.
#define NMS_SCAN_NBUF 4
// Workgroup barrier that orders LDS traffic only. __syncthreads() also drains the vector-memory counter (s_waitcnt vmcnt(0)),
// which would wait for the speculative row loads that are deliberately left in flight across several iterations.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__global__ __launch_bounds__(1024) void k_nms_scan_super(const u64* __restrict__ mask, int64_t n, int col_blocks,
                                                         int c0, int c1, u64* __restrict__ remv,
                                                         u64* __restrict__ keepbits, int64_t bs_mask = 0) {
    extern __shared__ __attribute__((aligned(16))) char nms_smem[];
    mask += (int64_t)blockIdx.x * bs_mask;                           // blockIdx.x = image of a batch (one workgroup each)
    remv += (int64_t)blockIdx.x * col_blocks;
    keepbits += (int64_t)blockIdx.x * col_blocks;
    u64* rows = reinterpret_cast<u64*>(nms_smem);                    // [NMS_SCAN_NBUF][64 rows][64 words]
    __shared__ u64 remv_l[64];
    __shared__ u64 keep_l;
    const int tid = threadIdx.x;
    const int nc = c1 - c0;
    if (tid < nc) remv_l[tid] = remv[c0 + tid];
    // thread -> (row r of the chunk, words jj, jj + 16, jj + 32, jj + 48 of the super chunk)
    const int r = tid >> 4, jj = tid & 15;
    u64 reg[3][4];                                // global -> register loads run THREE chunks ahead of their LDS write
    auto issue = [&](int ci, u64* dst) {          // chunk index ci relative to c0; words before the diagonal are never written by k_nms_mask
        const int64_t row = (int64_t)(c0 + ci) * 64 + r;
        const bool okr = ci < nc && row < n;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = jj + 16 * k;
            dst[k] = (okr && j >= ci && j < nc) ? mask[row * col_blocks + c0 + j] : 0ULL;
        }
    };
    auto commit = [&](int ci, const u64* src) {
        u64* b = rows + (size_t)(ci % NMS_SCAN_NBUF) * 4096 + r * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) b[jj + 16 * k] = src[k];
    };
    // prologue: chunks 0, 1, 2 into the ring, chunks 3, 4, 5 in flight (chunk k travels through reg[k % 3])
    issue(0, reg[0]); issue(1, reg[1]); issue(2, reg[2]);
    commit(0, reg[0]); commit(1, reg[1]); commit(2, reg[2]);
    issue(3, reg[0]); issue(4, reg[1]); issue(5, reg[2]);
    __syncthreads();
    auto step = [&](int ci, u64* rg) {
        const u64* buf = rows + (size_t)(ci % NMS_SCAN_NBUF) * 4096;
        if (tid < 64) {
            // Greedy NMS inside the chunk = the lexicographically first maximal independent set of its 64 x 64 overlap graph.
            // Instead of walking the 64 boxes one by one (2.6 us per chunk in round 1), every lane decides its own box as soon
            // as all higher-scored overlapping boxes of the chunk are decided: kept if none of them is kept, suppressed if one
            // is. A round is two ballots; the number of rounds is the longest dependency chain (1-4 for detector outputs, 64
            // at worst -- identical result either way).
            const u64 d = buf[tid * 64 + ci];                       // row tid of the diagonal tile: all overlaps inside the chunk
            const u64 in = d & ((1ULL << tid) - 1ULL);              // higher-scored boxes of the chunk that overlap this one
            const int valid = (int)min((int64_t)64, n - (int64_t)(c0 + ci) * 64);
            const u64 vmask = valid >= 64 ? ~0ULL : ((1ULL << valid) - 1ULL);
            u64 S = remv_l[ci] | ~vmask;                            // suppressed from outside (earlier chunks) or beyond n
            u64 K = 0;
            for (int round = 0; round < 64; ++round) {
                const u64 U = ~(K | S);
                if (U == 0) break;                                  // uniform
                const bool und = (U >> tid) & 1ULL;
                const bool now_s = und && (in & K) != 0ULL;
                const bool now_k = und && !now_s && (in & U) == 0ULL;
                S |= __ballot(now_s);
                K |= __ballot(now_k);
            }
            if (tid == 0) {
                keepbits[c0 + ci] = K;
                keep_l = K;
            }
        }
        lds_barrier();
        const u64 keep = keep_l;
        // OR the kept rows of this chunk into the removed-words of the later chunks of the super chunk (LDS only)
        if ((keep >> r) & 1ULL) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = jj + 16 * k;
                if (j > ci && j < nc) {
                    const u64 v = buf[r * 64 + j];
                    if (v) atomicOr(&remv_l[j], v);
                }
            }
        }
        // ring maintenance: the registers issued three iterations ago (chunk ci + 3) go to the slot chunk ci - 1 used; the loads of
        // chunk ci + 6 start in the same registers
        commit(ci + 3, rg);
        lds_barrier();
        issue(ci + 6, rg);
    };
    // unrolled by 3 so that every register set is addressed statically: the wait before a commit is then vmcnt(8) (the two
    // younger sets stay in flight) instead of a full drain
    for (int ci = 0; ci < nc; ci += 3) {
        step(ci, reg[0]);
        if (ci + 1 < nc) step(ci + 1, reg[1]);
        if (ci + 2 < nc) step(ci + 2, reg[2]);
    }
}

// grid (ceil((col_blocks - c1)/256), c1 - c0), block 256: thread = one later column word, block row = chunk.
__global__ __launch_bounds__(256) void k_nms_propagate(const u64* __restrict__ mask, int col_blocks, int c0, int c1,
                                                       const u64* __restrict__ keepbits, u64* __restrict__ remv, int64_t bs_mask = 0) {
    mask += (int64_t)blockIdx.z * bs_mask;                           // blockIdx.z = image of a batch
    keepbits += (int64_t)blockIdx.z * col_blocks;
    remv += (int64_t)blockIdx.z * col_blocks;
    const int c = c0 + blockIdx.y;
    const int j = c1 + blockIdx.x * 256 + threadIdx.x;
    const u64 keep = keepbits[c];
    if (j >= col_blocks || keep == 0) return;
    u64 v = 0;
    for (int r = 0; r < 64; ++r) {
        if ((keep >> r) & 1ULL) v |= mask[((int64_t)c * 64 + r) * col_blocks + j];
    }
    if (v) atomicOr(&remv[j], v);
}

// single block of 1024 threads: ordered compaction
// order == NULL: identity (the caller's boxes already are in score order). n_valid (device, may be NULL): only rows below
// *n_valid are real boxes, the rest is padding up to the launch capacity (never reported).
__global__ __launch_bounds__(1024) void k_nms_compact(const u64* __restrict__ keepbits, int col_blocks,
                                                      const int32_t* __restrict__ order, int64_t* __restrict__ keep_out,
                                                      int64_t* __restrict__ n_keep, const int64_t* __restrict__ n_valid,
                                                      int64_t bs_keep = 0) {
    __shared__ int wsum[16];
    __shared__ int running;
    keepbits += (int64_t)blockIdx.x * col_blocks;                    // blockIdx.x = image of a batch (order == NULL then)
    keep_out += (int64_t)blockIdx.x * bs_keep;
    n_keep += blockIdx.x;
    if (n_valid) n_valid += blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) running = 0;
    __syncthreads();
    const int64_t nv = n_valid ? *n_valid : ((int64_t)col_blocks * 64);
    for (int base = 0; base < col_blocks; base += 1024) {
        const int idx = base + tid;
        u64 bits = (idx < col_blocks) ? keepbits[idx] : 0ULL;
        const int64_t left = nv - (int64_t)idx * 64;
        if (left <= 0) bits = 0ULL;
        else if (left < 64) bits &= (1ULL << left) - 1ULL;
        const int cnt = __popcll(bits);
        int incl = cnt;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        int pos = running + woff + incl - cnt;
        u64 b = bits;
        while (b) {
            const int r = __ffsll((long long)b) - 1;
            b &= b - 1;
            keep_out[pos++] = order ? (int64_t)order[(int64_t)idx * 64 + r] : (int64_t)idx * 64 + r;
        }
        __syncthreads();
        if (tid == 1023) running += woff + incl;
        __syncthreads();
    }
    if (tid == 0) *n_keep = running;
}

static const size_t NMS_SCAN_LDS = (size_t)NMS_SCAN_NBUF * 64 * 64 * 8;
static int nms_scan_attr() {
    static NndetDevOnce done;
    if (done.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nms_scan_super), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NMS_SCAN_LDS);
        if (e != hipSuccess) return (int)e;
        done.done();
    }
    return 0;
}

struct NmsWs {
    float* sboxes;
    float* keys_out;
    int32_t* vals_in;
    int32_t* order;
    u64* mask;
    u64* remv;
    u64* keepbits;
    void* sort_tmp;
    size_t sort_tmp_bytes;
    size_t total;
};

static int nms_layout(int64_t n, char* base, NmsWs* ws) {
    const int64_t cb = ceil_div64(n, 64);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    size_t o_sb = take((size_t)n * 6 * 4), o_ko = take((size_t)n * 4), o_vi = take((size_t)n * 4), o_or = take((size_t)n * 4);
    size_t o_mask = take((size_t)n * cb * 8), o_remv = take((size_t)cb * 8), o_keep = take((size_t)cb * 8);
    size_t tmp = 0;
    hipError_t e = rocprim::radix_sort_pairs_desc<rocprim::default_config, const float*, float*, const int32_t*, int32_t*>(
        nullptr, tmp, nullptr, nullptr, nullptr, nullptr, (size_t)n, 0, 32, (hipStream_t)0, false);
    if (e != hipSuccess) return (int)e;
    size_t o_tmp = take(tmp);
    ws->sort_tmp_bytes = tmp; ws->total = off;
    if (!base) return 0;                       // size query: no pointer arithmetic on a null base (UBSan, round 6)
    ws->sboxes = (float*)(base + o_sb); ws->keys_out = (float*)(base + o_ko);
    ws->vals_in = (int32_t*)(base + o_vi); ws->order = (int32_t*)(base + o_or);
    ws->mask = (u64*)(base + o_mask); ws->remv = (u64*)(base + o_remv); ws->keepbits = (u64*)(base + o_keep);
    ws->sort_tmp = base + o_tmp; ws->sort_tmp_bytes = tmp; ws->total = off;
    return 0;
}

// order == NULL: `boxes` already are in score order (no gather); n_valid: see k_nms_compact
static int nms_core(const float* boxes, const int32_t* order, int64_t n, float thr, int64_t* keep_out,
                    int64_t* n_keep_out, NmsWs& ws, hipStream_t st, const int64_t* n_valid = nullptr, int box_dim = 6) {
    const int cb = (int)ceil_div64(n, 64);
    { const int arc = nms_scan_attr(); if (arc) return arc; }
    const float* sboxes = boxes;
    if (order && box_dim == 4) {
        k_gather_boxes2d<<<(unsigned)ceil_div64(n, 256), 256, 0, st>>>(boxes, order, n, ws.sboxes);
        LAUNCH_CHECK();
        sboxes = ws.sboxes;
    } else if (order) {
        k_gather_boxes<<<(unsigned)ceil_div64(n * 6, 256), 256, 0, st>>>(boxes, order, n, ws.sboxes);
        LAUNCH_CHECK();
        sboxes = ws.sboxes;
    }
    HIP_TRY(hipMemsetAsync(ws.remv, 0, (size_t)cb * 8, st));
    HIP_TRY(hipMemsetAsync(keep_out, 0xFF, (size_t)n * 8, st));
    k_nms_mask<<<dim3(cb, ceil_div(cb, 4)), 256, 0, st>>>(sboxes, n, thr, ws.mask, cb);
    LAUNCH_CHECK();
    for (int c0 = 0; c0 < cb; c0 += 64) {
        const int c1 = c0 + 64 < cb ? c0 + 64 : cb;
        k_nms_scan_super<<<1, 1024, NMS_SCAN_LDS, st>>>(ws.mask, n, cb, c0, c1, ws.remv, ws.keepbits);
        LAUNCH_CHECK();
        if (c1 < cb) {
            k_nms_propagate<<<dim3(ceil_div(cb - c1, 256), c1 - c0), 256, 0, st>>>(ws.mask, cb, c0, c1, ws.keepbits, ws.remv);
            LAUNCH_CHECK();
        }
    }
    k_nms_compact<<<1, 1024, 0, st>>>(ws.keepbits, cb, order, keep_out, n_keep_out, n_valid);
    LAUNCH_CHECK();
    return 0;
}

extern "C" size_t nndet_nms3d_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    NmsWs ws;
    if (nms_layout(n, nullptr, &ws) != 0) return 0;
    return ws.total;
}

extern "C" int nndet_nms3d_f32(const float* boxes, const float* scores, int64_t n, float thr, int64_t* keep_out,
                               int64_t* n_keep_out, void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = as_stream(stream);
    if (n < 0 || !n_keep_out) return NNDET_EINVAL;
    if (n == 0) return (int)hipMemsetAsync(n_keep_out, 0, 8, st);
    if (!boxes || !scores || !keep_out || !workspace) return NNDET_EINVAL;
    NmsWs ws;
    int rc = nms_layout(n, (char*)workspace, &ws);
    if (rc) return rc;
    if (ws.total > workspace_bytes) return NNDET_EWORKSPACE;
    k_iota<<<(unsigned)ceil_div64(n, 256), 256, 0, st>>>(ws.vals_in, n);
    LAUNCH_CHECK();
    size_t tmp = ws.sort_tmp_bytes;
    HIP_TRY((rocprim::radix_sort_pairs_desc<rocprim::default_config, const float*, float*, const int32_t*, int32_t*>(
        ws.sort_tmp, tmp, scores, ws.keys_out, ws.vals_in, ws.order, (size_t)n, 0, 32, st, false)));
    return nms_core(boxes, ws.order, n, thr, keep_out, n_keep_out, ws, st);
}

// 2D boxes [n, 4] (x1, y1, x2, y2): nndet._C.nms accepts them too (nms_kernel / devIoU, nndet/csrc/cuda/nms.cu:22-34,54-96; the Python
// dispatcher routes 2D to torchvision.ops.nms, nndet/core/boxes/nms.py:70-72 -- same rule: suppress iff IoU > thr, descending score order).
extern "C" int nndet_nms2d_f32(const float* boxes, const float* scores, int64_t n, float thr, int64_t* keep_out,
                               int64_t* n_keep_out, void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = as_stream(stream);
    if (n < 0 || !n_keep_out) return NNDET_EINVAL;
    if (n == 0) return (int)hipMemsetAsync(n_keep_out, 0, 8, st);
    if (!boxes || !scores || !keep_out || !workspace) return NNDET_EINVAL;
    NmsWs ws;
    int rc = nms_layout(n, (char*)workspace, &ws);
    if (rc) return rc;
    if (ws.total > workspace_bytes) return NNDET_EWORKSPACE;
    k_iota<<<(unsigned)ceil_div64(n, 256), 256, 0, st>>>(ws.vals_in, n);
    LAUNCH_CHECK();
    size_t tmp = ws.sort_tmp_bytes;
    HIP_TRY((rocprim::radix_sort_pairs_desc<rocprim::default_config, const float*, float*, const int32_t*, int32_t*>(
        ws.sort_tmp, tmp, scores, ws.keys_out, ws.vals_in, ws.order, (size_t)n, 0, 32, st, false)));
    return nms_core(boxes, ws.order, n, thr, keep_out, n_keep_out, ws, st, nullptr, 4);
}

extern "C" int nndet_nms3d_sorted_f32(const float* boxes, const int32_t* order, int64_t n, float thr,
                                      int64_t* keep_out, int64_t* n_keep_out, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    hipStream_t st = as_stream(stream);
    if (n < 0 || !n_keep_out) return NNDET_EINVAL;
    if (n == 0) return (int)hipMemsetAsync(n_keep_out, 0, 8, st);
    if (!boxes || !order || !keep_out || !workspace) return NNDET_EINVAL;
    NmsWs ws;
    int rc = nms_layout(n, (char*)workspace, &ws);
    if (rc) return rc;
    if (ws.total > workspace_bytes) return NNDET_EWORKSPACE;
    return nms_core(boxes, order, n, thr, keep_out, n_keep_out, ws, st);
}

// Used by the fused post-processing front end (postproc.hip): boxes [n_cap, 6] already in score order, of which only the
// first *n_valid_dev (device) are real -- the rest must be all-zero rows (IoU 0 / NaN against everything: they never
// suppress and are masked out of the result). keep_out [n_cap] / n_keep_out as nndet_nms3d_f32.
size_t nms_presorted_workspace_bytes(int64_t n_cap) { return nndet_nms3d_workspace_bytes(n_cap); }
int nms_presorted_run(const float* boxes, int64_t n_cap, const int64_t* n_valid_dev, float thr, int64_t* keep_out,
                      int64_t* n_keep_out, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (n_cap <= 0 || !boxes || !keep_out || !n_keep_out || !workspace) return NNDET_EINVAL;
    NmsWs ws;
    int rc = nms_layout(n_cap, (char*)workspace, &ws);
    if (rc) return rc;
    if (ws.total > workspace_bytes) return NNDET_EWORKSPACE;
    return nms_core(boxes, nullptr, n_cap, thr, keep_out, n_keep_out, ws, st, n_valid_dev);
}

// The same for the B images of a batch at once: boxes [B, n_cap, 6] (each image in score order, zero rows behind its *n_valid),
// n_valid_dev / n_keep_out [B], keep_out [B, n_cap]. Image = an extra grid dimension of every launch; workspace B x the single one.
size_t nms_presorted_batched_workspace_bytes(int64_t n_cap, int B) {
    if (n_cap <= 0 || B <= 0) return 256;
    const int64_t cb = ceil_div64(n_cap, 64);
    return align_up((size_t)B * n_cap * cb * 8, 256) + 2 * align_up((size_t)B * cb * 8, 256);
}
int nms_presorted_batched_run(const float* boxes, int64_t n_cap, int B, const int64_t* n_valid_dev, float thr, int64_t* keep_out,
                              int64_t* n_keep_out, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (n_cap <= 0 || B <= 0 || B > 65535 || !boxes || !keep_out || !n_keep_out || !workspace || !n_valid_dev) return NNDET_EINVAL;
    if (nms_presorted_batched_workspace_bytes(n_cap, B) > workspace_bytes) return NNDET_EWORKSPACE;
    const int cb = (int)ceil_div64(n_cap, 64);
    { const int arc = nms_scan_attr(); if (arc) return arc; }
    char* base = (char*)workspace;
    u64* mask = (u64*)base;
    u64* remv = (u64*)(base + align_up((size_t)B * n_cap * cb * 8, 256));
    u64* keepbits = (u64*)((char*)remv + align_up((size_t)B * cb * 8, 256));
    const int64_t bs_mask = n_cap * cb;
    HIP_TRY(hipMemsetAsync(remv, 0, (size_t)B * cb * 8, st));
    HIP_TRY(hipMemsetAsync(keep_out, 0xFF, (size_t)B * n_cap * 8, st));
    k_nms_mask<<<dim3(cb, ceil_div(cb, 4), B), 256, 0, st>>>(boxes, n_cap, thr, mask, cb, nullptr, 0, n_cap * 6, bs_mask);
    LAUNCH_CHECK();
    for (int c0 = 0; c0 < cb; c0 += 64) {
        const int c1 = c0 + 64 < cb ? c0 + 64 : cb;
        k_nms_scan_super<<<B, 1024, NMS_SCAN_LDS, st>>>(mask, n_cap, cb, c0, c1, remv, keepbits, bs_mask);
        LAUNCH_CHECK();
        if (c1 < cb) {
            k_nms_propagate<<<dim3(ceil_div(cb - c1, 256), c1 - c0, B), 256, 0, st>>>(mask, cb, c0, c1, keepbits, remv, bs_mask);
            LAUNCH_CHECK();
        }
    }
    k_nms_compact<<<B, 1024, 0, st>>>(keepbits, cb, nullptr, keep_out, n_keep_out, n_valid_dev, n_cap);
    LAUNCH_CHECK();
    return 0;
}

// Mask + greedy scan only (wbc3d.hip): sboxes [n, 6] in descending score order, slabels (may be NULL) their labels. Returns the
// device pointers of the bit mask [n][ceil(n/64)] and of the head bits (inside `workspace`, valid until it is reused).
int nms_heads_run(const float* sboxes, const int32_t* slabels, int64_t n, float thr, u64** mask_out, u64** keepbits_out,
                  void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (n <= 0 || !sboxes || !workspace || !mask_out || !keepbits_out) return NNDET_EINVAL;
    NmsWs ws;
    int rc = nms_layout(n, (char*)workspace, &ws);
    if (rc) return rc;
    if (ws.total > workspace_bytes) return NNDET_EWORKSPACE;
    const int cb = (int)ceil_div64(n, 64);
    { const int arc = nms_scan_attr(); if (arc) return arc; }
    HIP_TRY(hipMemsetAsync(ws.remv, 0, (size_t)cb * 8, st));
    k_nms_mask<<<dim3(cb, ceil_div(cb, 4)), 256, 0, st>>>(sboxes, n, thr, ws.mask, cb, slabels, 1);
    LAUNCH_CHECK();
    for (int c0 = 0; c0 < cb; c0 += 64) {
        const int c1 = c0 + 64 < cb ? c0 + 64 : cb;
        k_nms_scan_super<<<1, 1024, NMS_SCAN_LDS, st>>>(ws.mask, n, cb, c0, c1, ws.remv, ws.keepbits);
        LAUNCH_CHECK();
        if (c1 < cb) {
            k_nms_propagate<<<dim3(ceil_div(cb - c1, 256), c1 - c0), 256, 0, st>>>(ws.mask, cb, c0, c1, ws.keepbits, ws.remv);
            LAUNCH_CHECK();
        }
    }
    *mask_out = ws.mask; *keepbits_out = ws.keepbits;
    return 0;
}
