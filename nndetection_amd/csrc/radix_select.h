// Shared pieces of the exact MSB-first radix SELECT on 64-bit keys (8-bit digits) used by the ATSS matcher (k nearest
// anchors per GT and level), the post-processing front end (top-k scores per image) and the hard-negative sampler (pool of
// the highest-scoring negatives). A select pass is: histogram kernel (LDS histogram per workgroup of the digit at `shift`
// over the keys that still match the prefix, flushed with global atomics) -> pick (one WAVE per problem, below).
#pragma once
#include "common.h"

typedef unsigned long long u64;

// Order-preserving map float -> uint32 (ascending): negative values are bit-flipped, non-negative get the sign bit set.
__device__ __forceinline__ uint32_t f32_sortable(float v) {
    const uint32_t b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float f32_unsortable(uint32_t u) {
    return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu));
}

// One wave per problem: given the 256-bin histogram `h` of the current digit among the keys matching `prefix`, choose the
// bin holding the krem-th smallest key, append the digit to the prefix, reduce krem to the rank inside the bin and clear
// the histogram. Lane = 4 consecutive bins; wave-wide inclusive scan with shuffles.
// `done` (optional): set to 1 when the chosen bin is taken WHOLE (its count equals the remaining rank): the lower digits cannot change
// the selection any more, the prefix gets all-ones below `shift` (= the largest key of the bin) and the caller's remaining histogram
// passes may return at once. Typical for keys whose high 32 bits are (nearly) unique -- scores, hashes -- over 32 bits of index.
__device__ __forceinline__ void radix_pick_wave(u64* prefix, int* krem, unsigned* h, int shift, int lane, int* done = nullptr) {
    const uint4 c = reinterpret_cast<const uint4*>(h)[lane];
    const unsigned mine = c.x + c.y + c.z + c.w;
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const unsigned rem = (unsigned)*krem;
    const unsigned excl = incl - mine;
    // the lane whose bins contain the rem-th smallest element: excl < rem <= incl
    const bool owner = (excl < rem) && (rem <= incl);
    const unsigned long long vote = __ballot(owner);
    if (vote == 0ULL) {                       // cannot happen when k <= problem size; keep the state consistent anyway
        if (lane == 63) *prefix |= ((u64)255) << shift;
    } else if (owner) {
        unsigned cum = excl;
        int b = 0;
        const unsigned cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            if (cum + cc[k2] >= rem) { b = k2; break; }
            cum += cc[k2];
            b = k2 + 1;
        }
        if (b > 3) b = 3;
        u64 add = ((u64)(lane * 4 + b)) << shift;
        if (done && shift > 0 && rem - cum == cc[b]) {
            add |= (((u64)1) << shift) - 1;
            *done = 1;
        }
        *prefix |= add;
        *krem = (int)(rem - cum);
    }
    reinterpret_cast<uint4*>(h)[lane] = make_uint4(0u, 0u, 0u, 0u);
}
