// Shared device/host helpers for the gfx950 kernels. Written for CDNA4 only (wave = 64).
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nndet_amd.h"

#define NNDET_WAVE 64

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return (int)_e;           \
    } while (0)

#define LAUNCH_CHECK()                                  \
    do {                                                \
        hipError_t _e = hipGetLastError();              \
        if (_e != hipSuccess) return (int)_e;           \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved), bit-identical to torch's conversion
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x0040u);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// packed conversion of two floats with the hardware instruction v_cvt_pk_bf16_f32 (round-to-nearest-even):
// one VALU op instead of a ~10-instruction software rounding with a NaN branch
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

// ---- fp16 (IEEE half, round-to-nearest-even conversions: what torch.autocast(float16) stores). The reference trains with
// pl.Trainer(precision=16, amp_backend='native') (scripts/train.py:277-278), i.e. exactly this type.
typedef _Float16 f16_t;
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t));
}

// 16-byte vector used for every global <-> LDS <-> register fragment move
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// Everything the kernels need to know about a 16-bit storage type T (bf16_t / f16_t): packed conversion of two floats, the two
// halves of a packed pair as floats, 1.0 twice, the 16x16x32 MFMA. The kernels are written once against H16<T>.
// ---- float64 accumulators of the fp32 (parity) convolution kernels: v_mfma_f64_16x16x4_f64 leaves row q + 4 r of the 16 x 16 result
// in register r of lane (q = lane / 16, li = lane % 16) (tools/probe_mfma64.hip); the epilogues are written for the fp32 MFMA layout,
// row 4 q + r. Rounds to fp32 (the ONE rounding of the exactly accumulated sum) and transposes the 4 x 4 (q, r) index across the four
// 16-lane rows of the wave: destination (q, r) takes register q of lane (r, li). All 64 lanes must be active.
typedef double f64x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 f64acc_rows_to_f32(const f64x4_t& c) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int q = lane >> 4, li = lane & 15;
    const int v0 = __float_as_int((float)c[0]), v1 = __float_as_int((float)c[1]), v2 = __float_as_int((float)c[2]), v3 = __float_as_int((float)c[3]);
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int src = (r * 16 + li) * 4;
        const int g0 = __builtin_amdgcn_ds_bpermute(src, v0), g1 = __builtin_amdgcn_ds_bpermute(src, v1);
        const int g2 = __builtin_amdgcn_ds_bpermute(src, v2), g3 = __builtin_amdgcn_ds_bpermute(src, v3);
        o[r] = __int_as_float(q == 0 ? g0 : (q == 1 ? g1 : (q == 2 ? g2 : g3)));
    }
    return o;
}

template <typename T> struct H16;
template <> struct H16<bf16_t> {
    static constexpr uint32_t ONE2 = 0x3f803f80u;
    __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
    __device__ static __forceinline__ float lo(uint32_t u) { return __uint_as_float(u << 16); }
    __device__ static __forceinline__ float hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
    __device__ static __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    // 32 x 32 x 16: lane l holds A[m = l % 32][k = 8 (l / 32) + 0..7], B[k][n = l % 32]; D register r = row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32
    __device__ static __forceinline__ f32x16_t mma32(const u32x4& a, const u32x4& b, const f32x16_t& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct H16<f16_t> {
    static constexpr uint32_t ONE2 = 0x3c003c00u;
    __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
    __device__ static __forceinline__ float lo(uint32_t u) { return (float)__builtin_bit_cast(f16x2_t, u)[0]; }
    __device__ static __forceinline__ float hi(uint32_t u) { return (float)__builtin_bit_cast(f16x2_t, u)[1]; }
    __device__ static __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ f32x16_t mma32(const u32x4& a, const u32x4& b, const f32x16_t& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct H16<float> {   // never used: keeps discarded `if constexpr (sizeof(T) == 2)` branches well-formed
    static constexpr uint32_t ONE2 = 0;
    __device__ static __forceinline__ uint32_t pack2(float, float) { return 0; }
    __device__ static __forceinline__ float lo(uint32_t) { return 0.f; }
    __device__ static __forceinline__ float hi(uint32_t) { return 0.f; }
    __device__ static __forceinline__ f32x4 mma(const u32x4&, const u32x4&, const f32x4& c) { return c; }
    __device__ static __forceinline__ f32x16_t mma32(const u32x4&, const u32x4&, const f32x16_t& c) { return c; }
};

// Timing experiments that leave work out (the results are WRONG): honoured only together with NNDET_TIMING_EXPERIMENTS=1, announced once.
static inline int nndet_timing_experiment(const char* name) {
    const char* v = getenv(name);
    if (!v || !atoi(v)) return 0;
    const char* gate = getenv("NNDET_TIMING_EXPERIMENTS");
    if (!gate || atoi(gate) != 1) return 0;
    fprintf(stderr, "nndet_amd: timing experiment %s=%s is active -- results of this process are WRONG by design\n", name, v);
    return atoi(v);
}

// run `F<T>` for the activation type a dtype code names (NNDET_F32 / NNDET_BF16 / NNDET_F16)
#define NNDET_DISPATCH_DTYPE(dt, CALL)                                      \
    ((dt) == NNDET_BF16 ? CALL(bf16_t) : (dt) == NNDET_F16 ? CALL(f16_t) : CALL(float))
static inline int nndet_esize(int dt) { return dt == NNDET_F32 ? 4 : 2; }
static inline bool nndet_is16(int dt) { return dt == NNDET_BF16 || dt == NNDET_F16; }

template <typename T> struct Elem;
template <> struct Elem<float> {
    __device__ static __forceinline__ float ld(float v) { return v; }
    __device__ static __forceinline__ float st(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    __device__ static __forceinline__ float ld(bf16_t v) { return bf16_to_f32(v); }
    __device__ static __forceinline__ bf16_t st(float v) { return f32_to_bf16(v); }
};
template <> struct Elem<f16_t> {
    __device__ static __forceinline__ float ld(f16_t v) { return (float)v; }
    __device__ static __forceinline__ f16_t st(float v) { return (f16_t)v; }
};

// Workgroup ids are dealt round-robin to the 8 XCDs (each with its own 4 MiB L2), so spatially adjacent tiles -- whose halos
// overlap -- land on different L2s and every halo is fetched from HBM / Infinity Cache again (measured: 1.9x / 2.3x the
// algorithmic read bytes for k_ig3 / k_wgrad3, profiles/round1_pmc_traffic.json). Remap id b within groups of `group` ids
// (a multiple of 8, ~ the number of co-resident workgroups) so that the ids of one XCD cover a CONTIGUOUS run of tiles.
__device__ __forceinline__ int xcd_compact(int b, int total, int group) {
    const int g0 = (b / group) * group;
    const int n = min(group, total - g0);
    if (n & 7) return b;                       // ragged tail: identity
    const int r = b - g0;
    return g0 + (r & 7) * (n >> 3) + (r >> 3);
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// "Done once per DEVICE" flag for per-kernel attributes (hipFuncAttributeMaxDynamicSharedMemorySize is a property of the function ON A
// DEVICE: a process that drives a second GPU must set it there too). Thread-safe: two threads may both set the attribute
// (idempotent), the bit is published afterwards. Usage: static NndetDevOnce f; if (f.need()) { ...hipFuncSetAttribute...; f.done(); }
#include <atomic>
struct NndetDevOnce {
    std::atomic<unsigned long long> bits[4];
    NndetDevOnce() { for (auto& b : bits) b.store(0ull); }
    static int dev() { int d = 0; return hipGetDevice(&d) == hipSuccess ? (d & 255) : 0; }
    bool need() const { const int d = dev(); return !(bits[d >> 6].load(std::memory_order_acquire) & (1ull << (d & 63))); }
    void done() { const int d = dev(); bits[d >> 6].fetch_or(1ull << (d & 63), std::memory_order_release); }
};
