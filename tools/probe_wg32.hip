// What does one ds_read_b64_tr_b16 pair (or one ds_read_b128) per v_mfma_f32_32x32x16_bf16 cost a wave on gfx950?
// (round 4: k_wgrad3d's MFMA phase runs a lone wave at 47 cycles per MFMA instead of 32 -- is it the LDS feed?)
//   P0  MFMA only, 7 accumulators in turn
//   P1  + 2 ds_read_b64_tr_b16 per MFMA, prefetched 3 MFMAs ahead (k_wgrad3d's stream)
//   P2  + 1 ds_read_b128 per MFMA instead
//   P3  + 2 ds_read_b64 (plain) per MFMA instead
//   P4  as P1 with an s_setprio 3 around... (not used)
// at 1 and 2 waves per SIMD. Build: hipcc --offload-arch=gfx950 -O3 tools/probe_wg32.hip -o /tmp/probe_wg32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <int V>
__global__ __launch_bounds__(256, 2) void k_probe(const uint32_t* x, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = x[i];
    __syncthreads();
    const int li = lane & 15, q = lane >> 4;
    const char* base = smem + (q >> 1) * 640 + (q & 1) * 32 + (li >> 2) * 64 + (li & 3) * 8 + (tid >> 6) * 6400;
    const char* base128 = smem + lane * 16 + (tid >> 6) * 6400;
    f32x16 acc[7];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    auto ld = [&](int u) -> u32x4 {
        const int off = (u % 28) * 640 + ((u / 28) & 1) * 64;
        if (V == 1) {
            const uint2 a = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off)));
            const uint2 b = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off + 256)));
            return u32x4{a.x, a.y, b.x, b.y};
        } else if (V == 2) {
            return *reinterpret_cast<const u32x4*>(base128 + off);
        } else {
            const uint2 a = *reinterpret_cast<const uint2*>(base + off), b = *reinterpret_cast<const uint2*>(base + off + 256);
            return u32x4{a.x, a.y, b.x, b.y};
        }
    };
    u32x4 pf = *reinterpret_cast<const u32x4*>(base128);
    constexpr int U = 56, QD = 3;
    for (int it = 0; it < iters; ++it) {
        u32x4 qf[QD + 1];
        if (V >= 1) {
#pragma unroll
            for (int u = 0; u < QD; ++u) qf[u] = ld(u);
        } else {
#pragma unroll
            for (int u = 0; u <= QD; ++u) qf[u] = pf;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (V >= 1 && u + QD < U) qf[(u + QD) % (QD + 1)] = ld(u + QD);
            __builtin_amdgcn_sched_barrier(0);
            acc[u % 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pf), __builtin_bit_cast(bf16x8, qf[u % (QD + 1)]), acc[u % 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 7; ++t) s += acc[t][0] + acc[t][9];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V> static void run(const char* name, const uint32_t* x, float* out, int wg_per_cu) {
    const int iters = 400, grid = 256 * wg_per_cu;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_probe<V><<<grid, 256, 65536, 0>>>(x, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    k_probe<V><<<grid, 256, 65536, 0>>>(x, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * iters * 56;
    const double ns_per = ms * 1e6 / ((double)wg_per_cu * iters * 56);
    printf("%-44s waves/SIMD %d  %8.3f ms  %8.1f TF/s  %6.2f ns per MFMA per SIMD  err=%s\n", name, wg_per_cu, ms, mfma * 32768.0 / ms / 1e9, ns_per,
           hipGetErrorString(hipGetLastError()));
}

__global__ void k_fill(uint32_t* p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (h & 0x80ff80ffu) | 0x3f003f00u;
    }
}

int main() {
    uint32_t* x; float* out;
    hipMalloc(&x, 65536); k_fill<<<64, 256>>>(x, 16384);
    hipMalloc(&out, 512 * 256 * 4);
    for (int o = 1; o <= 2; ++o) {
        run<0>("P0 mfma 32x32x16 only", x, out, o);
        run<1>("P1 + 2 ds_read_b64_tr_b16 per MFMA", x, out, o);
        run<2>("P2 + 1 ds_read_b128 per MFMA", x, out, o);
        run<3>("P3 + 2 ds_read_b64 per MFMA", x, out, o);
    }
    return 0;
}
