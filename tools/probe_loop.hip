// Ceiling probe for the k_ig3 main loop on gfx950: how many cycles per v_mfma_f32_16x16x32_bf16 does a SIMD sustain with
//   V0  MFMAs only (16 independent accumulators per wave)
//   V1  + the 8 ds_read_b128 per tap (half-tap prefetch, pinned with sched_barrier)
//   V2  + the 2 weight buffer loads per tap (L2 resident, 2 taps ahead)
//   V3  + a staging phase (10 buffer loads + 10 ds_write_b128 per thread, 2 barriers) every 27 taps
//   V4  MFMAs only, v_mfma_f32_32x32x16_bf16 (4 accumulators of 16 registers)
// at 1 / 2 / 3 workgroups (of 4 waves) per CU. Build: hipcc --offload-arch=gfx950 -O3 tools/probe_loop.hip -o /tmp/probe_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int V, int MINW, int NT>
__global__ __launch_bounds__(256, MINW) void k_probe(const void* w, const void* x, float* out, int chunks, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, 27 * 64 * 64 * 2, 0x00020000);
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, 1 << 30, 0x00020000);
    const int voff0 = ((lane & 15) * 64 + (lane >> 4) * 8) * 2, voff1 = voff0 + 16 * 64 * 2;
    const int lb = (((lane & 15) >> 3) * 10 + (lane & 7)) * 64 + (lane >> 4) * 16 + (tid >> 7) * 2 * 100 * 64;
    for (int i = tid; i < lds_bytes / 16; i += 256) reinterpret_cast<u32x4*>(smem)[i] = reinterpret_cast<const u32x4*>(x)[blockIdx.x * 4096 + i];
    __syncthreads();
    if (V == 4) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + lb));
        const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + lb + 640));
        for (int c = 0; c < chunks; ++c)
#pragma unroll
            for (int tp = 0; tp < 27; ++tp)
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
        out[blockIdx.x * 256 + tid] = s;
        return;
    }
    constexpr int SPT = NT / 4;   // steps (of 4 point tiles) per tap
    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 af[3][2], bf[2][4];
#pragma unroll
    for (int t = 0; t < 3; ++t) { af[t][0] = *reinterpret_cast<const u32x4*>(smem + lb); af[t][1] = *reinterpret_cast<const u32x4*>(smem + lb + 64); }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[t][j] = *reinterpret_cast<const u32x4*>(smem + lb + j * 1280);
    int goff[10];
#pragma unroll
    for (int s = 0; s < 10; ++s) goff[s] = ((blockIdx.x * 10 + s) * 256 + tid) * 16;

    for (int c = 0; c < chunks; ++c) {
        if (V >= 3) {
            __syncthreads();
            u32x4 v[10];
#pragma unroll
            for (int s = 0; s < 10; ++s) v[s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, goff[s], c * 64, 0));
            if (tid < 240) {
#pragma unroll
                for (int s = 0; s < 10; ++s) *reinterpret_cast<u32x4*>(smem + tid * 16 + s * 3840) = v[s];
            }
            __syncthreads();
        }
        if (V >= 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, voff0, t * 8192 + c * 64, 0));
                af[t][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, voff1, t * 8192 + c * 64, 0));
            }
        }
        auto lds_half = [&](int h, u32x4* dst) {
            const int tp = h / SPT, j0 = (h % SPT) * 4;
            const int a = tp / 9, cc = (tp / 3) % 3, b = tp % 3;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = (j0 + jj) & 7;
                dst[jj] = *reinterpret_cast<const u32x4*>(smem + lb + ((((j >> 2) + a) * 10 + 2 * (j & 3) + b) * 10 + cc) * 64);
            }
        };
        // V5: weights through an LDS ring (3 slots of 4 KB behind the halo): per tap every thread loads ONE 16-byte piece two taps
        // ahead, writes it to the ring one tap ahead, one barrier per tap; A fragments are ds_read_b128
        char* ring = smem + 38400;
        u32x4 wreg[2];
        if (V == 5) {
            wreg[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, tid * 16, 0 * 8192 + c * 64, 0));
            wreg[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, tid * 16, 1 * 8192 + c * 64, 0));
            *reinterpret_cast<u32x4*>(ring + tid * 16) = wreg[0];
        }
        if (V >= 1) lds_half(0, bf[0]);
#pragma unroll
        for (int h = 0; h < 27 * SPT; ++h) {
            const int tp = h / SPT;
            if (V >= 2 && V != 5 && (h % SPT) == 0 && tp + 2 < 27) {
                af[(tp + 2) % 3][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, voff0, (tp + 2) * 8192 + c * 64, 0));
                af[(tp + 2) % 3][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, voff1, (tp + 2) * 8192 + c * 64, 0));
            }
            if (V == 5 && (h % SPT) == 0) {
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): my ring write of the previous tap is done
                __builtin_amdgcn_s_barrier();
                af[tp % 3][0] = *reinterpret_cast<const u32x4*>(ring + (tp % 3) * 4096 + (((lane & 15) + (tid >> 7) * 32) * 64 + (lane >> 4) * 16));
                af[tp % 3][1] = *reinterpret_cast<const u32x4*>(ring + (tp % 3) * 4096 + (((lane & 15) + 16 + (tid >> 7) * 32) * 64 + (lane >> 4) * 16));
                if (tp + 1 < 27) *reinterpret_cast<u32x4*>(ring + ((tp + 1) % 3) * 4096 + tid * 16) = wreg[(tp + 1) & 1];
                if (tp + 2 < 27) wreg[tp & 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, tid * 16, (tp + 2) * 8192 + c * 64, 0));
            }
            if (V >= 1 && h + 1 < 27 * SPT) lds_half(h + 1, bf[(h + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[i][(h % SPT) * 4 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, af[tp % 3][i]), __builtin_bit_cast(bf16x8, bf[h & 1][jj]), acc[i][(h % SPT) * 4 + jj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = s;
}

__global__ void k_fill(uint16_t* p, size_t n) {   // pseudo-random bf16 in (-2, 2): constant data lets the chip clock higher (DVFS)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (uint16_t)((h & 0x8000u) | 0x3f00u | (h & 0xffu));
    }
}

template <int V, int MINW, int NT>
static void run(const char* name, void* w, void* x, float* out, int wg_per_cu) {
    const int chunks = 64, lds = 38400 + 3 * 4096, grid = 256 * wg_per_cu;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe<V, MINW, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_probe<V, MINW, NT><<<grid, 256, lds, 0>>>(w, x, out, 4, lds);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    k_probe<V, MINW, NT><<<grid, 256, lds, 0>>>(w, x, out, chunks, lds);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * chunks * 27 * (V == 4 ? 8 : 2 * NT);           // instructions
    const double flops = mfma * (V == 4 ? 32768.0 : 16384.0);
    // ns per MFMA per SIMD: each SIMD runs wg_per_cu waves
    const double ns_per = ms * 1e6 / ((double)wg_per_cu * chunks * 27 * (V == 4 ? 8 : 2 * NT));
    printf("%-28s wg/cu %d  %8.3f ms  %8.1f TF/s  %6.2f ns per MFMA per SIMD (%.1f cyc @2.4GHz) err=%s\n", name, wg_per_cu, ms, flops / ms / 1e9,
           ns_per, ns_per * 2.4, hipGetErrorString(hipGetLastError()));
}

int main() {
    void *w, *x; float* out;
    hipMalloc(&w, 27 * 64 * 64 * 2 + 4096); k_fill<<<256, 256>>>((uint16_t*)w, 27 * 64 * 64);
    hipMalloc(&x, (size_t)1 << 30); k_fill<<<4096, 256>>>((uint16_t*)x, (size_t)1 << 29);
    hipMalloc(&out, 256 * 3 * 256 * 4);
    for (int o = 1; o <= 3; ++o) {
        run<0, 3, 8>("V0 mfma 16x16x32 only", w, x, out, o);
        run<1, 3, 8>("V1 + lds reads", w, x, out, o);
        run<2, 3, 8>("V2 + weight buffer loads", w, x, out, o);
        run<3, 3, 8>("V3 + staging/barriers", w, x, out, o);
        run<5, 3, 8>("V5 weights via LDS ring", w, x, out, o);
        if (o <= 2) {
            run<1, 2, 16>("V1 NT=16 lds reads", w, x, out, o);
            run<3, 2, 16>("V3 NT=16 wloads+staging", w, x, out, o);
            run<5, 2, 16>("V5 NT=16 LDS ring", w, x, out, o);
        }
        run<4, 3, 8>("V4 mfma 32x32x16 only", w, x, out, o);
    }
    return 0;
}
