// Probe for the semantics of buffer_load_dwordx4 ... offen lds on gfx950 that the persistent 3x3x3 kernel relies on:
//   (1) LDS destination = M0 + inst_offset + lane * 16 for LDS addresses above 64 KB (M0 wider than 16 bits?)
//   (2) lanes whose buffer offset is out of range write ZEROS to LDS (= the conv padding) instead of leaving LDS untouched
//   (3) a per-lane permuted source lands lane-linear
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_ldsdma.hip -o build/probe_ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(256, 1) void k_probe(const uint32_t* src, int nbytes, uint32_t* out, uint32_t lds_base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // fill LDS with a pattern so that "untouched" is distinguishable from "zero"
    for (int i = tid; i < 160 * 1024 / 4; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, nbytes, 0x00020000);
    // wave w: lanes read piece (63 - lane) of block w; lanes 5, 17 out of range (0x80000000), lane 40 just past num_records
    int voff = (wv * 64 + (63 - lane)) * 16;
    if (lane == 5 || lane == 17) voff = (int)0x80000000;
    if (lane == 40) voff = nbytes;
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;     // LDS byte address of the dynamic segment (0 when there is no static LDS)
    const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + lds_base + wv * 1024);
    dma16(rs, voff, 0, dst);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 4096 / 4; i += 256) out[i] = reinterpret_cast<uint32_t*>(smem + lds_base)[i];
    if (tid == 0) out[1024] = smem_base;
}

int main() {
    const int n = 4 * 64 * 4;   // dwords: 4 waves x 64 pieces x 4
    std::vector<uint32_t> h(n);
    for (int i = 0; i < n; ++i) h[i] = 0x1000u + i;
    uint32_t *src, *out;
    hipMalloc(&src, n * 4 + 4096);
    hipMalloc(&out, 8192);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (uint32_t base : {0u, 65536u, 131072u, 155648u}) {
        k_probe<<<1, 256, 160 * 1024, 0>>>(src, n * 4, out, base);
        hipError_t e = hipDeviceSynchronize();
        std::vector<uint32_t> r(1025);
        hipMemcpy(r.data(), out, 1025 * 4, hipMemcpyDeviceToHost);
        int ok = 0, zero_oob = 0, untouched_oob = 0, bad = 0;
        for (int w = 0; w < 4; ++w)
            for (int l = 0; l < 64; ++l)
                for (int k = 0; k < 4; ++k) {
                    const uint32_t got = r[(w * 64 + l) * 4 + k];
                    const bool oob = (l == 5 || l == 17 || l == 40);
                    if (oob) {
                        if (got == 0) ++zero_oob; else if (got == 0xdeadbeefu) ++untouched_oob; else ++bad;
                    } else {
                        const uint32_t want = 0x1000u + (w * 64 + (63 - l)) * 4 + k;
                        if (got == want) ++ok; else { if (bad < 4) printf("  w%d l%d k%d got %08x want %08x\n", w, l, k, got, want); ++bad; }
                    }
                }
        printf("lds_base %6u smem_base %u: in-range ok %d/%d, oob lanes: zero %d untouched %d, bad %d, err=%s\n", base, r[1024], ok, 4 * 61 * 4,
               zero_oob, untouched_oob, bad, hipGetErrorString(e));
    }
    return 0;
}
