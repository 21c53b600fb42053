// v_permlane16_swap_b32 on gfx950: what do the two results hold? (needed to turn two 8-byte epilogue stores into one 16-byte store)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_permlane.hip -o build/probe_permlane
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    const v2u r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    (void)hipMalloc(&d, 512);
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int row = 0; row < 4; ++row) printf("row %d (lanes %2d-%2d): r0 = %3u..%3u   r1 = %3u..%3u\n", row, row * 16, row * 16 + 15, h[row * 16], h[row * 16 + 15], h[64 + row * 16], h[64 + row * 16 + 15]);
    return 0;
}
