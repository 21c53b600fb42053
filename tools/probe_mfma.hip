// Standalone probe (GPU box): empirically prints the lane -> (row, col, k) mappings the kernels rely on.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ uint16_t f2bf(float f) { return (uint16_t)(__float_as_uint(f) >> 16); }

// A[i][k] = 1 only at (i0,k0), B[k][j] = 1 only at (k0', j0): find which C register/lane lights up.
__global__ void k_bf16(const uint16_t* a /*[64][8]*/, const uint16_t* b, float* c /*[64][4]*/) {
    int l = threadIdx.x;
    u32x4 ua, ub;
    for (int r = 0; r < 4; ++r) { ua[r] = a[l * 8 + 2 * r] | ((uint32_t)a[l * 8 + 2 * r + 1] << 16); ub[r] = b[l * 8 + 2 * r] | ((uint32_t)b[l * 8 + 2 * r + 1] << 16); }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) c[l * 4 + r] = acc[r];
}
__global__ void k_f32(const float* a, const float* b, float* c) {
    int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) c[l * 4 + r] = acc[r];
}
// ds_read_b64_tr_b16 semantics: LDS holds element value = its own index; every lane reads from its natural address (lane*8 bytes)
__global__ void k_tr(uint32_t* out /*[64][2]*/) {
    __shared__ uint16_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    uint32_t addr = (uint32_t)(uintptr_t)lds + threadIdx.x * 8;
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 2] = (uint32_t)v; out[threadIdx.x * 2 + 1] = (uint32_t)(v >> 32);
}

int main() {
    // ---- bf16 16x16x32: consistency test = random GEMM vs CPU with the layouts the kernels assume:
    //      A lane l: row l&15, k = 8*(l>>4)+j ; B lane l: col l&15, k = 8*(l>>4)+j ; C lane l reg r: col l&15, row 4*(l>>4)+r
    std::vector<uint16_t> ha(512), hb(512); std::vector<float> A(16 * 32), B(32 * 16), C(256, 0), hc(256);
    srand(1);
    for (int i = 0; i < 16 * 32; ++i) { A[i] = (float)(rand() % 7 - 3); B[i] = (float)(rand() % 5 - 2); }
    auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) { int k = 8 * (l >> 4) + j; ha[l * 8 + j] = bf(A[(l & 15) * 32 + k]); hb[l * 8 + j] = bf(B[k * 16 + (l & 15)]); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 16 + j]; C[i * 16 + j] = s; }
    uint16_t *da, *db; float* dc;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dc, 1024);
    hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
    k_bf16<<<1, 64>>>(da, db, dc); hipMemcpy(hc.data(), dc, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hc[l * 4 + r] != C[(4 * (l >> 4) + r) * 16 + (l & 15)]) ++bad;
    printf("PROBE bf16 16x16x32 layout (A row=l&15,k=8q+j | B col=l&15,k=8q+j | C col=l&15,row=4q+r): %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    // ---- f32 16x16x4
    std::vector<float> fa(64), fb(64), A4(16 * 4), B4(4 * 16);
    for (int i = 0; i < 64; ++i) { A4[i] = (float)(rand() % 7 - 3); B4[i] = (float)(rand() % 5 - 2); }
    for (int l = 0; l < 64; ++l) { fa[l] = A4[(l & 15) * 4 + (l >> 4)]; fb[l] = B4[(l >> 4) * 16 + (l & 15)]; }
    float *dfa, *dfb; hipMalloc(&dfa, 256); hipMalloc(&dfb, 256);
    hipMemcpy(dfa, fa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dfb, fb.data(), 256, hipMemcpyHostToDevice);
    k_f32<<<1, 64>>>(dfa, dfb, dc); hipMemcpy(hc.data(), dc, 1024, hipMemcpyDeviceToHost);
    bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { float s = 0; int i = 4 * (l >> 4) + r, j = l & 15; for (int k = 0; k < 4; ++k) s += A4[i * 4 + k] * B4[k * 16 + j]; if (hc[l * 4 + r] != s) ++bad; }
    printf("PROBE f32 16x16x4 layout (A[l&15][l>>4], B[l>>4][l&15], C col=l&15,row=4q+r): %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    // ---- ds_read_b64_tr_b16
    uint32_t* dout; hipMalloc(&dout, 512); std::vector<uint32_t> ho(128);
    k_tr<<<1, 64>>>(dout); hipMemcpy(ho.data(), dout, 512, hipMemcpyDeviceToHost);
    printf("PROBE ds_read_b64_tr_b16 (lane: 4 element indices read when lane address = lane*8 bytes):\n");
    for (int l = 0; l < 64; ++l) { printf("  l%02d: %4u %4u %4u %4u%s", l, ho[2 * l] & 0xffff, ho[2 * l] >> 16, ho[2 * l + 1] & 0xffff, ho[2 * l + 1] >> 16, (l % 4 == 3) ? "\n" : " |"); }
    hipError_t e = hipDeviceSynchronize();
    printf("PROBE done: %s\n", hipGetErrorString(e));
    return 0;
}
