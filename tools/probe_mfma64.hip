// Layout probe for v_mfma_f64_16x16x4_f64 on gfx950 (round 4: the fp32 parity kernels accumulate in float64).
// A[i][k] and B[k][j] are given one double per lane under the hypothesis (i or j = lane % 16, k = lane / 16); D = A * B is decoded
// from products of distinct primes-like codes, and both candidate D layouts are tested.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void k(const double* A, const double* B, double* out) {   // A [16][4], B [4][16] row-major; out [64][4]
    const int lane = threadIdx.x;
    const double a = A[(lane % 16) * 4 + lane / 16];
    const double b = B[(lane / 16) * 16 + lane % 16];
    f64x4 c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

int main() {
    double A[64], B[64], D[256], out[256];
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) A[i * 4 + kk] = sin(1.0 + i * 4 + kk) + 0.25 * kk;
    for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) B[kk * 16 + j] = cos(2.0 + kk * 16 + j) - 0.5 * kk;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += A[i * 4 + kk] * B[kk * 16 + j]; D[i * 16 + j] = s; }
    double *dA, *dB, *dO;
    hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dO, sizeof(out));
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dO);
    hipMemcpy(out, dO, sizeof(out), hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
        const int j = lane % 16;
        const int i1 = 4 * (lane / 16) + r, i2 = (lane / 16) + 4 * r;
        e1 = fmax(e1, fabs(out[lane * 4 + r] - D[i1 * 16 + j]));
        e2 = fmax(e2, fabs(out[lane * 4 + r] - D[i2 * 16 + j]));
    }
    printf("mfma_f64_16x16x4: max |err| under H1 (row = 4 * (lane / 16) + r): %.3e ; under H2 (row = lane / 16 + 4 * r): %.3e\n", e1, e2);
    printf("lane 0: %.6f %.6f %.6f %.6f  (D[0][0] %.6f D[1][0] %.6f D[4][0] %.6f)\n", out[0], out[1], out[2], out[3], D[0], D[16], D[64]);
    printf("lane 16: %.6f %.6f (D[4][0] %.6f D[1][0] %.6f)\n", out[64], out[65], D[64], D[16]);
    return 0;
}
