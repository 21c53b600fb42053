/* Driver for an AddressSanitizer + UndefinedBehaviorSanitizer build of oracle/nms_ref.c (tests/test_oracle_sanitizers.py): the CPU side
 * is where sanitizers can run (GPU ASAN / XNACK are not available on the pool). Exercises the sizes the parity tests use it at -- empty,
 * one box, exact duplicates, degenerate (zero-volume) boxes, a few thousand random boxes -- with EXACTLY sized buffers, so that an
 * out-of-bounds access of the checker itself cannot hide behind slack. Prints a checksum of the kept indices. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int64_t oracle_nms3d(const float* boxes, const int64_t* order, int64_t n, float thr, int64_t* keep);

static uint32_t rng = 12345u;
static float frand(void) { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) / 16777216.0f; }

static uint64_t run(int64_t n, int mode, float thr) {
    float* boxes = (float*)malloc((size_t)(n > 0 ? n : 1) * 6 * sizeof(float));
    int64_t* order = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    int64_t* keep = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) {
        float c[3] = {frand() * 64.f, frand() * 64.f, frand() * 32.f}, s[3] = {1.f + frand() * 12.f, 1.f + frand() * 12.f, 1.f + frand() * 8.f};
        if (mode == 1 && i > 0) { for (int k = 0; k < 6; ++k) boxes[i * 6 + k] = boxes[k]; order[i] = i; continue; }    /* exact duplicates */
        if (mode == 2 && (i & 3) == 0) s[0] = 0.f;                                                                         /* zero-volume boxes: 0 / 0 */
        boxes[i * 6 + 0] = c[0]; boxes[i * 6 + 1] = c[1]; boxes[i * 6 + 2] = c[0] + s[0]; boxes[i * 6 + 3] = c[1] + s[1];
        boxes[i * 6 + 4] = c[2]; boxes[i * 6 + 5] = c[2] + s[2];
        order[i] = n - 1 - i;                                                                                              /* any permutation is a valid order */
    }
    const int64_t nk = oracle_nms3d(boxes, order, n, thr, keep);
    uint64_t h = (uint64_t)nk * 1000003u;
    for (int64_t i = 0; i < nk; ++i) {
        if (keep[i] < 0 || keep[i] >= n) { fprintf(stderr, "kept index out of range\n"); exit(3); }
        h = h * 31u + (uint64_t)keep[i];
    }
    free(boxes); free(order); free(keep);
    return h;
}

int main(void) {
    uint64_t h = 0;
    const int64_t sizes[] = {0, 1, 2, 63, 64, 65, 1000, 4097};
    for (unsigned s = 0; s < sizeof(sizes) / sizeof(sizes[0]); ++s)
        for (int mode = 0; mode < 3; ++mode) {
            h ^= run(sizes[s], mode, 0.5f);
            h ^= run(sizes[s], mode, 0.0f) << 1;
            h ^= run(sizes[s], mode, 1.0f) << 2;
        }
    printf("ok %llu\n", (unsigned long long)h);
    return 0;
}
