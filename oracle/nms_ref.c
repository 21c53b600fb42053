/* CPU oracle: greedy 3D NMS, plain C. TEST INFRASTRUCTURE (checker + cpu_baseline), never shipped.
 *
 * Restates the reference's algorithm: IoU of devIoU_3d (nndet/csrc/cuda/nms.cu:36-51) and the greedy
 * order of the host loop of nms_cuda (nms.cu:203-215) == nms_cpu (nndet/core/boxes/nms.py:31-53):
 * boxes are visited by decreasing score; a visited, not yet suppressed box is kept and suppresses every
 * later box with IoU > thr. Unlike nms_cpu it needs no [N,N] matrix, so N = 100 000 is feasible
 * (SURVEY.md 8d config 5). Compiled with -ffp-contract=off: same fp32 operations as the GPU kernel.
 *
 * order:  indices sorted by decreasing score (stable), supplied by the caller.
 * returns the number of kept boxes; keep[] receives indices into the input order.
 */
#include <stdint.h>
#include <stdlib.h>

static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline float fminf_(float a, float b) { return a < b ? a : b; }

static inline float iou3d(const float* a, const float* b) {
    float bottom = fmaxf_(a[0], b[0]), top = fminf_(a[2], b[2]);
    float left = fmaxf_(a[1], b[1]), right = fminf_(a[3], b[3]);
    float front = fmaxf_(a[4], b[4]), back = fminf_(a[5], b[5]);
    float width = fmaxf_(right - left, 0.f), height = fmaxf_(top - bottom, 0.f), depth = fmaxf_(back - front, 0.f);
    float inter = width * height * depth;
    float sa = (a[2] - a[0]) * (a[3] - a[1]) * (a[5] - a[4]);
    float sb = (b[2] - b[0]) * (b[3] - b[1]) * (b[5] - b[4]);
    return inter / (sa + sb - inter);
}

int64_t oracle_nms3d(const float* boxes, const int64_t* order, int64_t n, float thr, int64_t* keep) {
    unsigned char* dead = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
    int64_t nk = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (dead[i]) continue;
        const float* a = boxes + order[i] * 6;
        keep[nk++] = order[i];
        for (int64_t j = i + 1; j < n; ++j) {
            if (!dead[j] && iou3d(a, boxes + order[j] * 6) > thr) dead[j] = 1;
        }
    }
    free(dead);
    return nk;
}
