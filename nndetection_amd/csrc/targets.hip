// Device-side target preparation for gfx950 -- replaces the three transforms every training / validation step runs on the
// GPU before the network (nndet/ptmodule/retinaunet/base.py:141,163):
//   FindInstances          (nndet/io/transforms/instances.py:26-41)   unique instance ids > 0 per image
//   Instances2Boxes        (:44-142)   per instance: nonzero() -> min / max per axis -> box (min - 1, max + 1), class lookup
//   Instances2Segmentation (:207-296)  semantic map: class + 1 on the instance's voxels
// The reference loops over instances in Python with one nonzero() / boolean mask per instance and `.item()` round trips.
// Here: ONE streaming pass over the instance volume (k_inst_reduce: segmented min / max with LDS atomics for small ids and
// the semantic map written on the fly) + one tiny ordered compaction per image (k_inst_finalize). Integer work: bit-exact.
#include "common.h"

#define INST_LDS_IDS 256

// grid (ceil(nvox / (256 * 8)), B), block 256. ext [B][max_id][6] int32: (min d, min h, min w, max d, max h, max w),
// initialised to (INT_MAX x3, -1 x3) by the caller (k_inst_init).
__global__ __launch_bounds__(256) void k_inst_reduce(const float* __restrict__ inst, int64_t nvox, int H, int W,
                                                     const int32_t* __restrict__ cls_table, int max_id,
                                                     float* __restrict__ seg_out, int32_t* __restrict__ ext, int32_t* __restrict__ err) {
    __shared__ int32_t mn[INST_LDS_IDS][3];
    __shared__ int32_t mx[INST_LDS_IDS][3];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < INST_LDS_IDS * 3; i += 256) { (&mn[0][0])[i] = 0x7fffffff; (&mx[0][0])[i] = -1; }
    __syncthreads();
    const float* src = inst + (int64_t)b * nvox;
    float* dst = seg_out + (int64_t)b * nvox;
    const int32_t* cls = cls_table + (int64_t)b * max_id;
    int32_t* eb = ext + (int64_t)b * max_id * 6;
    const int64_t i0 = (int64_t)blockIdx.x * 256 * 8 + threadIdx.x;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int64_t i = i0 + (int64_t)t * 256;
        if (i >= nvox) break;
        const int id = (int)src[i];                       // .to(dtype=torch.int): truncation (instances.py:35)
        float s = 0.f;
        if (id > 0) {
            if (id >= max_id) { atomicOr(err, 1); }
            else {
                const int c = cls[id];
                if (c < 0) atomicOr(err, 2);              // instance without an entry in instance_mapping (KeyError in the reference)
                s = (float)(c + 1);                       // add_background=True (instances.py:228)
                const int w = (int)(i % W);
                const int64_t r = i / W;
                const int h = (int)(r % H), d = (int)(r / H);
                if (id < INST_LDS_IDS) {
                    atomicMin(&mn[id][0], d); atomicMin(&mn[id][1], h); atomicMin(&mn[id][2], w);
                    atomicMax(&mx[id][0], d); atomicMax(&mx[id][1], h); atomicMax(&mx[id][2], w);
                } else {
                    atomicMin(&eb[id * 6 + 0], d); atomicMin(&eb[id * 6 + 1], h); atomicMin(&eb[id * 6 + 2], w);
                    atomicMax(&eb[id * 6 + 3], d); atomicMax(&eb[id * 6 + 4], h); atomicMax(&eb[id * 6 + 5], w);
                }
            }
        }
        dst[i] = s;
    }
    __syncthreads();
    const int lim = max_id < INST_LDS_IDS ? max_id : INST_LDS_IDS;
    for (int i = threadIdx.x; i < lim * 3; i += 256) {
        const int id = i / 3, a = i - id * 3;
        if (mx[id][a] >= 0) {
            atomicMin(&eb[id * 6 + a], mn[id][a]);
            atomicMax(&eb[id * 6 + 3 + a], mx[id][a]);
        }
    }
}

__global__ void k_inst_init(int64_t n6, int32_t* ext, int32_t* err) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *err = 0;
    if (i < n6) ext[i] = (i % 6) < 3 ? 0x7fffffff : -1;
}

// grid (B), block 1024: ids ascending (FindInstances returns them sorted) -> compacted boxes / classes / ids + count
__global__ __launch_bounds__(1024) void k_inst_finalize(const int32_t* __restrict__ ext, const int32_t* __restrict__ cls_table,
                                                        int max_id, float* __restrict__ boxes, int64_t* __restrict__ classes,
                                                        int32_t* __restrict__ ids, int32_t* __restrict__ counts) {
    __shared__ int wsum[16];
    __shared__ int running;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) running = 0;
    __syncthreads();
    const int32_t* eb = ext + (int64_t)b * max_id * 6;
    for (int base = 0; base < max_id; base += 1024) {
        const int id = base + tid;
        const bool ok = id > 0 && id < max_id && eb[id * 6 + 3] >= 0;
        const unsigned long long vote = __ballot(ok);
        const int before = __popcll(vote & ((1ULL << lane) - 1ULL));
        if (lane == 0) wsum[w] = __popcll(vote);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int k = 0; k < 16; ++k) { if (k < w) woff += wsum[k]; tot += wsum[k]; }
        if (ok) {
            const int64_t p = (int64_t)b * max_id + running + woff + before;
            // instances_to_boxes (instances.py:113-121): (x1, y1, x2, y2, z1, z2) = (min0 - 1, min1 - 1, max0 + 1, max1 + 1, min2 - 1, max2 + 1)
            boxes[p * 6 + 0] = (float)(eb[id * 6 + 0] - 1); boxes[p * 6 + 1] = (float)(eb[id * 6 + 1] - 1);
            boxes[p * 6 + 2] = (float)(eb[id * 6 + 3] + 1); boxes[p * 6 + 3] = (float)(eb[id * 6 + 4] + 1);
            boxes[p * 6 + 4] = (float)(eb[id * 6 + 2] - 1); boxes[p * 6 + 5] = (float)(eb[id * 6 + 5] + 1);
            classes[p] = (int64_t)cls_table[(int64_t)b * max_id + id];
            ids[p] = id;
        }
        __syncthreads();
        if (tid == 0) running += tot;
        __syncthreads();
    }
    if (tid == 0) counts[b] = running;
}

extern "C" int nndet_instances_to_targets_f32(const float* inst, int32_t B, int32_t D, int32_t H, int32_t W,
                                              const int32_t* cls_table, int32_t max_id, float* seg_out, int32_t* ext_ws,
                                              float* boxes_out, int64_t* classes_out, int32_t* ids_out, int32_t* counts_out,
                                              int32_t* err_out, void* stream) {
    hipStream_t st = as_stream(stream);
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || max_id <= 1 || max_id > (1 << 20)) return NNDET_EINVAL;
    if (!inst || !cls_table || !seg_out || !ext_ws || !boxes_out || !classes_out || !ids_out || !counts_out || !err_out) return NNDET_EINVAL;
    const int64_t nvox = (int64_t)D * H * W;
    const int64_t n6 = (int64_t)B * max_id * 6;
    k_inst_init<<<(unsigned)ceil_div64(n6, 256), 256, 0, st>>>(n6, ext_ws, err_out);
    LAUNCH_CHECK();
    k_inst_reduce<<<dim3((unsigned)ceil_div64(nvox, 256 * 8), B), 256, 0, st>>>(inst, nvox, H, W, cls_table, max_id, seg_out, ext_ws, err_out);
    LAUNCH_CHECK();
    k_inst_finalize<<<B, 1024, 0, st>>>(ext_ws, cls_table, max_id, boxes_out, classes_out, ids_out, counts_out);
    LAUNCH_CHECK();
    return 0;
}
