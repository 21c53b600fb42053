// Declarations shared by the convolution translation units.
#pragma once
#include "common.h"
#include <string.h>
#include <stdlib.h>

// fp64 (sum, sumsq) statistics are accumulated into this many replicas to spread atomic contention;
// consumers (norm_apply) add the replicas up. Layout [NNDET_STATS_REPLICAS][N][C_p][2].
#define NNDET_STATS_REPLICAS 32

// conv_igemm.hip
int igemm_run(const NndetConv* c, int kind /*0 fwd, 1 bwd-data*/, const void* x, const void* w, const float* bias,
              const void* res, void* y, double* stats, hipStream_t st, float* dbias = nullptr, void* ws = nullptr, size_t ws_bytes = 0,
              const struct DgsNormRed* nr = nullptr);   // nr (kind 1 only): also the norm-backward sums of the block that produced this conv's input
int ig3_fuses_norm_reduce(const NndetConv* c);          // 1 if igemm_run accepts nr for this problem (stride-1 3x3x3 data gradients in k_ig3, 16 bits)
size_t igemm_splitk_bytes(const NndetConv* c, int kind);   // > 0: igemm_run splits K when given that much workspace
// ragged batches (NndetItems): 3x3x3 / stride 1 / pad 1 only
int items_check(const NndetConv* c, const NndetItems* it);
int igemm_items_run(const NndetConv* c, const NndetItems* it, int kind, const void* x, const void* w, const float* bias, void* y,
                    double* stats, hipStream_t st);
int wgrad_items_run(const NndetConv* c, const NndetItems* it, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                    size_t ws_bytes, hipStream_t st);
// conv_pw.hip: 1x1x1 / kernel == stride transposed convolutions streamed without LDS staging; returns 1 = not covered
int pw_run(const NndetConv* c, int kind, const void* x, const void* w, const float* bias, const void* res, void* y, hipStream_t st,
           float* dbias = nullptr);
int pw_covers(const NndetConv* c, int kind);   // 1 if pw_run would take this problem (so a fused dbias is available for kind 1)
// conv_dgs.hip: data gradient of the strided 3x3x3 convolutions, all parity classes from one staged halo; returns 1 = not covered
// nr (optional): also accumulate the norm-backward sums of the block that produced this convolution's input (dx = its complete gradient)
struct DgsNormRed { const void* y; const float* mean_rstd; const float* gamma; const float* beta; double* red_ws; int relu, c; };
int dgs_run(const NndetConv* c, const void* dy, const void* w, const void* res, void* dx, hipStream_t st, const DgsNormRed* nr = nullptr);
int dgs_covers(const NndetConv* c);
int dgs_fuses_norm_reduce(const NndetConv* c);   // 1 if dgs_run accepts nr for this problem
// conv_ig3s.hip: forward of the 32 -> 64 stride-2 3x3x3 transition (LDS-DMA double buffering, weights in registers); returns 1 = not covered
int ig3s_run(const NndetConv* c, int kind, const void* x, const void* w, const float* bias, const void* res, void* y, double* stats, hipStream_t st,
             void* xn = nullptr);   // xn: see nndet_conv3d_forward_norm_input
int ig3s_covers_pre(const NndetConv* c);   // 1 if ig3s_run(.., xn != NULL) takes this problem (in_affine set)
// conv_wgrad.hip
int wgrad_run(const NndetConv* c, const void* x, const void* dy, float* dw, float* dbias, int* bias_done, void* ws, size_t ws_bytes,
              hipStream_t st);   // bias_done = 1: dbias (may be NULL) was accumulated by the weight-gradient kernel itself
size_t wgrad_workspace_bytes(const NndetConv* c);
// conv_stem.hip (Cin_p == 1)
int stem_forward(const NndetConv* c, const void* x, const float* w_f32, const float* bias, void* y, double* stats, int* stats_done,
                 hipStream_t st);   // stats_done = 1: the IN statistics were accumulated by the kernel (stats may be NULL)
int stem_wgrad(const NndetConv* c, const void* x, const void* dy, float* dw, hipStream_t st);
// norm.hip
int colsum_run(int dtype, const void* x, int64_t rows, int c_p, int c, float* out, hipStream_t st);
int norm_stats_run(int dtype, const void* x, int batch, int64_t spatial, int c_p, double* stats, hipStream_t st);
int norm_finalize_run(const double* stats, int N, int c, int c_p, int groups, int64_t spatial, float eps, float* mean_rstd, hipStream_t st);

// ---- deferred input normalisation (NndetConv.in_affine): x' = relu?(x * scale + shift) on one 16-byte piece while it is staged.
// Exactly the arithmetic of k_norm_apply (fmaf, fmaxf, round-to-nearest-even pack), so a consumer that applies the norm on load
// sees bit-identical activations to one that reads the materialised tensor.
template <typename T> struct AffinePiece {       // the 16-bit storage types (bf16_t, f16_t: both sign-magnitude)
    static constexpr int E = 8;
    // 20 VALU ops per piece: 8 unpack, 4 v_pk_fma_f32, 4 v_cvt_pk_bf16_f32, 4 v_pk_max_i16 (ReLU on the packed pair: a
    // negative bf16 / fp16 -- incl. -0 -- is a negative int16; rounding is monotone, so relu(round(x)) == round(relu(x)))
    __device__ static __forceinline__ u32x4 apply(const u32x4 v, const float* sc, const float* sh, int relu) {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2_t x = {H16<T>::lo(v[i]), H16<T>::hi(v[i])};
            const f32x2_t r = __builtin_elementwise_fma(x, f32x2_t{sc[2 * i], sc[2 * i + 1]}, f32x2_t{sh[2 * i], sh[2 * i + 1]});
            uint32_t p = H16<T>::pack2(r[0], r[1]);
            if (relu) p = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), s16x2{0, 0}));
            o[i] = p;
        }
        return o;
    }
};
template <> struct AffinePiece<float> {
    static constexpr int E = 4;
    __device__ static __forceinline__ u32x4 apply(const u32x4 v, const float* sc, const float* sh, int relu) {
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x = fmaf(__uint_as_float(v[i]), sc[i], sh[i]);
            if (relu) x = fmaxf(x, 0.f);
            o[i] = __float_as_uint(x);
        }
        return o;
    }
};
// (scale, shift) pairs of E consecutive channels starting at `ch` of image n: table [N][C_p][2]
template <int E> __device__ __forceinline__ void load_affine(const float* __restrict__ ss, int n, int C_p, int ch, float* sc, float* sh) {
    const float2* p = reinterpret_cast<const float2*>(ss) + (int64_t)n * C_p + ch;
#pragma unroll
    for (int e = 0; e < E; ++e) { const float2 v = p[e]; sc[e] = v.x; sh[e] = v.y; }
}
