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
              const void* res, void* y, double* stats, hipStream_t st);
// conv_pw.hip: 1x1x1 / kernel == stride transposed convolutions streamed without LDS staging; returns 1 = not covered
int pw_run(const NndetConv* c, int kind, const void* x, const void* w, const float* bias, const void* res, void* y, hipStream_t st);
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
