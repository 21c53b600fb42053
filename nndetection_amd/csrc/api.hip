// C-ABI dispatch for the convolution entry points + library identification.
#include "common.h"
#include "conv_common.h"

extern "C" const char* nndet_version(void) { return "nndetection_amd 0.3.0 (round 3: f32 | bf16 | f16)"; }
extern "C" const char* nndet_arch(void) { return "gfx950"; }

// A HIP stream restricted to a set of compute units (hipExtStreamCreateWithCUMask): torch cannot create one, the host side wraps the
// handle in torch.cuda.ExternalStream. Used for the A/B of the weight-gradient stream on a CU partition (NNDET_WGRAD_CUMASK).
extern "C" int nndet_stream_create_cumask(const uint32_t* cu_mask, int32_t words, void** stream_out) {
    if (!cu_mask || words <= 0 || words > 32 || !stream_out) return NNDET_EINVAL;
    hipStream_t st = nullptr;
    HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, cu_mask));
    *stream_out = st;
    return 0;
}
extern "C" int nndet_stream_destroy(void* stream) {
    if (!stream) return NNDET_EINVAL;
    HIP_TRY(hipStreamDestroy(as_stream(stream)));
    return 0;
}

static int check_conv(const NndetConv* c) {
    if (!c) return NNDET_EINVAL;
    if (c->dtype != NNDET_F32 && c->dtype != NNDET_BF16 && c->dtype != NNDET_F16) return NNDET_EINVAL;
    if (c->batch <= 0 || c->cin <= 0 || c->cout <= 0 || c->cin > c->cin_p || c->cout > c->cout_p) return NNDET_EINVAL;
    if (c->cout_p % 32) return NNDET_EINVAL;
    if (c->cin_p != 1 && c->cin_p % 32) return NNDET_EINVAL;
    return 0;
}

extern "C" int nndet_conv3d_forward(const NndetConv* c, const void* x, const void* w, const float* bias, const void* residual,
                                    void* y, double* stats, void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!x || !w || !y) return NNDET_EINVAL;
    hipStream_t st = as_stream(stream);
    if (c->cin_p == 1) {
        if (residual || c->in_affine) return NNDET_EINVAL;
        int stats_done = 0;
        rc = stem_forward(c, x, (const float*)w, bias, y, stats, &stats_done, st);
        if (rc) return rc;
        if (stats && !stats_done) {
            const int64_t spatial = (int64_t)c->out_d * c->out_h * c->out_w;
            return norm_stats_run(c->dtype, y, c->batch, spatial, c->cout_p, stats, st);
        }
        return 0;
    }
    return igemm_run(c, 0, x, w, bias, residual, y, stats, st);
}

extern "C" int32_t nndet_conv3d_forward_norm_input_fused(const NndetConv* c) {
    if (!c || check_conv(c) || !c->in_affine) return 0;
    return ig3s_covers_pre(c);
}

extern "C" int nndet_conv3d_forward_norm_input(const NndetConv* c, const void* x_pre, void* x_norm, const void* w, const float* bias,
                                               void* y, double* stats, void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!x_pre || !x_norm || !w || !y || !c->in_affine || c->transposed || c->cin_p == 1 || x_pre == x_norm) return NNDET_EINVAL;
    hipStream_t st = as_stream(stream);
    if (ig3s_covers_pre(c)) {
        rc = ig3s_run(c, 0, x_pre, w, bias, nullptr, y, stats, st, x_norm);
        if (rc != 1) return rc;
    }
    const int64_t spatial = (int64_t)c->in_d * c->in_h * c->in_w;
    rc = nndet_affine_apply(c->dtype, x_pre, c->in_affine, c->batch, spatial, c->cin_p, c->in_relu, x_norm, stream);
    if (rc) return rc;
    NndetConv plain = *c;
    plain.in_affine = nullptr; plain.in_relu = 0;
    return igemm_run(&plain, 0, x_norm, w, bias, nullptr, y, stats, st);
}

extern "C" int nndet_conv3d_backward_data(const NndetConv* c, const void* dy, const void* w, void* dx, void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!dy || !w || !dx || c->cin_p == 1) return NNDET_EINVAL;
    return igemm_run(c, 1, dy, w, nullptr, nullptr, dx, nullptr, as_stream(stream));
}

extern "C" size_t nndet_conv3d_splitk_workspace_bytes(const NndetConv* c, int32_t kind) {
    if (check_conv(c) || c->cin_p == 1 || (kind != 0 && kind != 1)) return 0;
    return igemm_splitk_bytes(c, kind);
}

extern "C" int nndet_conv3d_forward_ws(const NndetConv* c, const void* x, const void* w, const float* bias, const void* residual,
                                       void* y, double* stats, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!x || !w || !y || c->cin_p == 1) return NNDET_EINVAL;
    return igemm_run(c, 0, x, w, bias, residual, y, stats, as_stream(stream), nullptr, ws, ws_bytes);
}

extern "C" int nndet_conv3d_backward_data_ws(const NndetConv* c, const void* dy, const void* w, void* dx, void* ws, size_t ws_bytes,
                                             void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!dy || !w || !dx || c->cin_p == 1) return NNDET_EINVAL;
    return igemm_run(c, 1, dy, w, nullptr, nullptr, dx, nullptr, as_stream(stream), nullptr, ws, ws_bytes);
}

extern "C" int32_t nndet_conv3d_dgrad_fuses_bias(const NndetConv* c) {
    if (check_conv(c) || c->cin_p == 1) return 0;
    static const int on = getenv("NNDET_PW_BIAS") ? atoi(getenv("NNDET_PW_BIAS")) : 1;
    return on && pw_covers(c, 1) ? 1 : 0;
}

extern "C" int nndet_conv3d_backward_data_bias(const NndetConv* c, const void* dy, const void* w, void* dx, float* dbias, void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!dy || !w || !dx || !dbias || c->cin_p == 1) return NNDET_EINVAL;
    if (!nndet_conv3d_dgrad_fuses_bias(c)) return NNDET_EINVAL;
    return igemm_run(c, 1, dy, w, nullptr, nullptr, dx, nullptr, as_stream(stream), dbias);
}

extern "C" int nndet_conv3d_backward_data_acc(const NndetConv* c, const void* dy, const void* w, void* dx, float* dbias, void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!dy || !w || !dx || c->cin_p == 1 || c->transposed) return NNDET_EINVAL;
    if (dbias && !nndet_conv3d_dgrad_fuses_bias(c)) return NNDET_EINVAL;
    // every kernel reads `res` at exactly the element it then writes: in place is safe (residual == output buffer)
    return igemm_run(c, 1, dy, w, nullptr, dx, dx, nullptr, as_stream(stream), dbias);
}

extern "C" int nndet_conv3d_dgrad_fuses_norm_reduce(const NndetConv* c) {
    if (check_conv(c) || c->cin_p == 1 || c->transposed) return 0;
    return dgs_fuses_norm_reduce(c);
}

extern "C" int32_t nndet_conv3d_dgrad_normred_supported(const NndetConv* c) {
    if (check_conv(c) || c->cin_p == 1 || c->transposed) return 0;
    return ig3_fuses_norm_reduce(c);
}

extern "C" int nndet_conv3d_backward_data_normred(const NndetConv* c, const void* dy, const void* w, void* dx, const void* y_norm,
                                                  const float* mean_rstd, const float* gamma, const float* beta, int32_t relu,
                                                  int32_t c_norm, double* red_ws, void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!dy || !w || !dx || !y_norm || !mean_rstd || !gamma || !beta || !red_ws || c->cin_p == 1 || c->transposed) return NNDET_EINVAL;
    if (c_norm <= 0 || c_norm > c->cin_p || !ig3_fuses_norm_reduce(c)) return NNDET_EINVAL;
    const DgsNormRed nr = {y_norm, mean_rstd, gamma, beta, red_ws, relu ? 1 : 0, c_norm};
    return igemm_run(c, 1, dy, w, nullptr, nullptr, dx, nullptr, as_stream(stream), nullptr, nullptr, 0, &nr);
}

extern "C" int nndet_conv3d_backward_data_acc_normred(const NndetConv* c, const void* dy, const void* w, void* dx, const void* y_norm,
                                                      const float* mean_rstd, const float* gamma, const float* beta, int32_t relu,
                                                      int32_t c_norm, double* red_ws, void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!dy || !w || !dx || !y_norm || !mean_rstd || !gamma || !beta || !red_ws || c->cin_p == 1 || c->transposed) return NNDET_EINVAL;
    if (c_norm <= 0 || c_norm > c->cin_p || !dgs_fuses_norm_reduce(c)) return NNDET_EINVAL;
    const DgsNormRed nr = {y_norm, mean_rstd, gamma, beta, red_ws, relu ? 1 : 0, c_norm};
    return dgs_run(c, dy, w, dx, dx, as_stream(stream), &nr);
}

extern "C" size_t nndet_conv3d_wgrad_workspace_bytes(const NndetConv* c) {
    if (check_conv(c)) return 0;
    return c->cin_p == 1 ? 256 : wgrad_workspace_bytes(c);
}

extern "C" int nndet_conv3d_backward_weight(const NndetConv* c, const void* x, const void* dy, float* dw, float* dbias,
                                            void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!x || !dy || !dw) return NNDET_EINVAL;
    hipStream_t st = as_stream(stream);
    int bias_done = 0;
    rc = (c->cin_p == 1) ? stem_wgrad(c, x, dy, dw, st) : wgrad_run(c, x, dy, dw, dbias, &bias_done, workspace, workspace_bytes, st);
    if (rc) return rc;
    if (dbias && !bias_done) {
        const int64_t rows = (int64_t)c->batch * c->out_d * c->out_h * c->out_w;
        rc = colsum_run(c->dtype, dy, rows, c->cout_p, c->cout, dbias, st);
    }
    return rc;
}

// ---- ragged batches (NndetItems): the pyramid levels through the shared detection-head convolutions in one launch
extern "C" int nndet_conv3d_forward_items(const NndetConv* c, const NndetItems* items, const void* x, const void* w, const float* bias,
                                          void* y, double* stats, void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!items || !x || !w || !y || c->cin_p == 1) return NNDET_EINVAL;
    return igemm_items_run(c, items, 0, x, w, bias, y, stats, as_stream(stream));
}

extern "C" int nndet_conv3d_backward_data_items(const NndetConv* c, const NndetItems* items, const void* dy, const void* w, void* dx,
                                                void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!items || !dy || !w || !dx || c->cin_p == 1) return NNDET_EINVAL;
    return igemm_items_run(c, items, 1, dy, w, nullptr, dx, nullptr, as_stream(stream));
}

extern "C" int nndet_conv3d_backward_weight_items(const NndetConv* c, const NndetItems* items, const void* x, const void* dy, float* dw,
                                                  float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv(c);
    if (rc) return rc;
    if (!items || !x || !dy || !dw || c->cin_p == 1) return NNDET_EINVAL;
    return wgrad_items_run(c, items, x, dy, dw, dbias, workspace, workspace_bytes, as_stream(stream));
}
