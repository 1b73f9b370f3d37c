// api.cpp -- the C ABI of libobia_hip.so (include/obia_hip.h): argument checking and orchestration of
// the kernels for the single-raster operators B1 (SLIC) and B2 (zonal statistics).
#include "slic.hpp"

#include <cmath>

using namespace obia;

namespace {

int check_ctx(obia_ctx *ctx) {
    if (!ctx) { set_error("null context"); return OBIA_E_INVALID; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", ctx->device); return OBIA_E_HIP; }
    return OBIA_OK;
}

int check_slic_args(const float *img, int H, int W, int C, const obia_slic_params *p) {
    if (!img || !p) { set_error("null image or params"); return OBIA_E_INVALID; }
    if (H <= 0 || W <= 0 || C <= 0) { set_error("bad image shape (%d,%d,%d)", H, W, C); return OBIA_E_INVALID; }
    if ((long long)H * W > 0x7fffffffLL) { set_error("rasters above 2^31 pixels must go through the tiled driver"); return OBIA_E_INVALID; }
    if (C > 16) { set_error("more than 16 bands not supported (got %d)", C); return OBIA_E_UNSUPPORTED; }
    if (p->start_label != 0 && p->start_label != 1) { set_error("start_label should be 0 or 1."); return OBIA_E_INVALID; }
    if (p->convert2lab == 1 && C != 3) { set_error("Lab colorspace conversion requires a RGB image."); return OBIA_E_INVALID; }
    if (!(p->compactness > 0.0)) { set_error("compactness must be positive"); return OBIA_E_INVALID; }
    if (p->n_segments <= 0) { set_error("n_segments must be positive"); return OBIA_E_INVALID; }
    if (p->max_num_iter < 0) { set_error("max_num_iter must be >= 0"); return OBIA_E_INVALID; }
    for (int i = 0; i < 3; ++i) {
        if (!(p->sigma_zyx[i] >= 0.0)) { set_error("sigma must be >= 0"); return OBIA_E_INVALID; }
        if (!(p->spacing_zyx[i] > 0.0) || !(p->spacing_zyx[i] < 1.0e30)) { set_error("spacing must be positive and finite"); return OBIA_E_INVALID; }
    }
    return OBIA_OK;
}

// Whole-raster SLIC up to the pre-connectivity labels (device pointers).  On return b holds the plan.
int slic_single(obia_ctx *ctx, const float *img, int H, int W, int C, const uint8_t *mask,
                const obia_slic_params *p, SlicBatch &b, const ExternalSeeds *ext = nullptr) {
    Arena &A = ctx->arena;
    b.nprob = 1;
    b.C = C;
    b.CP = (C + 3) & ~3;
    b.masked = mask != nullptr;
    b.start_label = p->start_label;
    b.max_iter = p->max_num_iter;
    b.exit_on_fixed_point = p->exit_on_fixed_point != 0;
    b.slic_zero = p->slic_zero != 0;
    for (int i = 0; i < 3; ++i) { b.sigma[i] = p->sigma_zyx[i]; b.spacing[i] = p->spacing_zyx[i]; }
    const bool direct = (float)b.spacing[1] != 1.0f || (float)b.spacing[2] != 1.0f;   // anisotropic spacing: the direct sweep path (slic_sweep.hip)
    if (direct) b.exit_on_fixed_point = false;
    b.total_pix = (long long)H * W;
    SlicProblem P{};
    P.H = H; P.W = W; P.pix_off = 0; P.feat_off = 0; P.XB = feat_xb(W); P.fb_off = 0;
    b.probs.assign(1, P);
    b.windows.assign(1, SrcWindow{0, 0, H, W, 0, 0, 0});
    b.d_windows = A.get<SrcWindow>(1);
    b.total_feat_f4 = feat_block_f4(H, W, b.CP);
    b.d_feat = A.get<float>(4 * (size_t)b.total_feat_f4);
    b.d_labels = A.get<int32_t>((size_t)b.total_pix);
    if (!b.d_windows || !b.d_feat || !b.d_labels) return OBIA_E_NOMEM;
    OBIA_HIP_TRY(hipMemcpyAsync(b.d_windows, b.windows.data(), sizeof(SrcWindow), hipMemcpyHostToDevice, ctx->stream));
    b.d_mask = const_cast<uint8_t *>(mask);
    const int to_lab = (C == 3 && p->convert2lab != 0) ? 1 : 0;
    const float ratio = (float)(1.0 / p->compactness);   // `image * ratio`: float32 array times Python float
    b.col_lb = slic_use_colour_bound(ratio, to_lab != 0) && !b.slic_zero && !b.exit_on_fixed_point && !direct;
    if (b.col_lb) {
        b.d_fbox = A.get<float>((size_t)feat_boxes(H, W) * 2 * b.CP);
        if (!b.d_fbox) return OBIA_E_NOMEM;
    }
    b.prescale = slic_prescale(ratio, p->normalize_bands, to_lab, b.slic_zero);
    OBIA_TRY(slic_prepare_features(ctx, b, img, H, W, p->normalize_bands, to_lab, ratio * b.prescale));
    std::vector<int> nseg(1, p->n_segments);
    OBIA_TRY(slic_plan_and_seed(ctx, b, nseg, nullptr, ext));
    if (b.probs[0].K <= 0) {
        set_error("mask is empty: nothing to segment");
        return OBIA_E_EMPTY;
    }
    OBIA_TRY(slic_run_sweeps(ctx, b));
    return OBIA_OK;
}

}  // namespace

extern "C" {

static int check_seeds(const obia_slic_seeds *seeds, int H, int W, ExternalSeeds &ext) {
    if (!seeds) { set_error("null seeds"); return OBIA_E_INVALID; }
    if (!seeds->yx || seeds->n < 1) { set_error("seeds: need at least one (y, x) pair"); return OBIA_E_INVALID; }
    for (int i = 0; i < seeds->n; ++i) {
        const double y = seeds->yx[2 * (size_t)i], x = seeds->yx[2 * (size_t)i + 1];
        if (!(y >= 0.0 && y <= (double)(H - 1) && x >= 0.0 && x <= (double)(W - 1))) {
            set_error("seed %d = (%g, %g) lies outside the (%d, %d) raster", i, y, x, H, W);
            return OBIA_E_INVALID;
        }
    }
    double st = seeds->steps_zyx[0];
    if (seeds->steps_zyx[1] > st) st = seeds->steps_zyx[1];
    if (seeds->steps_zyx[2] > st) st = seeds->steps_zyx[2];
    if (!(st > 0.0) || !(st < 1e9)) { set_error("seeds: steps must be positive and finite"); return OBIA_E_INVALID; }
    ext.yx = seeds->yx; ext.n = seeds->n; ext.step = st;   // step = max(steps), slic_superpixels.py:288
    return OBIA_OK;
}

static int slic_assign_only_impl(obia_ctx *ctx, const float *img, int H, int W, int C, const uint8_t *mask,
                                 const obia_slic_params *params, const ExternalSeeds *ext, int32_t *labels_pre_out,
                                 int *n_centroids_out) {
    OBIA_TRY(check_ctx(ctx));
    OBIA_TRY(check_slic_args(img, H, W, C, params));
    if (!labels_pre_out) { set_error("null output"); return OBIA_E_INVALID; }
    ctx->arena.reset();
    begin_timing(ctx);
    SlicBatch b;
    int rc;
    {
        ScopedSpan total(ctx, T_TOTAL);
        rc = slic_single(ctx, img, H, W, C, mask, params, b, ext);
        if (rc == OBIA_OK) {
            hipError_t e = hipMemcpyAsync(labels_pre_out, b.d_labels, sizeof(int32_t) * (size_t)H * W, hipMemcpyDeviceToDevice, ctx->stream);
            if (e != hipSuccess) { set_error("copy failed: %s", hipGetErrorString(e)); rc = OBIA_E_HIP; }
        }
    }
    if (rc != OBIA_OK) return rc;
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    resolve_timing(ctx);
    if (n_centroids_out) *n_centroids_out = b.probs[0].K;
    return OBIA_OK;
}

static int slic_full_impl(obia_ctx *ctx, const float *img, int H, int W, int C, const uint8_t *mask,
                          const obia_slic_params *params, const ExternalSeeds *ext, int32_t *labels_out, int *n_labels_out) {
    OBIA_TRY(check_ctx(ctx));
    OBIA_TRY(check_slic_args(img, H, W, C, params));
    if (!labels_out) { set_error("null output"); return OBIA_E_INVALID; }
    ctx->arena.reset();
    begin_timing(ctx);
    SlicBatch b;
    int n_labels = 0;
    int rc;
    {
        ScopedSpan total(ctx, T_TOTAL);
        rc = slic_single(ctx, img, H, W, C, mask, params, b, ext);
        if (rc == OBIA_OK) {
            if (params->enforce_connectivity) {
                // segment_size = mask.sum() / n_centroids  |  prod(shape) / n_centroids  (slic_superpixels.py:321-326)
                const double segment_size = (double)b.probs[0].n_valid / (double)b.probs[0].K;
                const int min_size = (int)(params->min_size_factor * segment_size);
                const double mxd = params->max_size_factor * segment_size;
                const int max_size = mxd >= 2147483647.0 ? 2147483647 : (mxd < 1.0 ? 1 : (int)mxd);
                rc = enforce_connectivity_dev(ctx, b.d_labels, H, W, min_size, max_size, params->start_label, labels_out, &n_labels);
            } else {
                hipError_t e = hipMemcpyAsync(labels_out, b.d_labels, sizeof(int32_t) * (size_t)H * W, hipMemcpyDeviceToDevice, ctx->stream);
                if (e != hipSuccess) { set_error("copy failed: %s", hipGetErrorString(e)); rc = OBIA_E_HIP; }
                n_labels = b.probs[0].K;
            }
        }
    }
    if (rc != OBIA_OK) return rc;
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    resolve_timing(ctx);
    if (n_labels_out) *n_labels_out = n_labels;
    return OBIA_OK;
}

int obia_slic_assign_only_f32_dev(obia_ctx *ctx, const float *img, int H, int W, int C, const uint8_t *mask,
                                  const obia_slic_params *params, int32_t *labels_pre_out, int *n_centroids_out) {
    return slic_assign_only_impl(ctx, img, H, W, C, mask, params, nullptr, labels_pre_out, n_centroids_out);
}

int obia_slic_f32_dev(obia_ctx *ctx, const float *img, int H, int W, int C, const uint8_t *mask,
                      const obia_slic_params *params, int32_t *labels_out, int *n_labels_out) {
    return slic_full_impl(ctx, img, H, W, C, mask, params, nullptr, labels_out, n_labels_out);
}

int obia_slic_seeded_f32_dev(obia_ctx *ctx, const float *img, int H, int W, int C, const uint8_t *mask,
                             const obia_slic_params *params, const obia_slic_seeds *seeds, int stage,
                             int32_t *labels_out, int *n_out) {
    if (H <= 0 || W <= 0) { set_error("bad image shape"); return OBIA_E_INVALID; }
    ExternalSeeds ext{};
    OBIA_TRY(check_seeds(seeds, H, W, ext));
    if (stage == 1) return slic_assign_only_impl(ctx, img, H, W, C, mask, params, &ext, labels_out, n_out);
    if (stage != 0) { set_error("stage must be 0 (full) or 1 (labels before connectivity)"); return OBIA_E_INVALID; }
    return slic_full_impl(ctx, img, H, W, C, mask, params, &ext, labels_out, n_out);
}

int obia_slic_f32(obia_ctx *ctx, const float *img, int H, int W, int C, const uint8_t *mask,
                  const obia_slic_params *params, int32_t *labels_out, int *n_labels_out) {
    OBIA_TRY(check_ctx(ctx));
    OBIA_TRY(check_slic_args(img, H, W, C, params));
    if (!labels_out) { set_error("null output"); return OBIA_E_INVALID; }
    const size_t npix = (size_t)H * W;
    float *d_img = nullptr;
    uint8_t *d_mask = nullptr;
    int32_t *d_lab = nullptr;
    int rc = OBIA_OK;
    // caller-visible buffers are not part of the workspace arena: plain allocations for the call
    if (hipMalloc(&d_img, npix * C * sizeof(float)) != hipSuccess || hipMalloc(&d_lab, npix * sizeof(int32_t)) != hipSuccess ||
        (mask && hipMalloc(&d_mask, npix) != hipSuccess)) {
        set_error("device allocation for host-pointer call failed");
        rc = OBIA_E_NOMEM;
    }
    if (rc == OBIA_OK && hipMemcpyAsync(d_img, img, npix * C * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = OBIA_E_HIP;
    if (rc == OBIA_OK && mask && hipMemcpyAsync(d_mask, mask, npix, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = OBIA_E_HIP;
    if (rc == OBIA_OK) rc = obia_slic_f32_dev(ctx, d_img, H, W, C, d_mask, params, d_lab, n_labels_out);
    if (rc == OBIA_OK && hipMemcpyAsync(labels_out, d_lab, npix * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = OBIA_E_HIP;
    (void)hipStreamSynchronize(ctx->stream);
    if (rc == OBIA_E_HIP) set_error("host<->device copy failed in obia_slic_f32");
    (void)hipFree(d_img); (void)hipFree(d_lab); (void)hipFree(d_mask);
    return rc;
}

int obia_enforce_connectivity_i32_dev(obia_ctx *ctx, const int32_t *labels_in, int H, int W, int min_size, int max_size,
                                      int start_label, int32_t *labels_out, int *n_labels_out) {
    OBIA_TRY(check_ctx(ctx));
    if (!labels_in || !labels_out || H <= 0 || W <= 0) { set_error("bad arguments"); return OBIA_E_INVALID; }
    if (start_label != 0 && start_label != 1) { set_error("start_label should be 0 or 1."); return OBIA_E_INVALID; }
    ctx->arena.reset();
    begin_timing(ctx);
    int n = 0;
    OBIA_TRY(enforce_connectivity_dev(ctx, labels_in, H, W, min_size, max_size, start_label, labels_out, &n));
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    resolve_timing(ctx);
    if (n_labels_out) *n_labels_out = n;
    return OBIA_OK;
}

int obia_zonal_stats_f32_dev(obia_ctx *ctx, const float *raw, const int32_t *labels, int H, int W, int C,
                             const int32_t *bands, int n_bands, int n_labels, int start_label, int64_t *count_out,
                             double *mean_out, double *var_out, float *min_out, float *max_out) {
    OBIA_TRY(check_ctx(ctx));
    if (!raw || !labels || !count_out || !mean_out || !var_out || !min_out || !max_out) { set_error("null pointer argument"); return OBIA_E_INVALID; }
    ctx->arena.reset();
    begin_timing(ctx);
    OBIA_TRY(zonal_stats_dev(ctx, raw, labels, H, W, C, bands, n_bands, n_labels, start_label, count_out, mean_out, var_out, min_out, max_out));
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    resolve_timing(ctx);
    return OBIA_OK;
}

int obia_zonal_stats_f32(obia_ctx *ctx, const float *raw, const int32_t *labels, int H, int W, int C,
                         const int32_t *bands, int n_bands, int n_labels, int start_label, int64_t *count_out,
                         double *mean_out, double *var_out, float *min_out, float *max_out) {
    OBIA_TRY(check_ctx(ctx));
    if (!raw || !labels || H <= 0 || W <= 0 || C <= 0 || n_labels < 0) { set_error("bad arguments"); return OBIA_E_INVALID; }
    const size_t npix = (size_t)H * W;
    const int nb = bands ? n_bands : C;
    if (nb < 1 || nb > 16) { set_error("n_bands %d out of range (1..16)", nb); return OBIA_E_UNSUPPORTED; }
    const size_t nl = (size_t)(n_labels > 0 ? n_labels : 1), nlb = nl * nb;
    float *d_raw = nullptr; int32_t *d_lab = nullptr;
    char *d_out = nullptr;
    const size_t out_bytes = nl * 8 + nlb * (8 + 8 + 4 + 4);
    int rc = OBIA_OK;
    if (hipMalloc(&d_raw, npix * C * sizeof(float)) != hipSuccess || hipMalloc(&d_lab, npix * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&d_out, out_bytes) != hipSuccess) {
        set_error("device allocation for host-pointer call failed");
        rc = OBIA_E_NOMEM;
    }
    int64_t *d_cnt = (int64_t *)d_out;
    double *d_mean = (double *)(d_out + nl * 8), *d_var = d_mean + nlb;
    float *d_mn = (float *)(d_var + nlb), *d_mx = d_mn + nlb;
    if (rc == OBIA_OK && hipMemcpyAsync(d_raw, raw, npix * C * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = OBIA_E_HIP;
    if (rc == OBIA_OK && hipMemcpyAsync(d_lab, labels, npix * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = OBIA_E_HIP;
    if (rc == OBIA_OK) rc = obia_zonal_stats_f32_dev(ctx, d_raw, d_lab, H, W, C, bands, n_bands, n_labels, start_label, d_cnt, d_mean, d_var, d_mn, d_mx);
    if (rc == OBIA_OK && n_labels > 0) {
        bool ok = hipMemcpyAsync(count_out, d_cnt, nl * 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                  hipMemcpyAsync(mean_out, d_mean, nlb * 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                  hipMemcpyAsync(var_out, d_var, nlb * 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                  hipMemcpyAsync(min_out, d_mn, nlb * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                  hipMemcpyAsync(max_out, d_mx, nlb * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        if (!ok) rc = OBIA_E_HIP;
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (rc == OBIA_E_HIP) set_error("host<->device copy failed in obia_zonal_stats_f32");
    (void)hipFree(d_raw); (void)hipFree(d_lab); (void)hipFree(d_out);
    return rc;
}

int obia_zonal_moments_f32_dev(obia_ctx *ctx, const float *raw, const int32_t *labels, int H, int W, int C,
                               const int32_t *bands, int n_bands, int n_labels, int start_label, const double *mean_dev,
                               double *skew_out, double *kurt_out) {
    OBIA_TRY(check_ctx(ctx));
    if (!raw || !labels || !mean_dev || !skew_out || !kurt_out) { set_error("null pointer argument"); return OBIA_E_INVALID; }
    ctx->arena.reset();
    begin_timing(ctx);
    OBIA_TRY(zonal_moments_dev(ctx, raw, labels, H, W, C, bands, n_bands, n_labels, start_label, mean_dev, skew_out, kurt_out));
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    resolve_timing(ctx);
    return OBIA_OK;
}

int obia_zonal_moments_f32(obia_ctx *ctx, const float *raw, const int32_t *labels, int H, int W, int C,
                           const int32_t *bands, int n_bands, int n_labels, int start_label, double *skew_out,
                           double *kurt_out) {
    OBIA_TRY(check_ctx(ctx));
    if (!raw || !labels || !skew_out || !kurt_out || H <= 0 || W <= 0 || C <= 0 || n_labels < 0) { set_error("bad arguments"); return OBIA_E_INVALID; }
    const size_t npix = (size_t)H * W;
    const int nb = bands ? n_bands : C;
    if (nb < 1 || nb > 16) { set_error("n_bands %d out of range (1..16)", nb); return OBIA_E_UNSUPPORTED; }
    const size_t nl = (size_t)(n_labels > 0 ? n_labels : 1), nlb = nl * nb;
    float *d_raw = nullptr; int32_t *d_lab = nullptr;
    char *d_out = nullptr;
    const size_t out_bytes = nl * 8 + nlb * (8 + 8 + 4 + 4 + 8 + 8);
    int rc = OBIA_OK;
    if (hipMalloc(&d_raw, npix * C * sizeof(float)) != hipSuccess || hipMalloc(&d_lab, npix * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&d_out, out_bytes) != hipSuccess) {
        set_error("device allocation for host-pointer call failed");
        rc = OBIA_E_NOMEM;
    }
    int64_t *d_cnt = (int64_t *)d_out;
    double *d_mean = (double *)(d_out + nl * 8), *d_var = d_mean + nlb, *d_skew = d_var + nlb, *d_kurt = d_skew + nlb;
    float *d_mn = (float *)(d_kurt + nlb), *d_mx = d_mn + nlb;
    if (rc == OBIA_OK && hipMemcpyAsync(d_raw, raw, npix * C * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = OBIA_E_HIP;
    if (rc == OBIA_OK && hipMemcpyAsync(d_lab, labels, npix * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = OBIA_E_HIP;
    if (rc == OBIA_OK) rc = obia_zonal_stats_f32_dev(ctx, d_raw, d_lab, H, W, C, bands, n_bands, n_labels, start_label, d_cnt, d_mean, d_var, d_mn, d_mx);
    if (rc == OBIA_OK) rc = obia_zonal_moments_f32_dev(ctx, d_raw, d_lab, H, W, C, bands, n_bands, n_labels, start_label, d_mean, d_skew, d_kurt);
    if (rc == OBIA_OK && n_labels > 0) {
        bool ok = hipMemcpyAsync(skew_out, d_skew, nlb * 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                  hipMemcpyAsync(kurt_out, d_kurt, nlb * 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        if (!ok) rc = OBIA_E_HIP;
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (rc == OBIA_E_HIP) set_error("host<->device copy failed in obia_zonal_moments_f32");
    (void)hipFree(d_raw); (void)hipFree(d_lab); (void)hipFree(d_out);
    return rc;
}

}  // extern "C"
