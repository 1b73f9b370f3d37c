/*
 * obia_oracle.c -- CPU restatement of the hot path of iosefa/obia.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may build, link or call it; the product path (obia_amd/) never does.
 *
 * The arithmetic of the path is not in /root/reference itself: obia calls
 *   skimage.segmentation.slic / quickshift   (obia/segmentation/segment_boundaries.py:48-51)
 *   np.mean/var/min/max                      (obia/segmentation/segment_statistics.py:165-172)
 * scikit-image is a third-party dependency (pyproject.toml:23, `scikit-image>=0.23.2`, no lock
 * file, not vendored).  What follows restates its published algorithm (the Python driver
 * slic_superpixels.py and the Cython kernels _slic_cython / _enforce_label_connectivity_cython /
 * _quickshift_cython) in plain C.  PINNING: every function here is checked bit-for-bit against
 * scikit-image 0.18.3 executed in the build container (tests/golden/gen_goldens.py, run with
 * /opt/conda/bin/python3.9; outputs committed under tests/golden/), and against scikit-image's own
 * known-answer block-image tests re-typed in tests/test_oracle_known_answers.py.
 *
 * All image arithmetic is float32, exactly as obia hands float32 `img_data` to scikit-image
 * (obia/handlers/geotif.py:100, obia/utils/tiling.py:47).
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off: no FMA contraction, like the x86-64
 * baseline wheels of scikit-image).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define OBIA_OK 0
#define OBIA_EINVAL -1
#define OBIA_ENOMEM -2

/* ------------------------------------------------------------------------------------------
 * obia.segmentation.segment_boundaries.normalize_band  (segment_boundaries.py:11-16,32-33)
 *   (band - min) / (max - min), float32, in place, every band; constant band -> 0/0 = NaN
 *   (the reference has no guard; the oracle reproduces it).
 * img: (H*W, C) interleaved.  mn/mx (nullable) receive the per-band extrema.
 * np.min/np.max propagate NaN; so do we.
 * ------------------------------------------------------------------------------------------ */
void obia_oracle_band_minmax(const float *img, int64_t npix, int C, float *mn, float *mx)
{
    for (int c = 0; c < C; ++c) {
        float lo = img[c], hi = img[c];
        int has_nan = 0;
        for (int64_t i = 0; i < npix; ++i) {
            float v = img[i * C + c];
            if (v != v) { has_nan = 1; break; }
            if (v < lo) lo = v;
            if (v > hi) hi = v;
        }
        if (has_nan) { lo = NAN; hi = NAN; }
        mn[c] = lo; mx[c] = hi;
    }
}

void obia_oracle_normalize(float *img, int64_t npix, int C)
{
    float mn[64], mx[64];
    if (C > 64) return;
    obia_oracle_band_minmax(img, npix, C, mn, mx);
    for (int c = 0; c < C; ++c) {
        float den = mx[c] - mn[c];
        for (int64_t i = 0; i < npix; ++i)
            img[i * C + c] = (img[i * C + c] - mn[c]) / den;
    }
}

/* ------------------------------------------------------------------------------------------
 * skimage.color.rgb2lab on float32 (colorconv.py: rgb2xyz + xyz2lab, illuminant D65, observer 2)
 * reached from slic() when the image has exactly 3 channels (slic_superpixels.py:250-254).
 * numpy evaluates power/cbrt with its own float32 SIMD kernels and the 3x3 product through
 * BLAS, so this restatement is pinned to scikit-image within a few float32 ulp, not bitwise
 * (tests/test_oracle_golden.py::test_rgb2lab states the tolerance).
 * ------------------------------------------------------------------------------------------ */
void obia_oracle_rgb2lab_f32(const float *rgb, float *lab, int64_t npix)
{
    /* xyz_from_rgb (sRGB, D65), cast to float32 as `xyz_from_rgb.T.astype(arr.dtype)` does */
    static const double M[3][3] = {
        {0.412453, 0.357580, 0.180423},
        {0.212671, 0.715160, 0.072169},
        {0.019334, 0.119193, 0.950227}};
    static const double white[3] = {0.95047, 1.0, 1.08883};
    float m[3][3], wr[3];
    for (int i = 0; i < 3; ++i) {
        wr[i] = (float)white[i];
        for (int j = 0; j < 3; ++j) m[i][j] = (float)M[i][j];
    }
    for (int64_t p = 0; p < npix; ++p) {
        float a[3];
        for (int c = 0; c < 3; ++c) {
            float v = rgb[p * 3 + c];
            if (v > 0.04045f) v = powf((v + 0.055f) / 1.055f, 2.4f);
            else v = v / 12.92f;
            a[c] = v;
        }
        float xyz[3];
        for (int i = 0; i < 3; ++i) {
            float s = a[0] * m[i][0];
            s = s + a[1] * m[i][1];
            s = s + a[2] * m[i][2];
            s = s / wr[i];
            if (s > 0.008856f) s = cbrtf(s);
            else s = 7.787f * s + 16.0f / 116.0f;
            xyz[i] = s;
        }
        lab[p * 3 + 0] = 116.0f * xyz[1] - 16.0f;
        lab[p * 3 + 1] = 500.0f * (xyz[0] - xyz[1]);
        lab[p * 3 + 2] = 200.0f * (xyz[1] - xyz[2]);
    }
}

/* ------------------------------------------------------------------------------------------
 * skimage.util.regular_grid(ar_shape=(1,H,W), n_points)   (util/_regular_grid.py:61-83)
 * out[0..3] = start_y, step_y, start_x, step_x ; step 0 means slice(None) (every index, and
 * `steps` entry 1.0 in _get_grid_centroids, slic_superpixels.py:101-103).
 * ------------------------------------------------------------------------------------------ */
static double py_round_half_even(double v) { return nearbyint(v); /* default FE_TONEAREST */ }

void obia_oracle_regular_grid(int64_t H, int64_t W, int64_t n_points, int64_t out[4])
{
    int64_t dims[3] = {1, H, W};
    int order[3] = {0, 1, 2};              /* argsort, stable */
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (dims[order[j]] < dims[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    double sd[3];
    for (int i = 0; i < 3; ++i) sd[i] = (double)dims[order[i]];
    double space = sd[0] * sd[1] * sd[2];
    int64_t start[3], step[3];
    if (space <= (double)n_points) {
        out[0] = 0; out[1] = 0; out[2] = 0; out[3] = 0;
        return;
    }
    double st[3];
    for (int i = 0; i < 3; ++i) st[i] = pow(space / (double)n_points, 1.0 / 3.0);
    if (sd[0] < st[0] || sd[1] < st[1] || sd[2] < st[2]) {
        for (int d = 0; d < 3; ++d) {
            st[d] = sd[d];
            double sp = 1.0;
            for (int e = d + 1; e < 3; ++e) sp *= sd[e];
            if (d < 2) {
                double v = pow(sp / (double)n_points, 1.0 / (double)(3 - d - 1));
                for (int e = d + 1; e < 3; ++e) st[e] = v;
            }
            if (sd[0] >= st[0] && sd[1] >= st[1] && sd[2] >= st[2]) break;
        }
    }
    for (int i = 0; i < 3; ++i) {
        start[i] = (int64_t)floor(st[i] / 2.0);
        step[i] = (int64_t)py_round_half_even(st[i]);
    }
    /* unsort: slices[i] belongs to dims[order[i]] */
    int64_t s_of_dim[3], t_of_dim[3];
    for (int i = 0; i < 3; ++i) { s_of_dim[order[i]] = start[i]; t_of_dim[order[i]] = step[i]; }
    out[0] = s_of_dim[1]; out[1] = t_of_dim[1];
    out[2] = s_of_dim[2]; out[3] = t_of_dim[2];
}

/* number of indices produced by slice(start, None, step) over length L */
static int64_t slice_len(int64_t L, int64_t start, int64_t step)
{
    if (step == 0) return L;
    if (start >= L) return 0;
    return (L - start + step - 1) / step;
}

/* _get_grid_centroids (slic_superpixels.py:71-104): returns K; fills yx (K,2) if non-NULL and
 * steps[2] (float steps for y and x; 1.0 where the slice step is None). Row-major order. */
int64_t obia_oracle_grid_centroids(int64_t H, int64_t W, int64_t n_segments, int64_t *yx, double steps[2])
{
    int64_t g[4];
    obia_oracle_regular_grid(H, W, n_segments, g);
    int64_t ny = slice_len(H, g[0], g[1]), nx = slice_len(W, g[2], g[3]);
    int64_t sy = g[1] ? g[1] : 1, sx = g[3] ? g[3] : 1;
    if (steps) { steps[0] = g[1] ? (double)g[1] : 1.0; steps[1] = g[3] ? (double)g[3] : 1.0; }
    if (yx) {
        int64_t k = 0;
        for (int64_t iy = 0; iy < ny; ++iy)
            for (int64_t ix = 0; ix < nx; ++ix) {
                yx[2 * k] = g[0] + iy * sy;
                yx[2 * k + 1] = g[2] + ix * sx;
                ++k;
            }
    }
    return ny * nx;
}

/* ------------------------------------------------------------------------------------------
 * _slic_cython (scikit-image 0.18.3 _slic.pyx; called from slic_superpixels.py:310-318)
 *   image   (H,W,C) float32, already multiplied by 1/compactness
 *   mask    (H,W) uint8 or NULL
 *   segments(K,2+C) float32: cy, cx, colour[C]   (the z column of the 3-D original is dropped:
 *            depth is 1, cz = 0, dz = 0 and `0 + dy` is exact)           -- updated IN PLACE
 *   step    float: max(steps) of the DRIVER's grid
 *   nearest (H,W) int64 out
 * Window steps come from regular_grid((1,H,W), K) with K = number of centroids.
 * ------------------------------------------------------------------------------------------ */
/* Sum mode of the centroid update (test infrastructure): 0 = the reference's -- float32 sums accumulated pixel by pixel in raster order
 * (_slic.pyx); 1 = the HIP path's -- every feature truncated to 32-bit fixed point (`(int)(f * 2^s)`, 2^s the largest power of two with
 * max|feature| * 2^s < 2^29), added in 64-bit integers, converted back and rounded ONCE (slic_sweep.hip: to_fixed32,
 * slic_prep_lane_kernel).  Mode 1 exists to show that the summation order is the ONLY difference between the two paths: with it the
 * oracle and the HIP path agree bit for bit at every compactness (tests/test_gpu_exact_sums.py). */
static int g_sum_mode = 0;
void obia_oracle_set_sum_mode(int mode) { g_sum_mode = mode; }
static double fixed_point_scale(const float *image, int64_t n)
{
    float maxabs = 0.0f;
    for (int64_t i = 0; i < n; ++i) { const float a = fabsf(image[i]); if (a > maxabs) maxabs = a; }
    int sh = 0;
    if (maxabs > 0.0f) { int e = 0; (void)frexp((double)maxabs, &e); sh = 29 - e; }
    if (sh > 100) sh = 100;
    if (sh < -90) sh = -90;
    return ldexp(1.0, sh);
}

/* spacing (sy, sx): _slic.pyx scales the coordinate differences before squaring them, `dy = (sy * (cy - y)) ** 2` in the image's
 * float type (the depth term is (sz * 0) ** 2 = 0 whatever sz); (1, 1) multiplies by 1.0f, which is exact. */
int obia_oracle_slic_core_sp(const float *image, const uint8_t *mask, float *segments,
                             int64_t H, int64_t W, int C, int64_t K, float step,
                             int max_iter, int slic_zero, int ignore_color, int start_label,
                             int64_t *nearest, float sy, float sx);
int obia_oracle_slic_core(const float *image, const uint8_t *mask, float *segments,
                          int64_t H, int64_t W, int C, int64_t K, float step,
                          int max_iter, int slic_zero, int ignore_color, int start_label,
                          int64_t *nearest)
{
    return obia_oracle_slic_core_sp(image, mask, segments, H, W, C, K, step, max_iter, slic_zero, ignore_color, start_label, nearest, 1.0f, 1.0f);
}
int obia_oracle_slic_core_sp(const float *image, const uint8_t *mask, float *segments,
                             int64_t H, int64_t W, int C, int64_t K, float step,
                             int max_iter, int slic_zero, int ignore_color, int start_label,
                             int64_t *nearest, float sy, float sx)
{
    int64_t g[4];
    obia_oracle_regular_grid(H, W, K, g);
    int64_t step_y = g[1] ? g[1] : 1, step_x = g[3] ? g[3] : 1;
    const int F = 2 + C;
    const int64_t npix = H * W;
    float *distance = (float *)malloc(sizeof(float) * (size_t)npix);
    int64_t *n_elems = (int64_t *)malloc(sizeof(int64_t) * (size_t)(K > 0 ? K : 1));
    float *max_dist_color = (float *)malloc(sizeof(float) * (size_t)(K > 0 ? K : 1));
    if (!distance || !n_elems || !max_dist_color) { free(distance); free(n_elems); free(max_dist_color); return OBIA_ENOMEM; }
    for (int64_t k = 0; k < K; ++k) max_dist_color[k] = 1.0f;
    const int64_t mask_label = start_label - 1;
    for (int64_t i = 0; i < npix; ++i) nearest[i] = mask_label;
    /* spatial_weight = float(1) / (step ** 2): double arithmetic on a float `step`, stored float */
    const float spatial_weight = (float)(1.0 / ((double)step * (double)step));

    for (int it = 0; it < max_iter; ++it) {
        int change = 0;
        for (int64_t i = 0; i < npix; ++i) distance[i] = INFINITY; /* `distance[:] = DBL_MAX` stored to float32 */
        for (int64_t k = 0; k < K; ++k) {
            const float cy = segments[k * F + 0], cx = segments[k * F + 1];
            if (cy != cy || cx != cx) continue;   /* NaN centroid: its window casts to an empty range */
            float fy0 = cy - (float)(2 * step_y); if (!(fy0 > 0.0f)) fy0 = 0.0f;
            float fy1 = cy + (float)(2 * step_y) + 1.0f; if (!(fy1 < (float)H)) fy1 = (float)H;
            float fx0 = cx - (float)(2 * step_x); if (!(fx0 > 0.0f)) fx0 = 0.0f;
            float fx1 = cx + (float)(2 * step_x) + 1.0f; if (!(fx1 < (float)W)) fx1 = (float)W;
            const int64_t y0 = (int64_t)fy0, y1 = (int64_t)fy1, x0 = (int64_t)fx0, x1 = (int64_t)fx1;
            for (int64_t y = y0; y < y1; ++y) {
                float ty = cy - (float)y;
                ty = sy * ty;
                const float dy = ty * ty;
                for (int64_t x = x0; x < x1; ++x) {
                    if (mask && !mask[y * W + x]) continue;
                    float tx = cx - (float)x;
                    tx = sx * tx;
                    const float dx = tx * tx;
                    float d = (dy + dx) * spatial_weight;
                    if (!ignore_color) {
                        float dc = 0.0f;
                        const float *px = image + (y * W + x) * C;
                        const float *sc = segments + k * F + 2;
                        for (int c = 0; c < C; ++c) {
                            float t = px[c] - sc[c];
                            dc += t * t;
                        }
                        if (slic_zero) d += dc / max_dist_color[k];
                        else d += dc;
                    }
                    if (distance[y * W + x] > d) {
                        nearest[y * W + x] = k + start_label;
                        distance[y * W + x] = d;
                        change = 1;
                    }
                }
            }
        }
        if (!change) break;
        if (g_sum_mode == 1) {   /* the HIP path's integer sums (see g_sum_mode) */
            const double fscale = fixed_point_scale(image, npix * C);
            const float fs = (float)fscale;
            const double inv_fscale = 1.0 / fscale;
            int64_t *acc = (int64_t *)calloc((size_t)(K > 0 ? K : 1) * (size_t)(F + 1), sizeof(int64_t));
            if (!acc) { free(distance); free(n_elems); free(max_dist_color); return OBIA_ENOMEM; }
            for (int64_t y = 0; y < H; ++y)
                for (int64_t x = 0; x < W; ++x) {
                    if (mask && !mask[y * W + x]) continue;
                    const int64_t k = nearest[y * W + x] - start_label;
                    if (k < 0) continue;
                    int64_t *a = acc + k * (F + 1);
                    a[0] += 1; a[1] += y; a[2] += x;
                    const float *px = image + (y * W + x) * C;
                    for (int c = 0; c < C; ++c) a[3 + c] += (int64_t)(int)(px[c] * fs);
                }
            for (int64_t k = 0; k < K; ++k) {
                const int64_t *a = acc + k * (F + 1);
                const float fn = (float)(double)a[0];
                n_elems[k] = a[0];
                segments[k * F + 0] = (float)(double)a[1] / fn;
                segments[k * F + 1] = (float)(double)a[2] / fn;
                for (int c = 0; c < C; ++c) segments[k * F + 2 + c] = (float)((double)a[3 + c] * inv_fscale) / fn;
            }
            free(acc);
        } else {
        /* recompute centres: sequential float32 accumulation in raster order */
        for (int64_t k = 0; k < K; ++k) n_elems[k] = 0;
        memset(segments, 0, sizeof(float) * (size_t)(K * F));
        for (int64_t y = 0; y < H; ++y)
            for (int64_t x = 0; x < W; ++x) {
                if (mask && !mask[y * W + x]) continue;
                const int64_t k = nearest[y * W + x] - start_label;
                /* A valid pixel that no window has reached yet still holds mask_label: the Cython code indexes
                 * n_elems[-1] there (boundscheck off: undefined behaviour).  Defined here, and in the HIP path, as
                 * "not accumulated". */
                if (k < 0) continue;
                n_elems[k] += 1;
                segments[k * F + 0] += (float)y;
                segments[k * F + 1] += (float)x;
                const float *px = image + (y * W + x) * C;
                for (int c = 0; c < C; ++c) segments[k * F + 2 + c] += px[c];
            }
        for (int64_t k = 0; k < K; ++k)
            for (int f = 0; f < F; ++f) segments[k * F + f] /= (float)n_elems[k];
        }
        if (slic_zero) {
            for (int64_t y = 0; y < H; ++y)
                for (int64_t x = 0; x < W; ++x) {
                    if (mask && !mask[y * W + x]) continue;
                    const int64_t k = nearest[y * W + x] - start_label;
                    if (k < 0) continue;   /* as above */
                    float dc = 0.0f;
                    const float *px = image + (y * W + x) * C;
                    for (int c = 0; c < C; ++c) {
                        float t = px[c] - segments[k * F + 2 + c];
                        dc += t * t;
                    }
                    if (max_dist_color[k] < dc) max_dist_color[k] = dc;
                }
        }
    }
    free(distance); free(n_elems); free(max_dist_color);
    return OBIA_OK;
}

/* ------------------------------------------------------------------------------------------
 * _enforce_label_connectivity_cython (scikit-image 0.18.3 _slic.pyx; slic_superpixels.py:320-328)
 * 2-D, 4-connectivity, neighbour order (x+1, x-1, y+1, y-1).
 * ------------------------------------------------------------------------------------------ */
int obia_oracle_enforce_connectivity(const int64_t *labels, int64_t H, int64_t W,
                                     int64_t min_size, int64_t max_size, int start_label,
                                     int64_t *out)
{
    static const int ddx[4] = {1, -1, 0, 0};
    static const int ddy[4] = {0, 0, 1, -1};
    const int64_t mask_label = start_label - 1;
    const int64_t npix = H * W;
    int64_t cap = max_size > 0 ? max_size : 1;
    if (cap > npix + 1) cap = npix + 1;   /* a component never holds more than npix pixels */
    int64_t *coord = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)cap);
    if (!coord) return OBIA_ENOMEM;
    for (int64_t i = 0; i < npix; ++i) out[i] = mask_label;
    int64_t cur = start_label;
    for (int64_t y = 0; y < H; ++y)
        for (int64_t x = 0; x < W; ++x) {
            if (labels[y * W + x] == mask_label) continue;
            if (out[y * W + x] > mask_label) continue;
            int64_t adjacent = 0;
            const int64_t label = labels[y * W + x];
            out[y * W + x] = cur;
            int64_t size = 1, visited = 0;
            coord[0] = y; coord[1] = x;
            while (visited < size && size < max_size) {
                for (int i = 0; i < 4; ++i) {
                    const int64_t yy = coord[2 * visited] + ddy[i];
                    const int64_t xx = coord[2 * visited + 1] + ddx[i];
                    if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                        if (labels[yy * W + xx] == label && out[yy * W + xx] == mask_label) {
                            out[yy * W + xx] = cur;
                            coord[2 * size] = yy; coord[2 * size + 1] = xx;
                            size += 1;
                            if (size >= max_size) break;
                        } else if (out[yy * W + xx] > mask_label && out[yy * W + xx] != cur) {
                            adjacent = out[yy * W + xx];
                        }
                    }
                }
                visited += 1;
            }
            if (size < min_size) {
                for (int64_t i = 0; i < size; ++i) out[coord[2 * i] * W + coord[2 * i + 1]] = adjacent;
            } else {
                cur += 1;
            }
        }
    free(coord);
    return OBIA_OK;
}

/* ------------------------------------------------------------------------------------------
 * The build's deterministic replacement for _get_mask_centroids (slic_superpixels.py:14-68).
 * The reference seeds maskSLIC with RandomState(123).choice + scipy kmeans2 + a KxK pdist; that is
 * neither stable across scikit-image versions nor scalable (hard part 3).  Rule used by
 * the product and restated here: lay regular_grid((1,H,W), n_eff) with n_eff = round(n_segments *
 * H*W / n_valid) and keep the grid points that fall on valid pixels; if none falls on one, seed the
 * first valid pixel.  steps = grid steps.  Returns K (0 if the mask is empty).
 * ------------------------------------------------------------------------------------------ */
int64_t obia_oracle_masked_grid_centroids(const uint8_t *mask, int64_t H, int64_t W, int64_t n_segments,
                                          int64_t *yx, double steps[2])
{
    int64_t n_valid = 0;
    for (int64_t i = 0; i < H * W; ++i) n_valid += mask[i] != 0;
    if (n_valid == 0 || n_segments <= 0) return 0;
    double ne = nearbyint((double)n_segments * (double)(H * W) / (double)n_valid);
    int64_t n_eff = ne < 1.0 ? 1 : (int64_t)ne;
    int64_t g[4];
    obia_oracle_regular_grid(H, W, n_eff, g);
    int64_t ny = slice_len(H, g[0], g[1]), nx = slice_len(W, g[2], g[3]);
    int64_t sy = g[1] ? g[1] : 1, sx = g[3] ? g[3] : 1;
    if (steps) { steps[0] = g[1] ? (double)g[1] : 1.0; steps[1] = g[3] ? (double)g[3] : 1.0; }
    int64_t k = 0;
    for (int64_t iy = 0; iy < ny; ++iy)
        for (int64_t ix = 0; ix < nx; ++ix) {
            int64_t y = g[0] + iy * sy, x = g[2] + ix * sx;
            if (mask[y * W + x]) {
                if (yx) { yx[2 * k] = y; yx[2 * k + 1] = x; }
                ++k;
            }
        }
    if (k == 0) {
        for (int64_t i = 0; i < H * W; ++i)
            if (mask[i]) { if (yx) { yx[0] = i / W; yx[1] = i % W; } k = 1; break; }
    }
    return k;
}

/* ------------------------------------------------------------------------------------------
 * slic() driver (slic_superpixels.py:107-333) for a 2-D multichannel float32 image.
 *   image (H,W,C) float32, NOT modified.  convert2lab: -1 auto (C==3), 0, 1.
 *   seeds_yx/n_seeds/seed_steps: optional externally supplied initial centroids (used to pin the
 *   maskSLIC path on scikit-image's own seeds); when NULL: grid seeding (mask NULL) or the
 *   build's masked-grid rule (mask given).
 *   labels_pre (nullable) receives the labels before connectivity enforcement.
 *   centroids_out (nullable, K*(2+C)) receives the final centroids; *K_out the centroid count.
 * ------------------------------------------------------------------------------------------ */
int obia_oracle_slic_sp(const float *image, const uint8_t *mask, int64_t H, int64_t W, int C,
                        int64_t n_segments, double compactness, int max_iter, int convert2lab,
                        int enforce_connectivity, double min_size_factor, double max_size_factor,
                        int slic_zero, int start_label,
                        const double *seeds_yx, int64_t n_seeds, const double *seed_steps,
                        int64_t *labels, int64_t *labels_pre, float *centroids_out, int64_t *K_out, const double *spacing_yx);
int obia_oracle_slic(const float *image, const uint8_t *mask, int64_t H, int64_t W, int C,
                     int64_t n_segments, double compactness, int max_iter, int convert2lab,
                     int enforce_connectivity, double min_size_factor, double max_size_factor,
                     int slic_zero, int start_label,
                     const double *seeds_yx, int64_t n_seeds, const double *seed_steps,
                     int64_t *labels, int64_t *labels_pre, float *centroids_out, int64_t *K_out)
{
    return obia_oracle_slic_sp(image, mask, H, W, C, n_segments, compactness, max_iter, convert2lab, enforce_connectivity, min_size_factor,
                               max_size_factor, slic_zero, start_label, seeds_yx, n_seeds, seed_steps, labels, labels_pre, centroids_out,
                               K_out, NULL);
}
/* spacing_yx (nullable = (1, 1)): slic()'s `spacing` for the row and column axes, cast to the image's float32 as
 * `np.ascontiguousarray(spacing, dtype=dtype)` does (slic_superpixels.py) */
int obia_oracle_slic_sp(const float *image, const uint8_t *mask, int64_t H, int64_t W, int C,
                        int64_t n_segments, double compactness, int max_iter, int convert2lab,
                        int enforce_connectivity, double min_size_factor, double max_size_factor,
                        int slic_zero, int start_label,
                        const double *seeds_yx, int64_t n_seeds, const double *seed_steps,
                        int64_t *labels, int64_t *labels_pre, float *centroids_out, int64_t *K_out, const double *spacing_yx)
{
    const float sp_y = spacing_yx ? (float)spacing_yx[0] : 1.0f, sp_x = spacing_yx ? (float)spacing_yx[1] : 1.0f;
    if (start_label != 0 && start_label != 1) return OBIA_EINVAL;
    if (convert2lab == 1 && C != 3) return OBIA_EINVAL;
    const int64_t npix = H * W;
    float *img = (float *)malloc(sizeof(float) * (size_t)(npix * C));
    if (!img) return OBIA_ENOMEM;
    if (C == 3 && convert2lab != 0) obia_oracle_rgb2lab_f32(image, img, npix);
    else memcpy(img, image, sizeof(float) * (size_t)(npix * C));

    int64_t K;
    double steps[2];
    int64_t *yx = NULL;
    float *segments;
    const int F = 2 + C;
    if (seeds_yx) {
        K = n_seeds;
        steps[0] = seed_steps[0]; steps[1] = seed_steps[1];
        segments = (float *)calloc((size_t)(K * F), sizeof(float));
        for (int64_t k = 0; k < K; ++k) { segments[k * F] = (float)seeds_yx[2 * k]; segments[k * F + 1] = (float)seeds_yx[2 * k + 1]; }
    } else {
        if (mask) K = obia_oracle_masked_grid_centroids(mask, H, W, n_segments, NULL, steps);
        else K = obia_oracle_grid_centroids(H, W, n_segments, NULL, steps);
        if (K <= 0) { free(img); return OBIA_EINVAL; }
        yx = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)K);
        if (mask) obia_oracle_masked_grid_centroids(mask, H, W, n_segments, yx, steps);
        else obia_oracle_grid_centroids(H, W, n_segments, yx, steps);
        segments = (float *)calloc((size_t)(K * F), sizeof(float));
        for (int64_t k = 0; k < K; ++k) { segments[k * F] = (float)yx[2 * k]; segments[k * F + 1] = (float)yx[2 * k + 1]; }
        free(yx);
    }
    /* step = max(steps); the 3-D original also has steps[0] = 1.0 for the depth axis */
    double stepd = steps[0] > steps[1] ? steps[0] : steps[1];
    if (stepd < 1.0) stepd = 1.0;
    const float step = (float)stepd;
    /* image = image * ratio, float32 array times Python float -> float32 multiply by float32(ratio) */
    const float ratio = (float)(1.0 / compactness);
    for (int64_t i = 0; i < npix * C; ++i) img[i] = img[i] * ratio;

    int rc = OBIA_OK;
    if (mask) /* step 2 of maskSLIC: spatial-only pre-pass updates `segments` in place */
        rc = obia_oracle_slic_core_sp(img, mask, segments, H, W, C, K, step, max_iter, slic_zero, 1, start_label, labels, sp_y, sp_x);
    if (rc == OBIA_OK)
        rc = obia_oracle_slic_core_sp(img, mask, segments, H, W, C, K, step, max_iter, slic_zero, 0, start_label, labels, sp_y, sp_x);
    if (rc == OBIA_OK && labels_pre) memcpy(labels_pre, labels, sizeof(int64_t) * (size_t)npix);
    if (rc == OBIA_OK && enforce_connectivity) {
        double segment_size;
        if (mask) {
            int64_t nv = 0;
            for (int64_t i = 0; i < npix; ++i) nv += mask[i] != 0;
            segment_size = (double)nv / (double)K;
        } else segment_size = (double)npix / (double)K;
        int64_t min_size = (int64_t)(min_size_factor * segment_size);
        int64_t max_size = (int64_t)(max_size_factor * segment_size);
        int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)npix);
        if (!tmp) rc = OBIA_ENOMEM;
        else {
            memcpy(tmp, labels, sizeof(int64_t) * (size_t)npix);
            rc = obia_oracle_enforce_connectivity(tmp, H, W, min_size, max_size, start_label, labels);
            free(tmp);
        }
    }
    if (centroids_out) memcpy(centroids_out, segments, sizeof(float) * (size_t)(K * F));
    if (K_out) *K_out = K;
    free(segments); free(img);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Zonal statistics, the timing port of calculate_spectral_stats (segment_statistics.py:143-172)
 * under the equivalence of SURVEY.md 3.3 (pixels of polygon p == pixels of label p).
 * float64 accumulation; used by bench.py's cpu_baseline.  The bit-faithful checker (np.mean /
 * np.var / np.min / np.max on the float32 pixels of each label) lives in oracle/oracle.py.
 * labels in [start_label, start_label + n_labels); others ignored.
 * ------------------------------------------------------------------------------------------ */
int obia_oracle_zonal_stats(const float *raw, const int64_t *labels, int64_t npix, int C,
                            const int *bands, int n_bands, int64_t n_labels, int start_label,
                            int64_t *count, double *mean, double *var, float *mn, float *mx)
{
    double *s1 = (double *)calloc((size_t)(n_labels * n_bands), sizeof(double));
    double *s2 = (double *)calloc((size_t)(n_labels * n_bands), sizeof(double));
    if (!s1 || !s2) { free(s1); free(s2); return OBIA_ENOMEM; }
    for (int64_t i = 0; i < n_labels; ++i) count[i] = 0;
    for (int64_t i = 0; i < n_labels * n_bands; ++i) { mn[i] = INFINITY; mx[i] = -INFINITY; }
    for (int64_t p = 0; p < npix; ++p) {
        int64_t l = labels[p] - start_label;
        if (l < 0 || l >= n_labels) continue;
        count[l] += 1;
        for (int b = 0; b < n_bands; ++b) {
            float v = raw[p * C + bands[b]];
            s1[l * n_bands + b] += v;
            s2[l * n_bands + b] += (double)v * (double)v;
            if (v < mn[l * n_bands + b]) mn[l * n_bands + b] = v;
            if (v > mx[l * n_bands + b]) mx[l * n_bands + b] = v;
        }
    }
    for (int64_t l = 0; l < n_labels; ++l)
        for (int b = 0; b < n_bands; ++b) {
            int64_t i = l * n_bands + b;
            if (count[l] == 0) { mean[i] = NAN; var[i] = NAN; mn[i] = NAN; mx[i] = NAN; continue; }
            double m = s1[i] / (double)count[l];
            mean[i] = m;
            double v = s2[i] / (double)count[l] - m * m;
            var[i] = v < 0 ? 0 : v;
        }
    free(s1); free(s2);
    return OBIA_OK;
}

/* ------------------------------------------------------------------------------------------
 * _quickshift_cython (scikit-image 0.18.3 _quickshift_cy.pyx; called from _quickshift.py:71-73).
 * image (H,W,C) float64 (already Lab-converted where applicable and multiplied by `ratio`),
 * noise (H,W) float64 = RandomState(random_seed).normal(scale=1e-5, size=(H,W)) supplied by caller.
 * ------------------------------------------------------------------------------------------ */
int obia_oracle_quickshift_core(const double *image, const double *noise, int64_t H, int64_t W, int C,
                                double kernel_size, double max_dist, int64_t *labels_out)
{
    const double inv_ks2 = -0.5 / (kernel_size * kernel_size);
    const int64_t kw = (int64_t)ceil(3.0 * kernel_size);
    const int64_t npix = H * W;
    double *dens = (double *)malloc(sizeof(double) * (size_t)npix);
    int64_t *parent = (int64_t *)malloc(sizeof(int64_t) * (size_t)npix);
    double *dist_parent = (double *)calloc((size_t)npix, sizeof(double));
    if (!dens || !parent || !dist_parent) { free(dens); free(parent); free(dist_parent); return OBIA_ENOMEM; }
    for (int64_t r = 0; r < H; ++r)
        for (int64_t c = 0; c < W; ++c) {
            int64_t r0 = r - kw > 0 ? r - kw : 0, r1 = r + kw + 1 < H ? r + kw + 1 : H;
            int64_t c0 = c - kw > 0 ? c - kw : 0, c1 = c + kw + 1 < W ? c + kw + 1 : W;
            const double *cur = image + (r * W + c) * C;
            double acc = 0.0;
            for (int64_t r_ = r0; r_ < r1; ++r_)
                for (int64_t c_ = c0; c_ < c1; ++c_) {
                    double dist = 0.0;
                    const double *o = image + (r_ * W + c_) * C;
                    for (int ch = 0; ch < C; ++ch) { double t = cur[ch] - o[ch]; dist += t * t; }
                    double tr = (double)(r - r_), tc = (double)(c - c_);
                    dist += tr * tr;
                    dist += tc * tc;
                    acc += exp(dist * inv_ks2);
                }
            dens[r * W + c] = acc;
        }
    for (int64_t i = 0; i < npix; ++i) { dens[i] += noise[i]; parent[i] = i; }
    for (int64_t r = 0; r < H; ++r)
        for (int64_t c = 0; c < W; ++c) {
            int64_t r0 = r - kw > 0 ? r - kw : 0, r1 = r + kw + 1 < H ? r + kw + 1 : H;
            int64_t c0 = c - kw > 0 ? c - kw : 0, c1 = c + kw + 1 < W ? c + kw + 1 : W;
            const double *cur = image + (r * W + c) * C;
            const double cd = dens[r * W + c];
            double closest = INFINITY;
            for (int64_t r_ = r0; r_ < r1; ++r_)
                for (int64_t c_ = c0; c_ < c1; ++c_) {
                    if (dens[r_ * W + c_] > cd) {
                        double dist = 0.0;
                        const double *o = image + (r_ * W + c_) * C;
                        for (int ch = 0; ch < C; ++ch) { double t = cur[ch] - o[ch]; dist += t * t; }
                        double tr = (double)(r - r_), tc = (double)(c - c_);
                        dist += tr * tr;
                        dist += tc * tc;
                        if (dist < closest) { closest = dist; parent[r * W + c] = r_ * W + c_; }
                    }
                }
            dist_parent[r * W + c] = sqrt(closest);
        }
    /* _quickshift.py / pyx tail: cut links longer than max_dist, flatten, relabel by root order */
    for (int64_t i = 0; i < npix; ++i) if (dist_parent[i] > max_dist) parent[i] = i;
    for (;;) {
        int changed = 0;
        for (int64_t i = 0; i < npix; ++i) {
            int64_t p = parent[parent[i]];
            if (p != parent[i]) { parent[i] = p; changed = 1; }
        }
        if (!changed) break;
    }
    /* np.unique(parent, return_inverse=True)[1]: consecutive ids by ascending root index */
    int64_t *rank = (int64_t *)malloc(sizeof(int64_t) * (size_t)npix);
    if (!rank) { free(dens); free(parent); free(dist_parent); return OBIA_ENOMEM; }
    int64_t n = 0;
    for (int64_t i = 0; i < npix; ++i) rank[i] = -1;
    for (int64_t i = 0; i < npix; ++i) rank[parent[i]] = 0;
    for (int64_t i = 0; i < npix; ++i) if (rank[i] == 0) rank[i] = n++;
    for (int64_t i = 0; i < npix; ++i) labels_out[i] = rank[parent[i]];
    free(rank); free(dens); free(parent); free(dist_parent);
    return OBIA_OK;
}
