// slic.hip -- batched SLIC engine for gfx950: feature preparation, seeding, centroid binning and the
// pixel-centric assign + fused-accumulate sweep.
//
// What it restates (all third-party arithmetic reached from obia/segmentation/segment_boundaries.py:48-51):
//   normalize_band                     segment_boundaries.py:11-16,32-33
//   slic() driver                      skimage slic_superpixels.py:107-333
//   _slic_cython assign/update loop    skimage _slic.pyx (0.18.3), see oracle/obia_oracle.c
// The reference loop is segment-centric (each centroid scatters into its (4S+1)^2 window, ties to the
// lowest k).  Here it is pixel-centric: centroids are binned by their CURRENT position every sweep, a
// workgroup owning a 32x32 pixel tile stages into LDS exactly the centroids whose window intersects
// the tile, and every lane takes the lexicographic minimum of (distance, k) over the candidates whose
// window contains its pixel -- the same candidate set and the same tie rule.  Distances use the
// reference's operation order in float32 with contraction off, so they are bit-equal to the x86
// build.  The centroid update is fused into the sweep: per-lane run sums -> LDS partials -> one
// global integer atomic per (tile, centroid, field); colour sums are 64-bit fixed point, so the sums
// (and therefore the whole segmentation) do not depend on the order of the atomics.
#include "slic.hpp"

#include <cmath>
#include <cstdlib>

namespace obia {

// ------------------------------------------------------------------------------------------------
// host: regular_grid((1,H,W), n)  -- skimage/util/_regular_grid.py:61-83
// ------------------------------------------------------------------------------------------------
void regular_grid_hw(long long H, long long W, long long n, long long out[4]) {
    long long dims[3] = {1, H, W};
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (dims[order[j]] < dims[order[i]]) std::swap(order[i], order[j]);
    double sd[3];
    for (int i = 0; i < 3; ++i) sd[i] = (double)dims[order[i]];
    double space = sd[0] * sd[1] * sd[2];
    if (space <= (double)n) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    double st[3];
    for (int i = 0; i < 3; ++i) st[i] = std::pow(space / (double)n, 1.0 / 3.0);
    auto all_ge = [&]() { return sd[0] >= st[0] && sd[1] >= st[1] && sd[2] >= st[2]; };
    if (!all_ge()) {
        for (int d = 0; d < 3; ++d) {
            st[d] = sd[d];
            double sp = 1.0;
            for (int e = d + 1; e < 3; ++e) sp *= sd[e];
            if (d < 2) {
                double v = std::pow(sp / (double)n, 1.0 / (double)(3 - d - 1));
                for (int e = d + 1; e < 3; ++e) st[e] = v;
            }
            if (all_ge()) break;
        }
    }
    long long start[3], step[3], s_of[3], t_of[3];
    for (int i = 0; i < 3; ++i) {
        start[i] = (long long)std::floor(st[i] / 2.0);
        step[i] = (long long)std::nearbyint(st[i]);   // np.round: half to even
    }
    for (int i = 0; i < 3; ++i) { s_of[order[i]] = start[i]; t_of[order[i]] = step[i]; }
    out[0] = s_of[1]; out[1] = t_of[1]; out[2] = s_of[2]; out[3] = t_of[2];
}

static long long slice_len(long long L, long long start, long long step) {
    if (step == 0) return L;
    if (start >= L) return 0;
    return (L - start + step - 1) / step;
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2key(float f) {   // order-preserving float -> uint
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

constexpr int FP_NT = 256;

// K0a: per-band min/max of each problem window.  A row of a band-interleaved window is a flat run of
// w*C floats; lanes read it as coalesced float4 (C % 4 == 0) or dwords, and the thread count in use is a
// multiple of the band period so that every thread owns a fixed band (group).  grid = (blocks, nprob).
// Global atomics that land on a handful of addresses cost several ns EACH, device-wide (measured: 35k atomicMax on one
// word add 180 us to a 240-us pass).  A running min / max only needs the atomic when it improves the published value.
__device__ __forceinline__ void lazy_atomic_min(unsigned *a, unsigned v) {
    if (v < __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(a, v);
}
__device__ __forceinline__ void lazy_atomic_max(unsigned *a, unsigned v) {
    if (v > __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a, v);
}

template <int VEC>
__global__ __launch_bounds__(FP_NT) void band_minmax_kernel(const float *__restrict__ src, int Ws, int C,
                                                            const SrcWindow *__restrict__ wins,
                                                            unsigned *__restrict__ keys /*[nprob][C][2]*/,
                                                            int *__restrict__ nonfinite) {
    __shared__ unsigned s_mn[32], s_mx[32];
    const int p = blockIdx.y;
    const SrcWindow wdw = wins[p];
    const int period = C / VEC;                 // threads per pixel
    const int active = (FP_NT / period) * period;
    const int tid = threadIdx.x;
    if (tid < 32) { s_mn[tid] = 0xffffffffu; s_mx[tid] = 0u; }
    __syncthreads();
    float lo[VEC], hi[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { lo[v] = INFINITY; hi[v] = -INFINITY; }
    bool bad = false;
    if (tid < active) {
        const long long row_vecs = (long long)wdw.w * period;
        for (int y = blockIdx.x; y < wdw.h; y += gridDim.x) {
            const float *row = src + ((long long)(wdw.y0 + y) * Ws + wdw.x0) * C;
            // four loads in flight per lane (a lane keeps its bands: the stride `active` is a multiple of the period)
            constexpr int MU = 4;
            for (long long e0 = tid; e0 < row_vecs; e0 += (long long)MU * active) {
                float v[MU][VEC];
#pragma unroll
                for (int u = 0; u < MU; ++u) {
                    const long long e = e0 + (long long)u * active;
                    if (e < row_vecs) {
                        if (VEC == 4) {
                            const float4 t = *reinterpret_cast<const float4 *>(row + 4 * e);
                            v[u][0] = t.x; v[u][1 % VEC] = t.y; v[u][2 % VEC] = t.z; v[u][3 % VEC] = t.w;
                        } else v[u][0] = row[e];
                    } else {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) v[u][q] = lo[q];   // neutral: already inside [lo, hi] or +inf (ignored below)
                    }
                }
#pragma unroll
                for (int u = 0; u < MU; ++u) {
                    const bool live = e0 + (long long)u * active < row_vecs;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        if (live) bad |= !(fabsf(v[u][q]) <= 3.4028234e38f);
                        lo[q] = fminf(lo[q], v[u][q]);
                        if (live) hi[q] = fmaxf(hi[q], v[u][q]);
                    }
                }
            }
        }
        const int band0 = (tid % period) * VEC;
#pragma unroll
        for (int q = 0; q < VEC; ++q)
            if (lo[q] <= hi[q]) {
                atomicMin(&s_mn[band0 + q], f2key(lo[q]));
                atomicMax(&s_mx[band0 + q], f2key(hi[q]));
            }
        if (bad && __hip_atomic_load(&nonfinite[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&nonfinite[p], 1);
    }
    __syncthreads();
    if (tid < C) {
        if (s_mn[tid] != 0xffffffffu) atomicMin(&keys[((long long)p * C + tid) * 2 + 0], s_mn[tid]);
        if (s_mx[tid] != 0u) atomicMax(&keys[((long long)p * C + tid) * 2 + 1], s_mx[tid]);
    }
}

// skimage.color.rgb2lab on float32 (colorconv.py rgb2xyz + xyz2lab, D65 / 2 degree observer).
__device__ __forceinline__ void rgb2lab_f32(float r, float g, float b, float &L, float &A, float &B) {
    float a[3] = {r, g, b};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = a[c];
        a[c] = (v > 0.04045f) ? powf((v + 0.055f) / 1.055f, 2.4f) : v / 12.92f;
    }
    const float m[3][3] = {{0.412453f, 0.357580f, 0.180423f},
                           {0.212671f, 0.715160f, 0.072169f},
                           {0.019334f, 0.119193f, 0.950227f}};
    const float wr[3] = {0.95047f, 1.0f, 1.08883f};
    float xyz[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float s = a[0] * m[i][0];
        s = s + a[1] * m[i][1];
        s = s + a[2] * m[i][2];
        s = s / wr[i];
        xyz[i] = (s > 0.008856f) ? cbrtf(s) : 7.787f * s + 16.0f / 116.0f;
    }
    L = 116.0f * xyz[1] - 16.0f;
    A = 500.0f * (xyz[0] - xyz[1]);
    B = 200.0f * (xyz[1] - xyz[2]);
}

// one pixel: [normalize_band] -> [rgb2lab] -> * ratio; returns max |feature|
template <int CP>
__device__ __forceinline__ float feature_pixel(const float *__restrict__ px, int C, bool vec, int normalize, int to_lab, float ratio,
                                               const float (&bmn)[CP], const float (&bden)[CP], float (&v)[CP]) {
    if (vec) {
#pragma unroll
        for (int q = 0; q < CP / 4; ++q) {
            const float4 t = reinterpret_cast<const float4 *>(px)[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int c = 0; c < CP; ++c) v[c] = (c < C) ? px[c] : 0.0f;
    }
    if (normalize) {
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            float t = (v[c] - bmn[c]) / bden[c];
            if (!(fabsf(t) <= 3.0e38f)) t = 0.0f;   // constant / non-finite band: the problem is rejected on the host
            v[c] = t;
        }
    } else {
#pragma unroll
        for (int c = 0; c < CP; ++c) if (!(fabsf(v[c]) <= 3.0e38f)) v[c] = 0.0f;
    }
    if (to_lab) {
        float L, A, B;
        rgb2lab_f32(v[0], v[1], v[2], L, A, B);
        v[0] = L; v[1] = A; v[2] = B;
    }
    float m = 0.0f;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        v[c] = v[c] * ratio;
        m = fmaxf(m, fabsf(v[c]));
    }
    return m;
}

template <int CP>
__device__ __forceinline__ void feature_band_params(const unsigned *__restrict__ keys, int p, int C, int normalize, float (&bmn)[CP],
                                                    float (&bden)[CP]) {
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        bmn[c] = 0.0f; bden[c] = 1.0f;
        if (normalize && c < C) {
            const float mn = key2f(keys[((long long)p * C + c) * 2 + 0]);
            const float mx = key2f(keys[((long long)p * C + c) * 2 + 1]);
            bmn[c] = mn; bden[c] = mx - mn;     // (band - min) / (max - min), segment_boundaries.py:16
        }
    }
}

// K0b: features = [normalize_band] -> [rgb2lab] -> * float32(1/compactness), padded to CP channels (padding is 0:
// `t = 0 - 0; dc += t*t` leaves every distance bit-identical).  Also reduces max|feature| for the fixed-point scale.
// Per-band min and (max - min) are hoisted into registers; C % 4 == 0 rasters are read as float4.
// Pixel-major output [pixel][CP] (quickshift reads its features per pixel).
template <int CP>
__global__ __launch_bounds__(256) void features_kernel(const float *__restrict__ src, int Ws, int C,
                                                       const SrcWindow *__restrict__ wins,
                                                       const unsigned *__restrict__ keys, int normalize,
                                                       int to_lab, float ratio, float *__restrict__ feat,
                                                       unsigned *__restrict__ maxabs_bits) {
    const int p = blockIdx.y;
    const SrcWindow wdw = wins[p];
    float bmn[CP], bden[CP];
    feature_band_params<CP>(keys, p, C, normalize, bmn, bden);
    const bool vec = (C == CP);                 // C % 4 == 0: aligned float4 reads
    float local_max = 0.0f;
    // blocks walk rows, threads walk the pixels of a row: no integer division per pixel
    for (int y = blockIdx.x; y < wdw.h; y += gridDim.x)
    for (int x = threadIdx.x; x < wdw.w; x += blockDim.x) {
        const long long i = (long long)y * wdw.w + x;
        const float *px = src + ((long long)(wdw.y0 + y) * Ws + wdw.x0 + x) * C;
        float v[CP];
        local_max = fmaxf(local_max, feature_pixel<CP>(px, C, vec, normalize, to_lab, ratio, bmn, bden, v));
        float4 *dst = reinterpret_cast<float4 *>(feat + (wdw.pix_off + i) * CP);
#pragma unroll
        for (int q = 0; q < CP / 4; ++q) dst[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
    // wave max -> one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local_max = fmaxf(local_max, __shfl_xor(local_max, off));
    if ((threadIdx.x & 63) == 0) lazy_atomic_max(maxabs_bits + p, __float_as_uint(local_max));   // one word per window
}

// The same arithmetic, written as quad-row blocks (slic.hpp: feat_block_f4): a thread owns the four pixels
// (4q .. 4q+3, x), reads them row by row (each row coalesced across the wave) and writes one float4 per channel
// (sixteen consecutive threads -> 256 contiguous bytes of a channel run).  Blocks walk quad rows.
// BOX: the colour boxes of the sweep's footprints (slic.hpp: feat_boxes; low compactness only) come out of the same pass -- a
// block then walks footprint bands (four quad rows), a thread keeps the lo / hi of its column over the band and the sixteen
// lanes of a footprint fold them by shuffles: no second pass over the planes (it cost 1.6 ms per step at C3).
template <int CP, bool BOX>
__global__ __launch_bounds__(256) void features_planes_kernel(const float *__restrict__ src, int Ws, int C,
                                                              const SrcWindow *__restrict__ wins,
                                                              const unsigned *__restrict__ keys, int normalize,
                                                              int to_lab, float ratio, float *__restrict__ feat,
                                                              unsigned *__restrict__ maxabs_bits, float *__restrict__ fbox, int dense) {
    // dense: `src` is not the caller's raster but the per-window [h][w][CP] arrays of the smoothing passes (features_kernel's layout)
    const int p = blockIdx.y;
    const SrcWindow wdw = wins[p];
    float bmn[CP], bden[CP];
    feature_band_params<CP>(keys, p, C, normalize, bmn, bden);
    const bool vec = (C == CP);
    float local_max = 0.0f;
    const int QH = (wdw.h + 3) >> 2, XB = (wdw.w + 15) >> 4;
    float4 *__restrict__ planes = reinterpret_cast<float4 *>(feat) + wdw.feat_off;
    // one quad row of one column: features of the four pixels, stored as one float4 per channel
    auto quad = [&](int q, int x, float (&v)[4][CP]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = 4 * q + i;
            if (y < wdw.h && x < wdw.w) {
                const float *px = dense ? src + (wdw.pix_off + (long long)y * wdw.w + x) * CP
                                        : src + ((long long)(wdw.y0 + y) * Ws + wdw.x0 + x) * C;
                local_max = fmaxf(local_max, feature_pixel<CP>(px, C, vec, normalize, to_lab, ratio, bmn, bden, v[i]));
            } else {
#pragma unroll
                for (int c = 0; c < CP; ++c) v[i][c] = 0.0f;
            }
        }
        float4 *dst = planes + ((long long)q * XB + (x >> 4)) * (CP * 16) + (x & 15);
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) dst[c * 16] = make_float4(v[0][c], v[1][c], v[2][c], v[3][c]);   // (the planes of padded channels are neither written nor read)
    };
    if (!BOX) {
        for (int q = blockIdx.x; q < QH; q += gridDim.x)
            for (int x = threadIdx.x; x < 16 * XB; x += blockDim.x) {
                float v[4][CP];
                quad(q, x, v);
            }
    } else {
        const int FH = (QH + 3) >> 2;   // footprint bands
        for (int fy = blockIdx.x; fy < FH; fy += gridDim.x)
            for (int x = threadIdx.x; x < 16 * XB; x += blockDim.x) {   // (whole 16-lane groups: 16 * XB is a multiple of 16)
                float lo[CP], hi[CP];
#pragma unroll
                for (int c = 0; c < CP; ++c) { lo[c] = INFINITY; hi[c] = -INFINITY; }
                for (int qq = 0; qq < 4; ++qq) {
                    const int q = 4 * fy + qq;
                    if (q >= QH) break;   // block-uniform
                    float v[4][CP];
                    quad(q, x, v);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (4 * q + i < wdw.h && x < wdw.w) {   // pixels outside the window are left out of the box
#pragma unroll
                            for (int c = 0; c < CP; ++c) { lo[c] = fminf(lo[c], v[i][c]); hi[c] = fmaxf(hi[c], v[i][c]); }
                        }
                }
#pragma unroll
                for (int c = 0; c < CP; ++c)
#pragma unroll
                    for (int off = 8; off > 0; off >>= 1) {
                        lo[c] = fminf(lo[c], __shfl_xor(lo[c], off));
                        hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], off));
                    }
                if ((x & 15) == 0) {
                    float *o = fbox + (wdw.fb_off + (long long)fy * XB + (x >> 4)) * (2 * CP);
#pragma unroll
                    for (int c = 0; c < CP; ++c) { o[c] = lo[c]; o[CP + c] = hi[c]; }
                }
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local_max = fmaxf(local_max, __shfl_xor(local_max, off));
    if ((threadIdx.x & 63) == 0) lazy_atomic_max(maxabs_bits + p, __float_as_uint(local_max));
}

// The colour-box bound pays when the colour term decides, i.e. when the features (after `* 1 / compactness`) are large against
// the spatial term: measured break-even near compactness 1 on [0, 1] features (13.6 -> 7.2 visits per footprint at 0.25,
// 5.1 -> 4.8 at 1, none at 10).  Lab features span ~100 units where normalised bands span 1: a three-band raster at compactness
// 10 is as colour-dominated as eight bands at 0.1 (`bench.py --bands 3`: 462 us per sweep launch without the bound).
bool slic_use_colour_bound(float ratio, bool to_lab) {
    static const char *env = std::getenv("OBIA_COLOUR_BOUND");   // developer switch (A/B timing)
    if (env) return env[0] == '1';
    return ratio * (to_lab ? 100.0f : 1.0f) >= 2.0f;
}

__global__ void keys_init_kernel(unsigned *keys, int nkeys, int ntot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ntot) keys[i] = (i < nkeys && (i & 1) == 0) ? 0xffffffffu : 0u;
}


// ---- Gaussian pre-smoothing (slic(..., sigma=...)) ---------------------------------------------------------------------------
// One axis pass of scipy.ndimage.correlate1d with symmetric weights and mode 'reflect' (d c b a | a b c d | d c b a, repeated), as
// gaussian_filter applies it per axis: the line is read as double,  tmp = line[i] * w[0];  for j = r .. 1:
// tmp += (line[i - j] + line[i + j]) * w[j]  (outermost pair first), the result stored in float32.  Per-window dense arrays
// [h][w][CP]; AXIS 0 is the depth axis of the (1, H, W, C) image slic() builds: one plane, every tap reads the pixel itself.
template <int AXIS>
__global__ __launch_bounds__(256) void gauss_axis_kernel(const SrcWindow *__restrict__ wins, const float *__restrict__ in,
                                                         float *__restrict__ out, int CP, const double *__restrict__ w, int r) {
    const SrcWindow wdw = wins[blockIdx.y];
    const long long n_el = (long long)wdw.h * wdw.w * CP;
    const float *a = in + wdw.pix_off * CP;
    float *o = out + wdw.pix_off * CP;
    const int n = AXIS == 1 ? wdw.h : (AXIS == 2 ? wdw.w : 1);
    const long long stride = AXIS == 1 ? (long long)wdw.w * CP : (long long)CP;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (long long)gridDim.x * blockDim.x) {
        const int pos = AXIS == 1 ? (int)(i / stride) : (AXIS == 2 ? (int)((i / CP) % wdw.w) : 0);
        const float *line = a + (i - (long long)pos * stride);   // element 0 of this element's line
        double tmp = (double)a[i] * w[0];
        for (int j = r; j >= 1; --j) {
            int lo = pos - j, hi = pos + j;
            if (AXIS == 0) { lo = 0; hi = 0; }
            else {
                const int per = 2 * n;
                lo %= per; if (lo < 0) lo += per; if (lo >= n) lo = per - 1 - lo;
                hi %= per; if (hi >= n) hi = per - 1 - hi;
            }
            tmp += ((double)line[(long long)lo * stride] + (double)line[(long long)hi * stride]) * w[j];
        }
        o[i] = (float)tmp;
    }
}

// NumPy's float64 pairwise summation (blocks of eight accumulators up to 128 elements, halves above): the weights are divided by
// `phi.sum()`, and the order of the additions decides its last bit
static double np_pairwise_sum(const double *a, size_t n) {
    if (n < 8) { double r = 0.0; for (size_t i = 0; i < n; ++i) r += a[i]; return r; }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        size_t i = 8;
        for (; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// scipy _gaussian_kernel1d: exp(-0.5 / sigma^2 * x^2), x = -r..r, r = int(4 sigma + 0.5), divided by the NumPy sum; w[0] = centre.
// sigma_is_f32: slic() hands scipy a float32 scalar (the image's dtype) and scipy squares it in float32 before everything else runs in
// double; quickshift() hands it the caller's Python number.
int gaussian_weights_host(double sigma, bool sigma_is_f32, std::vector<double> &w) {
    const float s32 = (float)sigma;
    const double sd = sigma_is_f32 ? (double)s32 : sigma;
    const int r = (int)(4.0 * sd + 0.5);
    const double sigma2 = sigma_is_f32 ? (double)(s32 * s32) : sigma * sigma;
    std::vector<double> phi((size_t)2 * r + 1);
    for (int x = -r; x <= r; ++x) phi[(size_t)(x + r)] = std::exp(-0.5 / sigma2 * (double)((long long)x * x));
    const double sum = np_pairwise_sum(phi.data(), phi.size());
    w.resize((size_t)r + 1);
    for (int j = 0; j <= r; ++j) w[(size_t)j] = phi[(size_t)(r + j)] / sum;
    return r;
}

// scratch arrays and the three weight tables
int smooth_prepare(obia_ctx *ctx, SmoothSpec &sm, long long total_pix, long long maxpix, int CP, int np) {
    if (!sm.on()) return OBIA_OK;
    Arena &A = ctx->arena;
    sm.tmp_a = A.get<float>((size_t)total_pix * CP);
    sm.tmp_b = A.get<float>((size_t)total_pix * CP);
    sm.d_scratch = A.get<unsigned>((size_t)np);
    sm.maxpix = maxpix;
    if (!sm.tmp_a || !sm.tmp_b || !sm.d_scratch) return OBIA_E_NOMEM;
    for (int ax = 0; ax < 3; ++ax) {
        if (!(sm.sigma[ax] > 1e-15)) continue;
        if (!(sm.sigma[ax] < 1.0e6)) { set_error("sigma %g not supported", sm.sigma[ax]); return OBIA_E_INVALID; }
        std::vector<double> w;
        const int r = gaussian_weights_host(sm.sigma[ax], true, w);
        sm.radius[ax] = r;
        sm.d_w[ax] = A.get<double>((size_t)r + 1);
        if (!sm.d_w[ax]) return OBIA_E_NOMEM;
        OBIA_TRY(upload_async(ctx, sm.d_w[ax], w.data(), sizeof(double) * w.size()));
    }
    return OBIA_OK;
}

// Launch half of the feature pass on `stream`: min / max of every band of every window, then the features.
// d_keys layout for np windows: keys[np][C][2] (min, max as ordered uints) | nonfinite[np] | max|feature| bits [np].
int slic_features_launch(hipStream_t stream, int C, int CP, int np, const SrcWindow *d_windows, int maxh, const float *src, int Ws,
                         int normalize, int to_lab, float ratio, float *d_feat, unsigned *d_keys, bool planes, float *d_fbox,
                         const SmoothSpec *smooth) {
    if (C < 1 || C > 16) { set_error("band count %d not supported (1..16)", C); return OBIA_E_UNSUPPORTED; }
    const size_t nkeys = (size_t)np * C * 2, ntot = nkeys + 2 * (size_t)np;
    unsigned *d_nonfinite = d_keys + nkeys, *d_maxabs = d_nonfinite + np;
    // (min, max) key pairs start at (0xffffffff, 0), the flags at 0: initialised on the device, no host round trip
    hipLaunchKernelGGL(keys_init_kernel, dim3(cdiv((long long)ntot, 256)), dim3(256), 0, stream, d_keys, (int)nkeys, (int)ntot);
    // (Window by window -- min / max pass and feature pass of one 134-MB window back to back, so that the second read could come from
    // the 256-MB Infinity Cache -- was measured in round 4: features 2.46 -> 3.8 ms per step, step +2.4 ms; the batch-wide passes stay.)
    if (normalize) {
        int gx = maxh < 2048 ? maxh : 2048;
        // float4 reads need 16-byte aligned rows: C % 4 == 0 and an aligned base pointer
        if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(band_minmax_kernel<4>), dim3(gx, np), dim3(FP_NT), 0, stream, src, Ws, C, d_windows,
                               d_keys, (int *)d_nonfinite);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(band_minmax_kernel<1>), dim3(gx, np), dim3(FP_NT), 0, stream, src, Ws, C, d_windows,
                               d_keys, (int *)d_nonfinite);
    }
    const bool box = planes && d_fbox != nullptr;   // colour boxes from the same pass (a block walks footprint bands of 16 rows)
    const int rows = box ? (maxh + 15) / 16 : (planes ? (maxh + 3) / 4 : maxh);
    dim3 grid(rows < 4096 ? rows : 4096, np);
    // Smoothing: normalise -> Lab (unscaled, pixel-major) -> the Gaussian passes -> scale + plane layout from the smoothed arrays
    const bool smoothing = smooth && smooth->on();
    if (smoothing && !planes) { set_error("Gaussian pre-smoothing is built for the SLIC feature layout only"); return OBIA_E_UNSUPPORTED; }
    const float *planes_src = src;
    int planes_norm = normalize, planes_lab = to_lab, dense = 0;
    if (smoothing) {
        OBIA_HIP_TRY(hipMemsetAsync(smooth->d_scratch, 0, sizeof(unsigned) * (size_t)np, stream));
        dim3 g1(maxh < 4096 ? maxh : 4096, np);
#define LAUNCH_UNSCALED(CPV) hipLaunchKernelGGL(HIP_KERNEL_NAME(features_kernel<CPV>), g1, dim3(256), 0, stream, src, Ws, C, d_windows, \
                                                d_keys, normalize, to_lab, 1.0f, smooth->tmp_a, smooth->d_scratch)
        switch (CP) {
            case 4: LAUNCH_UNSCALED(4); break;
            case 8: LAUNCH_UNSCALED(8); break;
            case 12: LAUNCH_UNSCALED(12); break;
            default: LAUNCH_UNSCALED(16); break;
        }
#undef LAUNCH_UNSCALED
        float *cur = smooth->tmp_a, *oth = smooth->tmp_b;
        long long nb = (smooth->maxpix * CP + 255) / 256;
        dim3 gg((unsigned)(nb < 65535 ? (nb < 1 ? 1 : nb) : 65535), np);
        for (int ax = 0; ax < 3; ++ax) {
            if (!(smooth->sigma[ax] > 1e-15)) continue;
            if (ax == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(gauss_axis_kernel<0>), gg, dim3(256), 0, stream, d_windows, cur, oth, CP, smooth->d_w[0], smooth->radius[0]);
            else if (ax == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(gauss_axis_kernel<1>), gg, dim3(256), 0, stream, d_windows, cur, oth, CP, smooth->d_w[1], smooth->radius[1]);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(gauss_axis_kernel<2>), gg, dim3(256), 0, stream, d_windows, cur, oth, CP, smooth->d_w[2], smooth->radius[2]);
            std::swap(cur, oth);
        }
        planes_src = cur; planes_norm = 0; planes_lab = 0; dense = 1;
    }
#define LAUNCH_FEAT(CPV)                                                                                                  \
    do {                                                                                                                  \
        if (box) hipLaunchKernelGGL(HIP_KERNEL_NAME(features_planes_kernel<CPV, true>), grid, dim3(256), 0, stream, planes_src, Ws, C, \
                                    d_windows, d_keys, planes_norm, planes_lab, ratio, d_feat, d_maxabs, d_fbox, dense);  \
        else if (planes) hipLaunchKernelGGL(HIP_KERNEL_NAME(features_planes_kernel<CPV, false>), grid, dim3(256), 0, stream, planes_src, Ws, C, \
                                       d_windows, d_keys, planes_norm, planes_lab, ratio, d_feat, d_maxabs, (float *)nullptr, dense); \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(features_kernel<CPV>), grid, dim3(256), 0, stream, src, Ws, C, d_windows,  \
                                d_keys, normalize, to_lab, ratio, d_feat, d_maxabs);                                      \
    } while (0)
    switch (CP) {
        case 4: LAUNCH_FEAT(4); break;
        case 8: LAUNCH_FEAT(8); break;
        case 12: LAUNCH_FEAT(12); break;
        case 16: LAUNCH_FEAT(16); break;
        default: set_error("bad CP %d", CP); return OBIA_E_INVALID;
    }
#undef LAUNCH_FEAT
    OBIA_HIP_TRY(hipGetLastError());
    return OBIA_OK;
}

// Host half: constant / non-finite bands (-> skip flags or an error) and the fixed-point scale of the batch.  `keys`,
// `nonfinite`, `maxabs` point at the read-back entries of the batch's FIRST window (np consecutive windows).
int slic_features_finish(SlicBatch &b, const unsigned *keys, const unsigned *nonfinite, const unsigned *maxabs_bits, int normalize,
                         std::vector<int> *skip) {
    const int C = b.C, np = b.nprob;
    auto k2f = [](unsigned k) { unsigned bb = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; memcpy(&f, &bb, 4); return f; };
    if (skip) skip->assign(np, 0);
    if (normalize) {
        for (int p = 0; p < np; ++p) {
            bool bad = nonfinite[p] != 0;
            int cb = -1;
            float cv = 0;
            for (int c = 0; c < C && !bad; ++c) {
                float mn = k2f(keys[((size_t)p * C + c) * 2]), mx = k2f(keys[((size_t)p * C + c) * 2 + 1]);
                if (!(mx > mn)) { cb = c; cv = mn; }
            }
            if (bad || cb >= 0) {
                if (skip) { (*skip)[p] = 1; continue; }
                if (bad) set_error("input raster holds NaN or infinite values");
                else set_error("band %d is constant (%g): normalize_band would divide 0 by 0 "
                               "(obia/segmentation/segment_boundaries.py:16)", cb, (double)cv);
                return OBIA_E_NONFINITE;
            }
        }
    }
    float maxabs = 0.0f;
    for (int p = 0; p < np; ++p) {
        float v; memcpy(&v, &maxabs_bits[p], 4);
        if (!(v <= 3.0e38f)) { set_error("non-finite feature values"); return OBIA_E_NONFINITE; }
        if (v > maxabs) maxabs = v;
    }
    // fixed-point scale for the colour sums (to_fixed32 in slic_sweep.hip): a power of two with |feature| * 2^s < 2^29, so
    // that the scaling is exact, the four pixels of a lane's strip add up in an int32 and the total over a cluster of up to
    // 2^31 pixels stays below 2^60
    int s = 0;
    if (maxabs > 0.0f) {
        int e = 0;
        (void)std::frexp((double)maxabs, &e);   // maxabs = m * 2^e, m in [0.5, 1)  =>  maxabs < 2^e
        s = 29 - e;
    }
    if (s > 100) s = 100;
    if (s < -90) s = -90;
    b.fscale = std::ldexp(1.0, s);
    return OBIA_OK;
}

float slic_prescale(float ratio, int normalize, int to_lab, bool slic_zero) {
    if (!normalize || to_lab || slic_zero || !(ratio > 0.0f) || !(ratio < 3.0e38f)) return 1.0f;
    if (const char *e = std::getenv("OBIA_NO_PRESCALE"); e && atoi(e) != 0) return 1.0f;   // developer switch (A/B timing; the labels are the same either way)
    int e = 0;
    (void)std::frexp((double)ratio, &e);   // the largest normalised feature is 1 * ratio: the rule of slic_features_finish
    int s = 29 - e;
    if (s > 40 || s < -24) return 1.0f;    // (the squared scale multiplies the spatial weight: stay far inside float32)
    return (float)std::ldexp(1.0, s);
}

int slic_prepare_features(obia_ctx *ctx, SlicBatch &b, const float *src, int Hs, int Ws, int normalize,
                          int to_lab, float ratio, std::vector<int> *skip) {
    (void)Hs;
    ScopedSpan span(ctx, T_FEAT);
    const int C = b.C, np = b.nprob;
    const size_t nkeys = (size_t)np * C * 2, ntot = nkeys + 2 * (size_t)np;
    unsigned *d_keys = ctx->arena.get<unsigned>(ntot);
    if (!d_keys) return OBIA_E_NOMEM;
    int maxh = 1;
    long long maxpix = 1;
    for (auto &w : b.windows) { if (w.h > maxh) maxh = w.h; if ((long long)w.h * w.w > maxpix) maxpix = (long long)w.h * w.w; }
    SmoothSpec sm;
    for (int i = 0; i < 3; ++i) sm.sigma[i] = b.sigma[i];
    OBIA_TRY(smooth_prepare(ctx, sm, b.total_pix, maxpix, b.CP, np));
    OBIA_TRY(slic_features_launch(ctx->stream, C, b.CP, np, b.d_windows, maxh, src, Ws, normalize, to_lab, ratio, b.d_feat, d_keys,
                                  b.feat_planes, (b.col_lb && b.feat_planes) ? b.d_fbox : nullptr, &sm));
    // one read-back: min/max keys (constant-band check), non-finite flags, max|feature| per window
    std::vector<unsigned> host(ntot);
    OBIA_TRY(read_back(ctx, host.data(), d_keys, ntot * sizeof(unsigned)));
    return slic_features_finish(b, host.data(), host.data() + nkeys, host.data() + nkeys + np, normalize, skip);
}

// ------------------------------------------------------------------------------------------------
// seeding
// ------------------------------------------------------------------------------------------------
struct SeedGrid { int start_y, step_y, ny, start_x, step_x, nx; int cent_off; int slots; };   // slots: centroid records reserved for the problem (0: empty problem)

// unmasked: centroid k of problem p sits on the regular grid (slic_superpixels.py:71-104)
__global__ void seed_grid_kernel(const SeedGrid *__restrict__ grids, int nprob, float *__restrict__ seed,
                                 int *__restrict__ cent_prob) {
    const int p = blockIdx.y;
    const SeedGrid g = grids[p];
    const int K = g.ny * g.nx;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
        const int iy = k / g.nx, ix = k % g.nx;
        seed[2 * (size_t)(g.cent_off + k)] = (float)(g.start_y + iy * g.step_y);
        seed[2 * (size_t)(g.cent_off + k) + 1] = (float)(g.start_x + ix * g.step_x);
        cent_prob[g.cent_off + k] = p;
    }
}

// masked: keep the grid points that fall on valid pixels, in row-major order (DESIGN.md "masked-grid
// seeding"); if none does, seed the first valid pixel.  One workgroup per problem.
__global__ __launch_bounds__(256) void seed_masked_kernel(const SeedGrid *__restrict__ grids,
                                                          const SlicProblem *__restrict__ probs,
                                                          const uint8_t *__restrict__ mask, float *__restrict__ seed,
                                                          int *__restrict__ cent_prob, int *__restrict__ K_out) {
    constexpr int U = 4;                      // grid points per thread and step: four mask bytes in flight (the step is one
    __shared__ int s_wave[2][U][4];           // memory round trip long), kept counts per (sub-chunk, wave), double-buffered by step
    __shared__ unsigned long long s_first;
    const int p = blockIdx.x;
    const SeedGrid g = grids[p];
    const SlicProblem P = probs[p];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_first = ~0ull;
    const int Kg = g.ny * g.nx;
    int base = 0;                             // seeds kept so far (the same in every thread: no shared counter, one barrier per step)
    for (int k0 = 0, it = 0; k0 < Kg; k0 += U * 256, ++it) {
        int y[U], x[U];
        bool keep[U];
        unsigned long long bal[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 256 + tid;
            y[u] = x[u] = 0;
            keep[u] = false;
            if (k < Kg) {
                y[u] = g.start_y + (k / g.nx) * g.step_y;
                x[u] = g.start_x + (k % g.nx) * g.step_x;
                keep[u] = mask[P.pix_off + (long long)y[u] * P.W + x[u]] != 0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bal[u] = __ballot(keep[u]);
            if (lane == 0) s_wave[it & 1][u][wv] = __popcll(bal[u]);
        }
        __syncthreads();
        int run = base;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const int c = s_wave[it & 1][u][w]; total += c; if (w < wv) before += c; }
            if (keep[u]) {
                const int rank = run + before + __popcll(bal[u] & ((1ull << lane) - 1ull));
                seed[2 * (size_t)(g.cent_off + rank)] = (float)y[u];
                seed[2 * (size_t)(g.cent_off + rank) + 1] = (float)x[u];
                cent_prob[g.cent_off + rank] = p;
            }
            run += total;
        }
        base = run;
    }
    __syncthreads();
    int K = base;
    // fallback seed (no grid point is valid): only for a problem that HAS a record reserved.  An "empty" problem -- no valid
    // pixel, or so few that n_segments rounds to zero -- reserved none: writing its fallback seed went one element past
    // d_seed / d_cent_prob (found by tests/test_gpu_tiling_random.py: a 35 x 53 window with 55 valid pixels).
    if (K == 0 && g.slots > 0) {
        const long long npix = (long long)P.H * P.W;
        unsigned long long best = ~0ull;
        for (long long i = tid; i < npix; i += 256)
            if (mask[P.pix_off + i]) { best = (unsigned long long)i; break; }
        atomicMin(&s_first, best);
        __syncthreads();
        if (tid == 0 && s_first != ~0ull) {
            seed[2 * (size_t)g.cent_off] = (float)(s_first / P.W);
            seed[2 * (size_t)g.cent_off + 1] = (float)(s_first % P.W);
            cent_prob[g.cent_off] = p;
        }
        K = (s_first != ~0ull) ? 1 : 0;
    }
    if (tid == 0) K_out[p] = K;
}

// valid-pixel count per problem (mask.sum(), tiling.py:133 / slic_superpixels.py:322); 16 mask bytes per load
__global__ __launch_bounds__(256) void count_valid_kernel(const SlicProblem *__restrict__ probs,
                                                          const uint8_t *__restrict__ mask, int *__restrict__ out) {
    const int p = blockIdx.y;
    const SlicProblem P = probs[p];
    const long long npix = (long long)P.H * P.W;
    const uint8_t *m = mask + P.pix_off;
    int c = 0;
    const long long head = ((16 - (reinterpret_cast<uintptr_t>(m) & 15)) & 15);
    const long long h0 = head < npix ? head : npix;
    const long long nvec = (npix - h0) / 16;
    const uint4 *mv = reinterpret_cast<const uint4 *>(m + h0);
    auto popnz = [](const uint4 &t) {
        int n = 0;
        const unsigned wds[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned nz = wds[q] | (wds[q] >> 4);
            nz |= nz >> 2; nz |= nz >> 1;
            n += __popc(nz & 0x01010101u);
        }
        return n;
    };
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {   // four loads in flight
        const uint4 t0 = mv[i], t1 = mv[i + stride], t2 = mv[i + 2 * stride], t3 = mv[i + 3 * stride];
        c += popnz(t0) + popnz(t1) + popnz(t2) + popnz(t3);
    }
    for (; i < nvec; i += stride) c += popnz(mv[i]);
    if (blockIdx.x == 0) {   // unaligned head and tail bytes
        for (long long j = threadIdx.x; j < h0; j += blockDim.x) c += m[j] != 0;
        for (long long j = h0 + nvec * 16 + threadIdx.x; j < npix; j += blockDim.x) c += m[j] != 0;
    }
    // one atomic per workgroup: atomics on one word serialise device-wide
    __shared__ int s_c[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        if (tot) atomicAdd(&out[p], tot);
    }
}

int slic_count_valid(obia_ctx *ctx, SlicBatch &b, std::vector<int> &nvalid) {
    const int np = b.nprob;
    Arena &A = ctx->arena;
    nvalid.assign(np, 0);
    if (!b.d_probs) b.d_probs = A.get<SlicProblem>(np);
    if (!b.d_probs) return OBIA_E_NOMEM;
    if (b.masked) {
        OBIA_TRY(upload_async(ctx, b.d_probs, b.probs.data(), sizeof(SlicProblem) * np));
        int *d_cnt = A.get<int>(np);
        if (!d_cnt) return OBIA_E_NOMEM;
        OBIA_HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int) * np, ctx->stream));
        long long maxpix = 1;
        for (auto &P : b.probs) { long long n = (long long)P.H * P.W; if (n > maxpix) maxpix = n; }
        int blocks = cdiv(maxpix, 256 * 16 * 32);
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(count_valid_kernel, dim3(blocks, np), dim3(256), 0, ctx->stream, b.d_probs, b.d_mask, d_cnt);
        OBIA_TRY(read_back(ctx, nvalid.data(), d_cnt, sizeof(int) * np));
    } else {
        for (int p = 0; p < np; ++p) nvalid[p] = b.probs[p].H * b.probs[p].W;
    }
    return OBIA_OK;
}

int slic_plan_and_seed(obia_ctx *ctx, SlicBatch &b, const std::vector<int> &n_segments, const std::vector<int> *nvalid_in,
                       const ExternalSeeds *ext) {
    const int np = b.nprob;
    if (ext && (np != 1 || ext->n < 1 || !ext->yx)) { set_error("external seeds need one raster and at least one seed"); return OBIA_E_INVALID; }
    Arena &A = ctx->arena;
    // problems carry H, W, pix_off already (set by the caller); upload a first version for the
    // counting / seeding kernels
    if (!b.d_probs) b.d_probs = A.get<SlicProblem>(np);
    if (!b.d_probs) return OBIA_E_NOMEM;
    std::vector<int> nvalid;
    if (nvalid_in) nvalid = *nvalid_in;
    else OBIA_TRY(slic_count_valid(ctx, b, nvalid));
    // caller-supplied seeds on a mask without a valid pixel: the reference raises before seeding (mask.sum() == 0).  Refused
    // here, BEFORE anything is sized or uploaded: the seed arrays below would hold one dummy record while the upload wrote
    // ext->n of them
    if (ext && nvalid[0] <= 0) { set_error("the mask has no valid pixel"); return OBIA_E_EMPTY; }
    std::vector<SeedGrid> grids(np);
    std::vector<double> stepmax(np);
    int cent_off = 0;
    for (int p = 0; p < np; ++p) {
        SlicProblem &P = b.probs[p];
        P.n_valid = nvalid[p];
        SeedGrid &g = grids[p];
        g.cent_off = cent_off; g.slots = 0;
        if (ext && nvalid[p] > 0) {   // seeds given by the caller: K = their number, step = max(steps) of the seeding
            g.start_y = g.start_x = 0; g.step_y = g.step_x = 1; g.ny = 1; g.nx = ext->n;
            stepmax[p] = ext->step < 1.0 ? 1.0 : ext->step;
            P.cent_off = cent_off;
            cent_off += ext->n;
            continue;
        }
        if (nvalid[p] <= 0 || n_segments[p] <= 0) {   // empty problem: no centroids, every pixel stays masked
            g.start_y = g.start_x = 0; g.step_y = g.step_x = 1; g.ny = g.nx = 0;
            stepmax[p] = 1.0;
            P.cent_off = cent_off;
            continue;
        }
        long long n_eff = n_segments[p];
        if (b.masked) {
            double ne = std::nearbyint((double)n_segments[p] * ((double)P.H * (double)P.W) / (double)nvalid[p]);
            n_eff = ne < 1.0 ? 1 : (long long)ne;
        }
        long long gr[4];
        regular_grid_hw(P.H, P.W, n_eff, gr);
        g.start_y = (int)gr[0]; g.step_y = gr[1] ? (int)gr[1] : 1; g.ny = (int)slice_len(P.H, gr[0], gr[1]);
        g.start_x = (int)gr[2]; g.step_x = gr[3] ? (int)gr[3] : 1; g.nx = (int)slice_len(P.W, gr[2], gr[3]);
        double sy = gr[1] ? (double)gr[1] : 1.0, sx = gr[3] ? (double)gr[3] : 1.0;
        stepmax[p] = sy > sx ? sy : sx;   // max(steps); the depth axis contributes 1.0
        if (stepmax[p] < 1.0) stepmax[p] = 1.0;
        P.cent_off = cent_off;
        long long Kg = (long long)g.ny * g.nx;
        if (Kg < 1) Kg = 1;   // masked fallback seed
        if (cent_off + Kg > 0x7fff0000LL) { set_error("too many centroids in one batch"); return OBIA_E_INVALID; }
        g.slots = (int)Kg;
        cent_off += (int)Kg;
    }
    b.total_cent = cent_off > 0 ? cent_off : 1;
    b.d_seed = A.get<float>((size_t)b.total_cent * 2);
    b.d_cent_prob = A.get<int>(b.total_cent);
    SeedGrid *d_grids = A.get<SeedGrid>(np);
    if (!b.d_seed || !b.d_cent_prob || !d_grids) return OBIA_E_NOMEM;
    OBIA_HIP_TRY(hipMemsetAsync(b.d_cent_prob, 0xff, sizeof(int) * b.total_cent, ctx->stream));
    OBIA_TRY(upload_async(ctx, d_grids, grids.data(), sizeof(SeedGrid) * np));
    std::vector<int> K(np);
    if (ext) {
        std::vector<float> hs((size_t)ext->n * 2);
        for (size_t i = 0; i < hs.size(); ++i) hs[i] = (float)ext->yx[i];   // segments.astype(float32): centroids are float
        OBIA_HIP_TRY(hipMemsetAsync(b.d_cent_prob, 0, sizeof(int) * b.total_cent, ctx->stream));
        OBIA_TRY(upload_async(ctx, b.d_seed, hs.data(), sizeof(float) * hs.size()));
        K[0] = nvalid[0] > 0 ? ext->n : 0;
    } else if (b.masked) {
        int *d_K = A.get<int>(np);
        if (!d_K) return OBIA_E_NOMEM;
        // (the descriptors slic_count_valid uploaded are still in place: the seeding reads H, W and pix_off only)
        hipLaunchKernelGGL(seed_masked_kernel, dim3(np), dim3(256), 0, ctx->stream, d_grids, b.d_probs, b.d_mask,
                           b.d_seed, b.d_cent_prob, d_K);
        OBIA_TRY(read_back(ctx, K.data(), d_K, sizeof(int) * np));
        for (int p = 0; p < np; ++p) if (nvalid[p] <= 0 || n_segments[p] <= 0) K[p] = 0;
    } else {
        int maxK = 1;
        for (int p = 0; p < np; ++p) { K[p] = grids[p].ny * grids[p].nx; if (K[p] > maxK) maxK = K[p]; }
        hipLaunchKernelGGL(seed_grid_kernel, dim3(cdiv(maxK, 256), np), dim3(256), 0, ctx->stream, d_grids, np,
                           b.d_seed, b.d_cent_prob);
    }
    // window steps, bins, tiles
    int cell_off = 0, tile_max = 0;
    long long tiles_all = 0, m4_total = 0;
    for (int p = 0; p < np; ++p) {
        SlicProblem &P = b.probs[p];
        P.K = K[p];
        long long gr[4] = {0, 0, 0, 0};
        if (P.K > 0) regular_grid_hw(P.H, P.W, P.K, gr);
        P.sy = gr[1] ? (int)gr[1] : 1;
        P.sx = gr[3] ? (int)gr[3] : 1;
        P.ncy = cdiv(P.H, P.sy);
        P.ncx = cdiv(P.W, P.sx);
        const float stepf = (float)stepmax[p];
        P.spatial_w = (float)(1.0 / ((double)stepf * (double)stepf)) * (b.prescale * b.prescale);   // (a power of two: exact)
        P.sp_y = (float)b.spacing[1]; P.sp_x = (float)b.spacing[2];   // np.ascontiguousarray(spacing, dtype=image dtype)
        P.direct = (P.sp_y != 1.0f || P.sp_x != 1.0f) ? 1 : 0;
        P.m4_off = (int)m4_total;
        m4_total += (long long)((P.H + 3) / 4) * P.W;
        P.cell_off = cell_off;
        long long nc = (long long)P.ncy * P.ncx;
        if (cell_off + nc > 0x7fff0000LL) { set_error("too many bins in one batch"); return OBIA_E_INVALID; }
        cell_off += (int)nc;
        P.tiles_x = cdiv(P.W, SWEEP_TW);
        P.tiles_y = cdiv(P.H, SWEEP_TH);
        const int nt = P.tiles_x * P.tiles_y;
        P.tile_off = (int)tiles_all;
        tiles_all += nt;
        if (nt > tile_max) tile_max = nt;
    }
    b.total_tiles_all = tiles_all > 0 ? tiles_all : 1;
    {   // tile -> problem table of the sweep's one-dimensional, XCD-aware grid (slic_sweep.hip)
        std::vector<int> tp((size_t)b.total_tiles_all, 0);
        for (int p = 0; p < np; ++p)
            for (int t = 0; t < b.probs[p].tiles_x * b.probs[p].tiles_y; ++t) tp[(size_t)b.probs[p].tile_off + t] = p;
        b.d_tile_prob = A.get<int>(tp.size());
        if (!b.d_tile_prob) return OBIA_E_NOMEM;
        OBIA_TRY(upload_async(ctx, b.d_tile_prob, tp.data(), sizeof(int) * tp.size()));   // (pinned ring: tp may go out of scope)
    }
    b.total_cells = cell_off > 0 ? cell_off : 1;
    b.total_tiles = tile_max;
    if (m4_total > 0x7fffffffLL) { set_error("mask of the batch too large"); return OBIA_E_INVALID; }
    b.total_m4 = m4_total > 0 ? m4_total : 1;
    OBIA_TRY(upload_async(ctx, b.d_probs, b.probs.data(), sizeof(SlicProblem) * np));
    const int RS = CENT_REC + b.CP;
    b.d_cent = A.get<float>((size_t)b.total_cent * RS);
    b.d_head = A.get<int>((size_t)b.total_cells * 2);
    const size_t acc_q = (size_t)b.total_cent * acc_record_qwords(b.CP);
    b.d_acc = A.get<unsigned long long>(acc_q);
    if (!b.d_cent || !b.d_head || !b.d_acc) return OBIA_E_NOMEM;
    OBIA_HIP_TRY(hipMemsetAsync(b.d_acc, 0, sizeof(unsigned long long) * acc_q, ctx->stream));
    return OBIA_OK;
}

}  // namespace obia
