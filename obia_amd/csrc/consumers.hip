// consumers.hip -- the two consumers of the label raster that SURVEY.md 8f4 lists, on gfx950.
//   label_edges  : obia/utils/cost.py:44-48 `slic_edge` -- a pixel is an edge when its label differs from the pixel below
//                  it or from the pixel to its right (the last row / column only look the other way)
//   sample_labels: the point-in-segment join of `label_segments` (obia/utils/utils.py:12-34, geopandas sjoin with
//                  predicate "intersects") on a label raster: map coordinates -> inverse affine -> pixel -> label
#include "slic.hpp"

namespace obia {

__global__ __launch_bounds__(256) void label_edges_kernel(const int32_t *__restrict__ lab, int H, int W, uint8_t *__restrict__ edge,
                                                          unsigned long long *__restrict__ n_edge) {
    unsigned cnt = 0;
    for (int y = blockIdx.x; y < H; y += gridDim.x) {
        const int32_t *row = lab + (long long)y * W;
        const int32_t *below = (y + 1 < H) ? row + W : nullptr;
        for (int x0 = threadIdx.x; x0 < W; x0 += 4 * 256) {   // four pixels in flight per lane
            int l[4], r[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int x = x0 + 256 * u;
                l[u] = r[u] = b[u] = 0;
                if (x < W) {
                    l[u] = row[x];
                    r[u] = (x + 1 < W) ? row[x + 1] : l[u];
                    b[u] = below ? below[x] : l[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int x = x0 + 256 * u;
                if (x >= W) continue;
                const uint8_t e = (l[u] != r[u]) || (l[u] != b[u]);
                edge[(long long)y * W + x] = e;
                cnt += e;
            }
        }
    }
    // one atomic per workgroup (same-word global atomics serialise device-wide)
    __shared__ unsigned s_c[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        if (t) atomicAdd(n_edge, (unsigned long long)t);
    }
}

// inv = [a, b, d, e, xoff, yoff] of the INVERSE transform: col = a*X + b*Y + xoff, row = d*X + e*Y + yoff (pixel-corner
// coordinates); a point belongs to the pixel floor(col), floor(row); outside the raster -> `outside`.
__global__ void sample_labels_kernel(const int32_t *__restrict__ lab, int H, int W, double a, double b, double d, double e,
                                     double xoff, double yoff, const double *__restrict__ xy, long long n, int outside,
                                     int32_t *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double X = xy[2 * i], Y = xy[2 * i + 1];
    const double col = floor(a * X + b * Y + xoff), row = floor(d * X + e * Y + yoff);
    int v = outside;
    if (col >= 0.0 && col < (double)W && row >= 0.0 && row < (double)H) v = lab[(long long)row * W + (long long)col];
    out[i] = v;
}

}  // namespace obia

using namespace obia;

extern "C" {

int obia_label_edges_u8_dev(obia_ctx *ctx, const int32_t *labels, int H, int W, uint8_t *edge_out, int64_t *n_edge_out) {
    if (!ctx) { set_error("null context"); return OBIA_E_INVALID; }
    if (!labels || !edge_out || H <= 0 || W <= 0) { set_error("bad arguments"); return OBIA_E_INVALID; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return OBIA_E_HIP; }
    ctx->arena.reset();
    unsigned long long *d_n = ctx->arena.get<unsigned long long>(1);
    if (!d_n) return OBIA_E_NOMEM;
    OBIA_HIP_TRY(hipMemsetAsync(d_n, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(label_edges_kernel, dim3(H < 8192 ? H : 8192), dim3(256), 0, ctx->stream, labels, H, W, edge_out, d_n);
    OBIA_HIP_TRY(hipGetLastError());
    unsigned long long h = 0;
    OBIA_TRY(read_back(ctx, &h, d_n, sizeof(h)));
    if (n_edge_out) *n_edge_out = (int64_t)h;
    return OBIA_OK;
}

int obia_sample_labels_i32_dev(obia_ctx *ctx, const int32_t *labels, int H, int W, const double *inverse_affine6,
                               const double *points_xy, int64_t n_points, int outside_value, int32_t *labels_out) {
    if (!ctx) { set_error("null context"); return OBIA_E_INVALID; }
    if (!labels || !inverse_affine6 || H <= 0 || W <= 0 || n_points < 0 || (n_points > 0 && (!points_xy || !labels_out))) {
        set_error("bad arguments");
        return OBIA_E_INVALID;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return OBIA_E_HIP; }
    if (n_points == 0) return OBIA_OK;
    const double *t = inverse_affine6;
    hipLaunchKernelGGL(sample_labels_kernel, dim3(cdiv(n_points, 256)), dim3(256), 0, ctx->stream, labels, H, W, t[0], t[1], t[2], t[3],
                       t[4], t[5], points_xy, (long long)n_points, outside_value, labels_out);
    OBIA_HIP_TRY(hipGetLastError());
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return OBIA_OK;
}

}  // extern "C"
