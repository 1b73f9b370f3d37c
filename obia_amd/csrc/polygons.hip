// polygons.hip -- label raster -> polygon rings on gfx950 (the step right after the hot path).
//
// Replaces the vectorisation loop of obia create_segments (segment_boundaries.py:59-77): for every segment id the
// reference builds a full-raster mask, runs rasterio.features.shapes (GDAL polygonize, 4-connected) and keeps the
// polygon -- O(N * H * W).  Here ONE pass finds every ring of every label:
//   * a ring is the closed chain of pixel edges that separate label L (kept on the RIGHT of the direction of travel,
//     image coordinates, y down) from anything else; exterior rings run clockwise on screen, holes counter-clockwise;
//   * at a corner the walk turns right if the pixel ahead-right is not L, else goes straight if the pixel ahead-left
//     is not L, else turns left -- at a corner where L touches itself only diagonally this keeps the two pixels
//     apart, i.e. regions are 4-connected like GDAL's default and like the label maps of cc.hip;
//   * every ring has exactly one smallest corner in raster order, and the ring can only look two ways there:
//     "pixel is L, the pixels above and to the left are not" (exterior ring, leaves the corner heading east) or
//     "pixel is not L, the pixels above and to the left are both L" (hole of L, leaves heading south).
//     Those corners are the candidates; one lane walks each candidate's ring and gives up as soon as it meets a
//     smaller corner, so exactly one lane per ring completes.  Segments are a few hundred pixels, rings a few dozen
//     edges: the walks are short.  (A label with a very long, very ragged outline costs candidates x length.)
// Vertices are emitted only where the direction changes, in pixel-corner coordinates (x, y) with (0,0) the top-left
// corner of the raster, first vertex repeated at the end; rings come out in raster order of their smallest corner.
// The host applies the affine transform and groups rings by label (obia_amd/polygons.py).
#include "slic.hpp"

namespace obia {

// ---- ordered compaction helpers: exclusive scan of int32 values, 4096 per workgroup ---------------------------------
constexpr int PS_NT = 256, PS_PER = 16, PS_CHUNK = PS_NT * PS_PER;

__device__ __forceinline__ int block_exclusive_scan(int v, int *s_wave, int &block_total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    int before = 0, tot = 0;
    for (int w = 0; w < PS_NT / 64; ++w) { if (w < wv) before += s_wave[w]; tot += s_wave[w]; }
    block_total = tot;
    __syncthreads();
    return before + inc - v;
}

__global__ __launch_bounds__(PS_NT) void scan_blocksum_kernel(const int *__restrict__ in, long long n, int *__restrict__ sums) {
    __shared__ int s_wave[PS_NT / 64];
    const long long base = (long long)blockIdx.x * PS_CHUNK;
    int c = 0;
    for (int j = 0; j < PS_PER; ++j) {
        const long long i = base + (long long)j * PS_NT + threadIdx.x;
        if (i < n) c += in[i];
    }
    int tot;
    (void)block_exclusive_scan(c, s_wave, tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// exclusive scan of the block sums in place, one workgroup; grand total -> *total
__global__ __launch_bounds__(1024) void scan_sums_kernel(int *__restrict__ sums, int nb, long long *__restrict__ total) {
    __shared__ long long s_part[1024];
    const int tid = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = tid * per, hi = min(nb, lo + per);
    long long t = 0;
    for (int i = lo; i < hi; ++i) t += sums[i];
    s_part[tid] = t;
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int i = 0; i < 1024; ++i) { const long long v = s_part[i]; s_part[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    long long run = s_part[tid];
    for (int i = lo; i < hi; ++i) { const int v = sums[i]; sums[i] = (int)run; run += v; }
}

__global__ __launch_bounds__(PS_NT) void scan_apply_kernel(const int *__restrict__ in, long long n, const int *__restrict__ sums,
                                                           int *__restrict__ out) {
    __shared__ int s_wave[PS_NT / 64];
    const long long base = (long long)blockIdx.x * PS_CHUNK + (long long)threadIdx.x * PS_PER;   // 16 consecutive values per lane
    int v[PS_PER], c = 0;
#pragma unroll
    for (int j = 0; j < PS_PER; ++j) { v[j] = (base + j < n) ? in[base + j] : 0; c += v[j]; }
    int tot;
    int run = sums[blockIdx.x] + block_exclusive_scan(c, s_wave, tot);
#pragma unroll
    for (int j = 0; j < PS_PER; ++j) { if (base + j < n) out[base + j] = run; run += v[j]; }
}

// out[i] = sum(in[0..i-1]); *d_total = sum of all.  in == out is allowed.
static int exclusive_scan_i32(obia_ctx *ctx, const int *in, long long n, int *out, long long *d_total) {
    if (n <= 0) { OBIA_HIP_TRY(hipMemsetAsync(d_total, 0, sizeof(long long), ctx->stream)); return OBIA_OK; }
    const int nb = cdiv(n, PS_CHUNK);
    int *sums = ctx->arena.get<int>((size_t)nb);
    if (!sums) return OBIA_E_NOMEM;
    hipLaunchKernelGGL(scan_blocksum_kernel, dim3(nb), dim3(PS_NT), 0, ctx->stream, in, n, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, ctx->stream, sums, nb, d_total);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(PS_NT), 0, ctx->stream, in, n, sums, out);
    return OBIA_OK;
}

// ---- candidates -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lab_at(const int32_t *__restrict__ lab, int H, int W, int y, int x) {
    return (y >= 0 && y < H && x >= 0 && x < W) ? lab[(long long)y * W + x] : INT32_MIN;
}

// per pixel: bit 0 = exterior-ring candidate of the pixel's own label, bit 1 = hole candidate of the label above/left
__device__ __forceinline__ int candidate_bits(const int32_t *__restrict__ lab, int H, int W, int y, int x, int start_label) {
    const int L = lab[(long long)y * W + x];
    const int up = lab_at(lab, H, W, y - 1, x), left = lab_at(lab, H, W, y, x - 1);
    int bits = 0;
    if (L >= start_label && up != L && left != L) bits |= 1;
    if (up == left && up >= start_label && up != L) bits |= 2;
    return bits;
}

__global__ __launch_bounds__(256) void poly_candidate_count_kernel(const int32_t *__restrict__ lab, int H, int W, int start_label,
                                                                   int *__restrict__ cnt) {
    const long long n = (long long)H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        const int b = candidate_bits(lab, H, W, y, x, start_label);
        cnt[i] = (b & 1) + (b >> 1);
    }
}

// candidate record: pixel index * 2 + kind (0 exterior, 1 hole); exterior before hole at the same corner
__global__ __launch_bounds__(256) void poly_candidate_emit_kernel(const int32_t *__restrict__ lab, int H, int W, int start_label,
                                                                  const int *__restrict__ pos, long long *__restrict__ cand) {
    const long long n = (long long)H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        const int b = candidate_bits(lab, H, W, y, x, start_label);
        if (!b) continue;
        int o = pos[i];
        if (b & 1) cand[o++] = i * 2;
        if (b & 2) cand[o] = i * 2 + 1;
    }
}

// ---- ring walk --------------------------------------------------------------------------------------------------------
// directions 0 = east (+x), 1 = south (+y), 2 = west, 3 = north; the pixel on the right / left of the edge that leaves
// corner (cy, cx) heading d:
__device__ __forceinline__ void side_pixels(int cy, int cx, int d, int &ry, int &rx, int &ly, int &lx) {
    switch (d) {
        case 0: ry = cy; rx = cx; ly = cy - 1; lx = cx; break;
        case 1: ry = cy; rx = cx - 1; ly = cy; lx = cx; break;
        case 2: ry = cy - 1; rx = cx - 1; ly = cy; lx = cx - 1; break;
        default: ry = cy - 1; rx = cx; ly = cy - 1; lx = cx - 1; break;
    }
}

// Walks the ring of `cand`; returns the number of vertices including the closing repeat, or 0 when a smaller corner is
// met (the ring belongs to another candidate).  With xy != nullptr the vertices are written as (x, y) pairs.
__device__ int walk_ring(const int32_t *__restrict__ lab, int H, int W, long long cand, int32_t *__restrict__ xy, int *label_out) {
    const long long pix = cand >> 1;
    const int kind = (int)(cand & 1);
    const int sy = (int)(pix / W), sx = (int)(pix - (long long)sy * W);
    const int L = kind ? lab_at(lab, H, W, sy - 1, sx) : lab[pix];
    if (label_out) *label_out = L;
    const int d0 = kind ? 1 : 0;
    int cy = sy, cx = sx, d = d0, nv = 0;
    if (xy) { xy[0] = sx; xy[1] = sy; }
    nv = 1;
    for (;;) {
        cy += (d == 1) - (d == 3);
        cx += (d == 0) - (d == 2);
        if (cy < sy || (cy == sy && cx < sx)) return 0;          // a smaller corner: not this candidate's ring
        int ry, rx, ly, lx;
        side_pixels(cy, cx, d, ry, rx, ly, lx);                  // the pixels ahead-right / ahead-left of the new corner
        const bool ar = lab_at(lab, H, W, ry, rx) == L, al = lab_at(lab, H, W, ly, lx) == L;
        const int nd = !ar ? ((d + 1) & 3) : (!al ? d : ((d + 3) & 3));
        if (cy == sy && cx == sx && nd == d0) break;             // back at the start, about to repeat the first edge
        if (nd != d) {
            if (xy) { xy[2 * nv] = cx; xy[2 * nv + 1] = cy; }
            ++nv;
        }
        d = nd;
    }
    if (xy) { xy[2 * nv] = sx; xy[2 * nv + 1] = sy; }
    return nv + 1;
}

__global__ __launch_bounds__(256) void poly_walk_count_kernel(const int32_t *__restrict__ lab, int H, int W,
                                                              const long long *__restrict__ cand, long long ncand,
                                                              int *__restrict__ nverts, int *__restrict__ isring) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncand) return;
    const int nv = walk_ring(lab, H, W, cand[c], nullptr, nullptr);
    nverts[c] = nv;
    isring[c] = nv > 0;
}

__global__ __launch_bounds__(256) void poly_walk_write_kernel(const int32_t *__restrict__ lab, int H, int W,
                                                              const long long *__restrict__ cand, long long ncand,
                                                              const int *__restrict__ nverts, const int *__restrict__ ring_idx,
                                                              const int *__restrict__ vert_off, int32_t *__restrict__ ring_label,
                                                              uint8_t *__restrict__ ring_hole, int64_t *__restrict__ ring_offset,
                                                              int32_t *__restrict__ xy) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncand || nverts[c] == 0) return;
    const int r = ring_idx[c];
    const long long off = vert_off[c];
    int L;
    (void)walk_ring(lab, H, W, cand[c], xy + 2 * off, &L);
    ring_label[r] = L;
    ring_hole[r] = (uint8_t)(cand[c] & 1);
    ring_offset[r] = off;
}

__global__ void poly_close_offsets_kernel(int64_t *ring_offset, const long long *n_rings, const long long *n_verts) {
    ring_offset[*n_rings] = *n_verts;
}

// Phase A (both entry points): candidates, walk counts, ring and vertex prefix sums.
struct PolyPlan {
    long long *cand = nullptr;
    long long ncand = 0, n_rings = 0, n_verts = 0;
    int *nverts = nullptr, *ring_idx = nullptr, *vert_off = nullptr;
    long long *d_totals = nullptr;   // [0] candidates, [1] rings, [2] vertices
};

static int polygon_plan(obia_ctx *ctx, const int32_t *lab, int H, int W, int start_label, PolyPlan &pl) {
    if (!lab || H <= 0 || W <= 0) { set_error("bad label raster"); return OBIA_E_INVALID; }
    const long long n = (long long)H * W;
    if (n > 0x3fffffffLL) { set_error("label raster above 2^30 pixels: polygonise it in slabs"); return OBIA_E_INVALID; }
    Arena &A = ctx->arena;
    int *cnt = A.get<int>((size_t)n);
    pl.d_totals = A.get<long long>(3);
    if (!cnt || !pl.d_totals) return OBIA_E_NOMEM;
    OBIA_HIP_TRY(hipMemsetAsync(pl.d_totals, 0, 3 * sizeof(long long), ctx->stream));
    int g = cdiv(n, 256 * 4);
    if (g > 262144) g = 262144;
    hipLaunchKernelGGL(poly_candidate_count_kernel, dim3(g), dim3(256), 0, ctx->stream, lab, H, W, start_label, cnt);
    OBIA_TRY(exclusive_scan_i32(ctx, cnt, n, cnt, pl.d_totals));
    long long h = 0;
    OBIA_TRY(read_back(ctx, &h, pl.d_totals, sizeof(long long)));
    pl.ncand = h;
    if (h > 0x7fffffffLL) { set_error("too many ring candidates (%lld)", h); return OBIA_E_UNSUPPORTED; }
    if (h == 0) return OBIA_OK;
    pl.cand = A.get<long long>((size_t)h);
    pl.nverts = A.get<int>((size_t)h);
    pl.ring_idx = A.get<int>((size_t)h);
    pl.vert_off = A.get<int>((size_t)h);
    if (!pl.cand || !pl.nverts || !pl.ring_idx || !pl.vert_off) return OBIA_E_NOMEM;
    hipLaunchKernelGGL(poly_candidate_emit_kernel, dim3(g), dim3(256), 0, ctx->stream, lab, H, W, start_label, cnt, pl.cand);
    hipLaunchKernelGGL(poly_walk_count_kernel, dim3(cdiv(h, 256)), dim3(256), 0, ctx->stream, lab, H, W, pl.cand, h, pl.nverts,
                       pl.ring_idx);
    OBIA_TRY(exclusive_scan_i32(ctx, pl.ring_idx, h, pl.ring_idx, pl.d_totals + 1));
    OBIA_TRY(exclusive_scan_i32(ctx, pl.nverts, h, pl.vert_off, pl.d_totals + 2));
    long long t[3];
    OBIA_TRY(read_back(ctx, t, pl.d_totals, sizeof(t)));
    pl.n_rings = t[1];
    pl.n_verts = t[2];
    if (pl.n_verts > 0x7fffffffLL) { set_error("too many ring vertices (%lld)", pl.n_verts); return OBIA_E_UNSUPPORTED; }
    OBIA_HIP_TRY(hipGetLastError());
    return OBIA_OK;
}

int polygon_count_dev(obia_ctx *ctx, const int32_t *lab, int H, int W, int start_label, int64_t *n_rings, int64_t *n_verts) {
    PolyPlan pl;
    OBIA_TRY(polygon_plan(ctx, lab, H, W, start_label, pl));
    *n_rings = pl.n_rings;
    *n_verts = pl.n_verts;
    return OBIA_OK;
}

int polygon_rings_dev(obia_ctx *ctx, const int32_t *lab, int H, int W, int start_label, int64_t cap_rings, int64_t cap_verts,
                      int32_t *ring_label, uint8_t *ring_hole, int64_t *ring_offset, int32_t *xy, int64_t *n_rings,
                      int64_t *n_verts) {
    PolyPlan pl;
    OBIA_TRY(polygon_plan(ctx, lab, H, W, start_label, pl));
    *n_rings = pl.n_rings;
    *n_verts = pl.n_verts;
    if (pl.n_rings > cap_rings || pl.n_verts > cap_verts) {
        set_error("polygon output needs %lld rings / %lld vertices, buffers hold %lld / %lld", pl.n_rings, pl.n_verts,
                  (long long)cap_rings, (long long)cap_verts);
        return OBIA_E_NOMEM;
    }
    if (pl.ncand > 0)
        hipLaunchKernelGGL(poly_walk_write_kernel, dim3(cdiv(pl.ncand, 256)), dim3(256), 0, ctx->stream, lab, H, W, pl.cand,
                           pl.ncand, pl.nverts, pl.ring_idx, pl.vert_off, ring_label, ring_hole, ring_offset, xy);
    hipLaunchKernelGGL(poly_close_offsets_kernel, dim3(1), dim3(1), 0, ctx->stream, ring_offset, pl.d_totals + 1, pl.d_totals + 2);
    OBIA_HIP_TRY(hipGetLastError());
    return OBIA_OK;
}

}  // namespace obia

using namespace obia;

extern "C" {

int obia_polygon_count_i32_dev(obia_ctx *ctx, const int32_t *labels, int H, int W, int start_label, int64_t *n_rings_out,
                               int64_t *n_vertices_out) {
    if (!ctx) { set_error("null context"); return OBIA_E_INVALID; }
    if (!n_rings_out || !n_vertices_out) { set_error("null pointer argument"); return OBIA_E_INVALID; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return OBIA_E_HIP; }
    ctx->arena.reset();
    OBIA_TRY(polygon_count_dev(ctx, labels, H, W, start_label, n_rings_out, n_vertices_out));
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return OBIA_OK;
}

int obia_polygon_rings_i32_dev(obia_ctx *ctx, const int32_t *labels, int H, int W, int start_label, int64_t cap_rings,
                               int64_t cap_vertices, int32_t *ring_label, uint8_t *ring_is_hole, int64_t *ring_offset,
                               int32_t *xy, int64_t *n_rings_out, int64_t *n_vertices_out) {
    if (!ctx) { set_error("null context"); return OBIA_E_INVALID; }
    if (!ring_label || !ring_is_hole || !ring_offset || !xy || !n_rings_out || !n_vertices_out) { set_error("null pointer argument"); return OBIA_E_INVALID; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return OBIA_E_HIP; }
    ctx->arena.reset();
    OBIA_TRY(polygon_rings_dev(ctx, labels, H, W, start_label, cap_rings, cap_vertices, ring_label, ring_is_hole, ring_offset, xy,
                               n_rings_out, n_vertices_out));
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return OBIA_OK;
}

}  // extern "C"
