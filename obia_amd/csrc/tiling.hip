// tiling.hip -- B3: the tiled driver on label rasters.
//
// Restates the two tile loops of obia.utils.tiling.create_tiled_segments (tiling.py:103-291), where the
// reference keeps two growing GeoDataFrames and tests every accumulated polygon against every white tile
// with shapely predicates (tiling.py:205-231).  Here the state is one global label raster G (0 = no
// segment) plus a per-segment pixel count, and the predicates are pixel counts:
//   within(tile_polygon)   <=>  every pixel of the segment lies inside the grown window minus the two
//                               bottom corner squares           (count inside == segment size)
//   overlaps(tile_polygon) <=>  some but not all of its pixels do
// (segments are 4-connected pixel sets and rasterize() uses the pixel-centre rule; the window is pixel-aligned, the two corner
// squares are when buffer/2 is a whole number of pixels -- otherwise the cut passes THROUGH pixels, and "inside" means three
// different things for the three uses of the squares: see TileWin).
//   pass 1  "black" tiles ((i/T + j/T) even), exact windows          tiling.py:103-153
//   pass 2  "white" tiles, windows grown by `buffer` and clamped      tiling.py:156-172
//           segments within the polygon are dropped and re-segmented  tiling.py:220-231
//           segments overlapping it are kept and masked out, together with the corner squares
//                                                                    tiling.py:213-260
//           ... but only when at least one existing segment is within / overlaps the polygon (tiling.py:212); otherwise
//           the mask is left as read and the corner squares ARE segmented (tiling.py:261-262): tile_any[]
//           n_segments = round(mask.sum() * pixel_area / (pi * crown_radius^2))   tiling.py:126-135
//   ids 1..N in the order black (survivors), then white              tiling.py:289-290
// All tiles of a pass (or of one white tile-row) are ONE batch for the SLIC engine and for the
// connectivity kernels: no per-tile launches, two host read-backs per batch.
#include "slic.hpp"

#include <cmath>
#include <cstdlib>

namespace obia {

// window, dense offset, corner squares in pixels.  The squares have side buffer/2 in MAP units (tiling.py:189), i.e. a = buffer/2/pixel
// pixels, which need not be whole; three pixel counts follow from the reference's three uses of the square (per axis):
//   cl*     pixels whose CENTRE lies inside: what rasterize() burns into the mask (all_touched=False)        tiling.py:245-255
//   cl*_in  pixels that lie WHOLLY inside: a segment made of such pixels only has no area in tile_polygon,
//           it is neither `within` nor `overlaps` (not selected)                                             tiling.py:205-210
//   cl*_any pixels that meet the square's interior at all: a segment with such a pixel is not `within`       tiling.py:220-231
// (equal when a is a whole number -- even buffer, pixel size 1 or 0.5 -- which is every case rounds 1-3 tested)
struct TileWin { int y0, x0, h, w; long long pix_off; int cly, clx; int cly_in, clx_in, cly_any, clx_any; int m4_off; };   // m4_off: SlicProblem::m4_off

// wave-aggregated histogram add: lanes of a wave that hold the same key add once
__device__ __forceinline__ void wave_hist_add(unsigned *hist, int key, bool active) {
    bool todo = active;
    while (true) {
        const unsigned long long act = __ballot(todo);
        if (!act) break;
        const int leader = __ffsll((long long)act) - 1;
        const int kk = __shfl(key, leader);
        const unsigned long long same = __ballot(todo && key == kk);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[kk], (unsigned)__popcll(same));
        if (key == kk) todo = false;
    }
}

__device__ __forceinline__ bool in_corner(const TileWin &t, int y, int x) {
    return (y >= t.h - t.cly) && (x < t.clx || x >= t.w - t.clx);
}
__device__ __forceinline__ bool wholly_in_corner(const TileWin &t, int y, int x) {
    return (y >= t.h - t.cly_in) && (x < t.clx_in || x >= t.w - t.clx_in);
}
__device__ __forceinline__ bool meets_corner(const TileWin &t, int y, int x) {
    return (y >= t.h - t.cly_any) && (x < t.clx_any || x >= t.w - t.clx_any);
}

// white tiles, step 1: inside[g] = pixels of every existing segment that lie in the window and do not meet a corner square
// (inside[g] == seg_size[g]  <=>  the segment is `within` tile_polygon)
// tile_any[tile] = 1 when some existing segment has area inside the polygon -- a pixel in the window that is not wholly inside a
// corner square: `not intersecting_black_segments.empty or not intersecting_white_segments.empty` (tiling.py:205-212)
__global__ __launch_bounds__(256) void tile_count_inside_kernel(const TileWin *__restrict__ wins, const int32_t *__restrict__ G,
                                                                int Wr, unsigned *__restrict__ inside, int *__restrict__ tile_any) {
    const TileWin t = wins[blockIdx.y];
    bool seen = false;
    const int wround = ((t.w + 255) / 256) * 256;   // whole waves stay in the loop for the wave-level histogram
    for (int y = blockIdx.x; y < t.h; y += gridDim.x) {
        const bool low = y >= t.h - t.cly_any;   // a row the corner squares reach (workgroup-uniform: the others test no pixel)
        for (int x0 = 0; x0 < wround; x0 += 4 * 256) {   // four loads in flight
            int g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int x = x0 + 256 * u + threadIdx.x;
                g[u] = 0;
                if (x < t.w && !(low && wholly_in_corner(t, y, x))) {
                    g[u] = G[(long long)(t.y0 + y) * Wr + t.x0 + x];
                    seen |= g[u] > 0;
                    if (low && meets_corner(t, y, x)) g[u] = 0;   // (a pixel the cut passes through: selected, but not counted as inside)
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (x0 + 256 * u < wround) wave_hist_add(inside, g[u], g[u] > 0);   // wave-uniform condition
        }
    }
    if (__ballot(seen) && (threadIdx.x & 63) == 0) tile_any[blockIdx.y] = 1;   // (every writer stores the same value)
}

// step 2: dense tile mask.  black: the input mask.  white: input mask minus kept (overlapping) segments
// minus the corner squares; segments within the polygon are erased from G (they will be re-segmented).
template <bool VEC4>
__global__ __launch_bounds__(256) void tile_mask_kernel(const TileWin *__restrict__ wins, const uint8_t *__restrict__ inmask,
                                                        int32_t *__restrict__ G, int Wr, int white,
                                                        const unsigned *__restrict__ inside, const unsigned *__restrict__ seg_size,
                                                        uint8_t *__restrict__ alive, uint8_t *__restrict__ dmask,
                                                        const int *__restrict__ tile_any, unsigned *__restrict__ dmask4) {
    const TileWin t = wins[blockIdx.y];
    // a white tile whose polygon no existing segment intersects keeps the mask it read: neither kept segments (there are none
    // inside the polygon) nor the corner squares are masked out (tiling.py:212, 261-262)
    if (white && tile_any[blockIdx.y] == 0) white = 0;
    // one pixel of the tile: its mask byte given the input mask byte and the global label under it
    auto decide = [&](int y, int x, long long gp, uint8_t m, int g) -> uint8_t {
        if (!white) return m;
        if (in_corner(t, y, x)) return 0;
        if (g > 0) {
            if (inside[g] == seg_size[g]) { G[gp] = 0; alive[g] = 0; }   // within: dropped
            else m = 0;                                                    // overlaps: kept, masked out
        }
        return m;
    };
    if (VEC4) {
        // four consecutive pixels per lane: one dword of mask bytes, one int4 of labels, one dword stored (the host checks that
        // every window's x0, w and pix_off and the row pitch are multiples of four and the base pointers aligned)
        // A thread owns a 4 x 4 block: the four rows of one quad row (the eight loads in flight together), so that it can also
        // write the block TRANSPOSED -- the packed mask of the sweeps (SlicBatch::d_mask4: dword (q, x) = the bytes of the rows
        // 4q .. 4q+3 at column x), which used to be a pass of its own over the batch's mask (round 4).
        const int w4 = t.w >> 2, nq = (t.h + 3) >> 2;
        for (int q = blockIdx.x; q < nq; q += gridDim.x)
            for (int x4 = threadIdx.x; x4 < w4; x4 += 256) {
                const int x = 4 * x4;
                unsigned mw[4];
                int4 g[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = 4 * q + j;
                    const long long gp = (long long)(t.y0 + (y < t.h ? y : t.h - 1)) * Wr + t.x0 + x;
                    mw[j] = inmask ? *reinterpret_cast<const unsigned *>(inmask + gp) : 0x01010101u;
                    g[j] = make_int4(0, 0, 0, 0);
                    if (white) g[j] = *reinterpret_cast<const int4 *>(G + gp);
                }
                unsigned out[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = 4 * q + j;
                    if (y >= t.h) continue;   // (rows past the window: zero bytes in the packed mask, nothing in the row-major one)
                    const long long gp = (long long)(t.y0 + y) * Wr + t.x0 + x;
                    const int gv[4] = {g[j].x, g[j].y, g[j].z, g[j].w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint8_t m = ((mw[j] >> (8 * u)) & 0xffu) != 0;
                        out[j] |= (unsigned)decide(y, x + u, gp + u, m, gv[u]) << (8 * u);
                    }
                    *reinterpret_cast<unsigned *>(dmask + t.pix_off + (long long)y * t.w + x) = out[j];
                }
                uint4 o;   // column c of the block: byte c of every row
                unsigned *ov = &o.x;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    ov[c] = ((out[0] >> (8 * c)) & 0xffu) | (((out[1] >> (8 * c)) & 0xffu) << 8) | (((out[2] >> (8 * c)) & 0xffu) << 16) |
                            (((out[3] >> (8 * c)) & 0xffu) << 24);
                *reinterpret_cast<uint4 *>(dmask4 + (long long)t.m4_off + (long long)q * t.w + x) = o;
            }
        return;
    }
    for (int y = blockIdx.x; y < t.h; y += gridDim.x)
    for (int x0 = threadIdx.x; x0 < t.w; x0 += 4 * 256) {   // four pixels in flight per lane
        uint8_t mv[4];
        int gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = x0 + 256 * u;
            mv[u] = 0; gv[u] = 0;
            if (x < t.w) {
                const long long gp = (long long)(t.y0 + y) * Wr + t.x0 + x;
                mv[u] = inmask ? (inmask[gp] != 0) : 1;
                if (white && !in_corner(t, y, x)) gv[u] = G[gp];   // (outside the burned square => not wholly inside it: the segment is selected)
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = x0 + 256 * u;
            if (x >= t.w) continue;
            const long long gp = (long long)(t.y0 + y) * Wr + t.x0 + x;
            dmask[t.pix_off + (long long)y * t.w + x] = decide(y, x, gp, mv[u], gv[u]);
        }
    }
}

// step 3: write the new segments of the batch into G with their provisional global ids + size histogram.
// Global atomics run at ~2e10 per second device-wide, so the histogram is built per 64x64 block of the window first:
// a lane walks one column (runs of equal label down the column are counted in a register), run totals go to a 128-slot
// LDS table keyed by label, and the table is flushed with one global atomic per (block, label).
constexpr int TS_SLOTS = 128;
__global__ __launch_bounds__(64) void tile_scatter_kernel(const TileWin *__restrict__ wins, const CcResolve R,
                                                          int32_t *__restrict__ G, int Wr, int id_base,
                                                          unsigned *__restrict__ seg_size) {
    __shared__ int s_key[TS_SLOTS];
    __shared__ unsigned s_cnt[TS_SLOTS];
    const TileWin t = wins[blockIdx.y];
    const int bw = (t.w + 63) / 64;
    const int by = blockIdx.x / bw, bx = blockIdx.x % bw;
    if (by * 64 >= t.h) return;   // whole workgroup
    const int lane = threadIdx.x;
    for (int i = lane; i < TS_SLOTS; i += 64) { s_key[i] = 0; s_cnt[i] = 0; }
    __syncthreads();
    const int x = bx * 64 + lane;
    const bool col_ok = x < t.w;
    const int y_lo = by * 64, y_hi = min(y_lo + 64, t.h);
    int rid = 0;
    unsigned rn = 0;
    auto close_run = [&]() {
        if (rid <= 0) return;
        const unsigned h = ((unsigned)rid * 2654435761u) >> 25;
        int slot = -1;
#pragma unroll 1
        for (int probe = 0; probe < TS_SLOTS; ++probe) {
            const int sidx = (h + probe) & (TS_SLOTS - 1);
            const int old = atomicCAS(&s_key[sidx], 0, rid);
            if (old == 0 || old == rid) { slot = sidx; break; }
        }
        if (slot >= 0) atomicAdd(&s_cnt[slot], rn);
        else atomicAdd(&seg_size[rid], rn);          // table full: straight to global memory
    };
#pragma unroll 1
    for (int y0 = y_lo; y0 < y_hi; y0 += 8) {       // eight rows in flight
        int l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int y = y0 + j;
            // (the label of the pixel's component, resolved here: the connectivity pass hands over its tables instead of a label map)
            l[j] = (col_ok && y < y_hi) ? cc_resolve_label(R, t.pix_off + (long long)y * t.w + x) : 0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int y = y0 + j;
            const int id = l[j] > 0 ? id_base + l[j] : 0;
            if (id > 0) G[(long long)(t.y0 + y) * Wr + t.x0 + x] = id;
            if (id != rid) { close_run(); rid = id; rn = 0; }
            rn += 1;
        }
    }
    close_run();
    __syncthreads();
    for (int i = lane; i < TS_SLOTS; i += 64)
        if (s_key[i] > 0 && s_cnt[i]) atomicAdd(&seg_size[s_key[i]], s_cnt[i]);
}

// final ids 1..N: exclusive scan over the alive flags of the provisional ids (the table has ~1e6 entries).
// Two launches: per-chunk counts, then every workgroup adds up the counts of the chunks before its own and scans its
// chunk of 1024 x 16 flags (coalesced 16-byte reads).
constexpr int IDS_CHUNK = 1024 * 16;

__global__ __launch_bounds__(1024) void ids_count_kernel(const uint8_t *__restrict__ alive, int n_ids, int *__restrict__ partial) {
    __shared__ int s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = blockIdx.x * IDS_CHUNK + tid * 16;
    int c = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) c += (i0 + q < n_ids) ? (alive[i0 + q] != 0) : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0) s_wave[wv] = c;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w2 = 0; w2 < 16; ++w2) t += s_wave[w2];
        partial[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void ids_scan_kernel(const uint8_t *__restrict__ alive, int n_ids, const int *__restrict__ partial,
                                                        int *__restrict__ newid, long long *__restrict__ total) {
    __shared__ int s_wave[16];
    __shared__ int s_base[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // alive ids in the chunks before this one
    int pre = 0;
    for (int j = tid; j < (int)blockIdx.x; j += 1024) pre += partial[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off);
    if (lane == 0) s_base[wv] = pre;
    const int i0 = blockIdx.x * IDS_CHUNK + tid * 16;
    unsigned char f[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) f[q] = (i0 + q < n_ids) ? alive[i0 + q] : 0;
    int c = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) c += f[q] != 0;
    // inclusive scan of c over the wave (shuffle up), then over the 16 waves
    int inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(inc, off);
        if (lane >= off) inc += v;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    int before = 0;
    for (int w2 = 0; w2 < 16; ++w2) { before += s_base[w2]; if (w2 < wv) before += s_wave[w2]; }
    int run = before + inc - c;
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if (i0 + q < n_ids) { if (f[q]) { run += 1; newid[i0 + q] = run; } else newid[i0 + q] = 0; }
    if (blockIdx.x == gridDim.x - 1 && tid == 1023) *total = before + inc;
}

// G[i] = newid[G[i]]: four pixels per lane and step (one 16-byte access, four gathers in flight)
__global__ __launch_bounds__(256) void ids_apply_kernel(int32_t *__restrict__ G, long long n, const int *__restrict__ newid) {
    const long long n4 = ((reinterpret_cast<uintptr_t>(G) & 15) == 0) ? n / 4 : 0;
    int4 *G4 = reinterpret_cast<int4 *>(G);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        int4 g = G4[i];
        g.x = g.x > 0 ? newid[g.x] : 0;
        g.y = g.y > 0 ? newid[g.y] : 0;
        g.z = g.z > 0 ? newid[g.z] : 0;
        g.w = g.w > 0 ? newid[g.w] : 0;
        G4[i] = g;
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int g = G[i];
        G[i] = g > 0 ? newid[g] : 0;
    }
}

struct TileState {
    int H, W, C;            // local raster: rows [row0, row0 + H) of a (Hg, W) raster
    int Hg, row0;
    const float *img;
    const uint8_t *inmask;
    int32_t *G;
    unsigned *seg_size, *inside;
    uint8_t *alive;
    int *newid;
    long long *d_total;
    int id_cap, next_id;   // provisional ids 1..next_id-1
    int clx, cly, clx_in, cly_in, clx_any, cly_any;   // corner squares in pixels (see TileWin)
    obia_tiling_params tp;
    obia_slic_params sp;
    // features of ALL white tiles, prepared in one batch (they depend on the raster only): windows in processing order,
    // one dense buffer, keys read back once before the white pass
    struct PreFeat {
        bool ready = false;
        std::vector<TileWin> wins;
        std::vector<SrcWindow> windows;
        std::vector<unsigned> host;      // keys | nonfinite | max|feature| of all windows
        float *d_feat = nullptr;
        float *d_fbox = nullptr;         // footprint colour boxes of the same windows (low compactness only)
        SrcWindow *d_windows = nullptr;
        unsigned *d_keys = nullptr;
        int maxh = 1;
        bool launched = false;
        bool on_side = false;            // the pass was queued on a side stream (beside the black sweeps): join before it is read
        size_t cursor = 0;               // next window to be consumed by a white batch
    } pf;
};

// The feature pass of the white tiles (bound by HBM, few vector instructions) runs on a side stream beside the black batch's
// sweeps, whose first nine launches -- the spatial pre-pass -- are bound by vector issue and move no data: 46.62 / 46.48 ->
// 46.10 / 46.00 ms per step (two pairs on one box; the pre-pass gives back 0.4 ms of the 2.7 ms hidden).  Round 2 had measured the
// same idea as a loss on the round-1 kernels.  OBIA_WHITE_FEATURES_BESIDE=0 puts the pass back in line (A/B, debugging).
static bool white_features_beside() {
    const char *e = std::getenv("OBIA_WHITE_FEATURES_BESIDE");
    return e ? atoi(e) != 0 : true;
}

static int grid_rows(const std::vector<TileWin> &wins) {   // row-walking kernels: one block per row (capped)
    int g = 1;
    for (auto &t : wins) if (t.h > g) g = t.h;
    return g > 8192 ? 8192 : g;
}

// One batch of tiles: mask -> features -> plan -> sweeps -> connectivity -> scatter.
static int prefetch_white_launch(obia_ctx *ctx, TileState &S, bool beside = false);

static int run_tile_batch(obia_ctx *ctx, TileState &S, std::vector<TileWin> &wins, bool white) {
    const int np = (int)wins.size();
    if (np == 0) return OBIA_OK;
    Arena &A = ctx->arena;
    const Arena::Mark mk = A.mark();
    SlicBatch b;
    b.nprob = np;
    b.C = S.C;
    b.CP = (S.C + 3) & ~3;
    b.masked = true;       // the tiler always hands a mask to slic (tiling.py:137-143): maskSLIC structure
    b.start_label = 1;
    b.max_iter = S.sp.max_num_iter;
    b.exit_on_fixed_point = S.sp.exit_on_fixed_point != 0;
    b.slic_zero = S.sp.slic_zero != 0;
    b.prescale = slic_prescale((float)(1.0 / S.sp.compactness), 1, (S.C == 3 && S.sp.convert2lab != 0) ? 1 : 0, b.slic_zero);   // (the prefetched planes carry the same factor)
    for (int i = 0; i < 3; ++i) { b.sigma[i] = S.sp.sigma_zyx[i]; b.spacing[i] = S.sp.spacing_zyx[i]; }
    const bool direct = (float)b.spacing[1] != 1.0f || (float)b.spacing[2] != 1.0f;   // anisotropic spacing: the direct sweep path
    if (direct) b.exit_on_fixed_point = false;
    long long off = 0, foff = 0, boff = 0, maxpix = 1, m4 = 0;
    b.probs.resize(np);
    b.windows.resize(np);
    for (int p = 0; p < np; ++p) {
        wins[p].pix_off = off;
        wins[p].m4_off = (int)m4;                      // the rule of slic_plan_and_seed (same order of problems): SlicProblem::m4_off
        m4 += (long long)((wins[p].h + 3) / 4) * wins[p].w;
        SlicProblem P{};
        P.H = wins[p].h; P.W = wins[p].w; P.pix_off = off; P.feat_off = foff; P.XB = feat_xb(wins[p].w); P.fb_off = boff;
        b.probs[p] = P;
        b.windows[p] = SrcWindow{wins[p].y0, wins[p].x0, wins[p].h, wins[p].w, off, foff, boff};
        const long long n = (long long)wins[p].h * wins[p].w;
        if (n > maxpix) maxpix = n;
        off += n;
        foff += feat_block_f4(wins[p].h, wins[p].w, b.CP);
        boff += feat_boxes(wins[p].h, wins[p].w);
    }
    b.total_feat_f4 = foff;
    b.col_lb = slic_use_colour_bound((float)(1.0 / S.sp.compactness), S.C == 3 && S.sp.convert2lab != 0) && !b.slic_zero && !b.exit_on_fixed_point && !direct;
    if (off > 0x7fffffffLL) { set_error("tile batch of %lld pixels too large", off); return OBIA_E_INVALID; }
    b.total_pix = off;
    TileWin *d_wins = A.get<TileWin>(np);
    b.d_windows = A.get<SrcWindow>(np);
    b.d_mask = A.get<uint8_t>((size_t)off);
    if (m4 > 0x7fffffffLL) { set_error("tile batch of %lld pixels too large", off); return OBIA_E_INVALID; }
    unsigned *d_mask4 = A.get<unsigned>((size_t)(m4 > 0 ? m4 : 1));   // the packed mask of the sweeps, written by tile_mask_kernel<true>
    if (!d_mask4) return OBIA_E_NOMEM;
    const bool pre = white && S.pf.ready;
    if (pre) {
        // this batch is the next np windows of the prefetched set (same order, same sizes: checked)
        if (S.pf.cursor + np > S.pf.wins.size()) { set_error("white batch beyond the prefetched windows"); return OBIA_E_INVALID; }
        for (int p = 0; p < np; ++p) {
            const TileWin &a = S.pf.wins[S.pf.cursor + p];
            if (a.y0 != wins[p].y0 || a.x0 != wins[p].x0 || a.h != wins[p].h || a.w != wins[p].w) { set_error("white batch does not match the prefetched windows"); return OBIA_E_INVALID; }
        }
        // (the prefetched planes of these windows lie back to back in the same order: the batch's feat_off values, which
        // start at 0, are offsets from the first window's block)
        b.d_feat = S.pf.d_feat + 4 * (size_t)S.pf.windows[S.pf.cursor].feat_off;
        if (b.col_lb) b.d_fbox = S.pf.d_fbox + (size_t)S.pf.windows[S.pf.cursor].fb_off * 2 * b.CP;
    } else {
        b.d_feat = A.get<float>(4 * (size_t)foff);
        if (b.col_lb) {
            b.d_fbox = A.get<float>((size_t)boff * 2 * b.CP);
            if (!b.d_fbox) return OBIA_E_NOMEM;
        }
    }
    b.d_labels = A.get<int32_t>((size_t)off);
    int32_t *d_final = A.get<int32_t>((size_t)off);
    if (!d_wins || !b.d_windows || !b.d_mask || !b.d_feat || !b.d_labels || !d_final) return OBIA_E_NOMEM;
    OBIA_TRY(upload_async(ctx, d_wins, wins.data(), sizeof(TileWin) * np));   // (pinned ring: no stream sync per small table)
    OBIA_TRY(upload_async(ctx, b.d_windows, b.windows.data(), sizeof(SrcWindow) * np));
    int *d_tile_any = A.get<int>(np);
    if (!d_tile_any) return OBIA_E_NOMEM;
    if (white) {
        OBIA_HIP_TRY(hipMemsetAsync(S.inside, 0, sizeof(unsigned) * (size_t)S.next_id, ctx->stream));
        OBIA_HIP_TRY(hipMemsetAsync(d_tile_any, 0, sizeof(int) * (size_t)np, ctx->stream));
        hipLaunchKernelGGL(tile_count_inside_kernel, dim3(grid_rows(wins), np), dim3(256), 0, ctx->stream, d_wins, S.G, S.W, S.inside, d_tile_any);
    }
    {
        bool vec4 = (S.W % 4 == 0) && (reinterpret_cast<uintptr_t>(S.inmask) % 4 == 0) && (reinterpret_cast<uintptr_t>(S.G) % 16 == 0) &&
                    (reinterpret_cast<uintptr_t>(b.d_mask) % 4 == 0) && (reinterpret_cast<uintptr_t>(d_mask4) % 16 == 0);
        for (auto &t : wins) vec4 = vec4 && (t.x0 % 4 == 0) && (t.w % 4 == 0) && (t.pix_off % 4 == 0);
        if (vec4) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(tile_mask_kernel<true>), dim3((grid_rows(wins) + 3) / 4, np), dim3(256), 0, ctx->stream, d_wins, S.inmask,
                               S.G, S.W, white ? 1 : 0, S.inside, S.seg_size, S.alive, b.d_mask, d_tile_any, d_mask4);
            b.d_mask4 = d_mask4;   // (slic_run_sweeps packs the mask itself when nobody did)
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(tile_mask_kernel<false>), dim3(grid_rows(wins), np), dim3(256), 0, ctx->stream, d_wins, S.inmask,
                               S.G, S.W, white ? 1 : 0, S.inside, S.seg_size, S.alive, b.d_mask, d_tile_any, d_mask4);
        }
    }
    // per-tile normalisation of every band (create_segments normalises the tile it is given, :32-33)
    std::vector<int> skip;
    const int to_lab = (S.C == 3 && S.sp.convert2lab != 0) ? 1 : 0;
    if (pre) {
        const size_t NP = S.pf.wins.size(), nkeys = NP * (size_t)S.C * 2;
        const unsigned *h = S.pf.host.data();
        OBIA_TRY(slic_features_finish(b, h + S.pf.cursor * (size_t)S.C * 2, h + nkeys + S.pf.cursor, h + nkeys + NP + S.pf.cursor, 1, &skip));
        S.pf.cursor += np;
    } else {
        OBIA_TRY(slic_prepare_features(ctx, b, S.img, S.H, S.W, 1, to_lab, (float)(1.0 / S.sp.compactness) * b.prescale, &skip));
    }
    std::vector<int> nvalid;
    debug_sync(ctx, "tiler: mask + features");
    OBIA_TRY(slic_count_valid(ctx, b, nvalid));
    debug_sync(ctx, "tiler: count_valid");
    std::vector<int> nseg(np);
    const double pixel_area = S.tp.pixel_width * S.tp.pixel_height;
    const double crown_area = M_PI * S.tp.crown_radius * S.tp.crown_radius;
    for (int p = 0; p < np; ++p) {
        double n;
        if (S.sp.n_segments > 0)   // extension: explicit n_segments per full tile, scaled by the valid area
            n = std::nearbyint((double)S.sp.n_segments * (double)nvalid[p] / ((double)S.tp.tile_size * S.tp.tile_size));
        else
            n = std::nearbyint((double)nvalid[p] * pixel_area / crown_area);   // Python round(): half to even
        nseg[p] = (skip[p] || n < 1.0) ? 0 : (n > 2.0e9 ? 2000000000 : (int)n);   // empty tile -> skipped (tiling.py:149-150)
    }
    OBIA_TRY(slic_plan_and_seed(ctx, b, nseg, &nvalid));
    debug_sync(ctx, "tiler: plan_and_seed");
    if (!white && S.pf.d_feat && !S.pf.launched && white_features_beside()) OBIA_TRY(prefetch_white_launch(ctx, S, true));
    OBIA_TRY(slic_run_sweeps(ctx, b, 1));   // (the orphan flag is looked at after the connectivity stage's own synchronisation)
    debug_sync(ctx, "tiler: sweeps");
    int n_new = 0;
    CcResolve resolve{};
    if (S.sp.enforce_connectivity) {
        std::vector<CcProblem> cps(np);
        for (int p = 0; p < np; ++p) {
            const SlicProblem &P = b.probs[p];
            const double segment_size = P.K > 0 ? (double)P.n_valid / (double)P.K : 1.0;
            const double mxd = S.sp.max_size_factor * segment_size;
            const int mx = mxd >= 2147483647.0 ? 2147483647 : (int)mxd;
            cps[p] = CcProblem{P.H, P.W, P.pix_off, (int)(S.sp.min_size_factor * segment_size), mx > 0 ? mx : 1};
        }
        OBIA_TRY(enforce_connectivity_batch(ctx, cps, b.d_labels, b.total_pix, 1, d_final, &n_new, &resolve));
        bool repeat = false;
        OBIA_TRY(slic_sweeps_settle(ctx, b, &repeat));
        if (repeat) {   // rare: a valid pixel no window reached kept a label that was not stored -- sweeps with stored labels, stage again
            OBIA_TRY(slic_run_sweeps(ctx, b, 2));
            OBIA_TRY(enforce_connectivity_batch(ctx, cps, b.d_labels, b.total_pix, 1, d_final, &n_new, &resolve));
        }
        debug_sync(ctx, "tiler: connectivity");
    } else {
        set_error("the tiled driver needs enforce_connectivity=True (segments must be connected pixel sets)");
        return OBIA_E_UNSUPPORTED;
    }
    if (S.next_id + n_new > S.id_cap) { set_error("segment id capacity exceeded (%d + %d > %d)", S.next_id, n_new, S.id_cap); return OBIA_E_NOMEM; }
    if (n_new > 0) {
        int sblocks = 1;
        for (auto &t : wins) sblocks = std::max(sblocks, cdiv(t.w, 64) * cdiv(t.h, 64));
        hipLaunchKernelGGL(tile_scatter_kernel, dim3(sblocks, np), dim3(64), 0, ctx->stream, d_wins, resolve, S.G, S.W,
                           S.next_id - 1, S.seg_size);
        OBIA_HIP_TRY(hipMemsetAsync(S.alive + S.next_id, 1, (size_t)n_new, ctx->stream));
        S.next_id += n_new;
    }
    OBIA_HIP_TRY(hipGetLastError());
    // (no synchronisation here: every host table of the batch went through upload_async, and the arena is reused in stream order)
    A.rewind(mk);
    return OBIA_OK;
}

// ---- seam import (sharded driver): wire codes -> local ids -------------------------------------------------------------------
// 1. every code is translated where it can be (my own ids, ids of `owner` that are in the map); a code of `owner` whose id is not in
//    the map yet claims its map entry (0 -> -1: one claimant per id) and joins the list of new ids; the largest owner id on the
//    seam goes to ctr[1].  An import that meets no new id -- the write-back imports, half of all -- is complete after this kernel.
__global__ __launch_bounds__(256) void seam_mark_kernel(const int32_t *__restrict__ codes, int n, int me, int owner, int32_t *__restrict__ fmap,
                                                        int cap, int *__restrict__ new_list, int *__restrict__ ctr, int32_t *__restrict__ ids) {
    int tmax = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c = codes[i];
        int id = 0;
        if (c > 0) {
            const int o = (c >> 24) - 1, t = c & 0xffffff;
            if (o == me) id = t;
            else if (o == owner) {
                tmax = max(tmax, t);
                if (t > 0 && t < cap) {
                    const int f = fmap[t];
                    if (f > 0) id = f;
                    else if (f == 0 && atomicCAS(&fmap[t], 0, -1) == 0) new_list[atomicAdd(&ctr[0], 1)] = t;
                }
            }
        }
        ids[i] = id;      // (0 for a new id: seam_translate_kernel fills it in)
    }
    for (int off = 32; off > 0; off >>= 1) tmax = max(tmax, __shfl_xor(tmax, off));
    if ((threadIdx.x & 63) == 0 && tmax > 0) atomicMax(&ctr[1], tmax);
}
// 2. a new id's local id = first + its rank among the new ids (ascending ids of the owner); its wire code is noted.  One workgroup sorts
//    the list in LDS (bitonic, up to SEAM_SORT entries: a seam of BASELINE configs[3] brings ~7 000 new ids); longer lists are ranked
//    by counting (seam_assign_count_kernel).
constexpr int SEAM_SORT = 16384;
__global__ __launch_bounds__(1024) void seam_assign_sort_kernel(int *__restrict__ new_list, int m, int first, int owner,
                                                                int32_t *__restrict__ fmap, int32_t *__restrict__ code_of) {
    __shared__ int s[SEAM_SORT];
    int np2 = 1;
    while (np2 < m) np2 <<= 1;
    for (int i = threadIdx.x; i < np2; i += 1024) s[i] = i < m ? new_list[i] : 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int a = s[i], b = s[ixj];
                    if (((i & k) == 0) ? (a > b) : (a < b)) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < m; i += 1024) {
        const int t = s[i];
        fmap[t] = first + i;
        code_of[first + i] = t | ((owner + 1) << 24);
    }
}
__global__ __launch_bounds__(256) void seam_assign_count_kernel(const int *__restrict__ new_list, int m, int first, int owner,
                                                                int32_t *__restrict__ fmap, int32_t *__restrict__ code_of) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int t = new_list[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += new_list[j] < t;
        fmap[t] = first + rank;
        code_of[first + rank] = t | ((owner + 1) << 24);
    }
}
// 3. the codes of the new ids, translated (only launched when there are new ids)
__global__ __launch_bounds__(256) void seam_translate_kernel(const int32_t *__restrict__ codes, int n, int owner,
                                                             const int32_t *__restrict__ fmap, int cap, int32_t *__restrict__ ids) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (ids[i] != 0) continue;
        const int c = codes[i];
        if (c <= 0 || (c >> 24) - 1 != owner) continue;
        const int t = c & 0xffffff;
        if (t > 0 && t < cap) ids[i] = fmap[t];
    }
}
__global__ void seam_register_kernel(unsigned *__restrict__ seg_size, uint8_t *__restrict__ alive, int first, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) { seg_size[first + i] = 0xffffffffu; alive[first + i] = 1; }
}

static int tiler_check(const float *img, int H, int W, int C, int Hg, int row0, const obia_tiling_params *tp,
                       const obia_slic_params *sp, const int32_t *labels) {
    if (!img || !tp || !sp || !labels) { set_error("null argument"); return OBIA_E_INVALID; }
    if (H <= 0 || W <= 0 || C <= 0 || C > 16) { set_error("bad raster shape (%d,%d,%d)", H, W, C); return OBIA_E_INVALID; }
    if (row0 < 0 || row0 + H > Hg) { set_error("local rows [%d,%d) outside the global raster of %d rows", row0, row0 + H, Hg); return OBIA_E_INVALID; }
    if ((long long)H * W > 0x7fffffffLL) { set_error("raster above 2^31 pixels: shard it across GPUs"); return OBIA_E_INVALID; }
    if (tp->tile_size <= 0 || tp->buffer < 0) { set_error("tile_size must be positive and buffer non-negative"); return OBIA_E_INVALID; }
    if (!(sp->compactness > 0.0) || sp->max_num_iter < 0) { set_error("bad SLIC parameters"); return OBIA_E_INVALID; }
    for (int i = 0; i < 3; ++i) {
        if (!(sp->sigma_zyx[i] >= 0.0)) { set_error("sigma must be >= 0"); return OBIA_E_INVALID; }
        if (!(sp->spacing_zyx[i] > 0.0) || !(sp->spacing_zyx[i] < 1.0e30)) { set_error("spacing must be positive and finite"); return OBIA_E_INVALID; }
    }
    if (!sp->enforce_connectivity) { set_error("the tiled driver needs enforce_connectivity=True (segments must be connected pixel sets)"); return OBIA_E_UNSUPPORTED; }
    if (sp->n_segments <= 0 && !(tp->crown_radius > 0.0 && tp->pixel_width > 0.0 && tp->pixel_height > 0.0)) {
        set_error("crown_radius and pixel size must be positive when n_segments is not given");
        return OBIA_E_INVALID;
    }
    return OBIA_OK;
}

static int tiler_init(obia_ctx *ctx, TileState &S, const float *img, const uint8_t *mask, int H, int W, int C, int Hg,
                      int row0, const obia_tiling_params *tp, const obia_slic_params *sp, int32_t *labels, int extra_ids) {
    OBIA_TRY(tiler_check(img, H, W, C, Hg, row0, tp, sp, labels));
    const int T = tp->tile_size, B = tp->buffer;
    Arena &A = ctx->arena;
    // capacity for provisional ids: every tile can create at most its grid of seeds
    const int ntx = cdiv(W, T), nty = cdiv(H, T) + 2;
    const double pixel_area = tp->pixel_width * tp->pixel_height;
    const double crown_area = M_PI * tp->crown_radius * tp->crown_radius;
    const long long wh = (long long)(T + 2 * B) * (T + 2 * B);
    double n = sp->n_segments > 0 ? (double)sp->n_segments * (double)wh / ((double)T * T) : (double)wh * pixel_area / crown_area;
    if (n < 1) n = 1;
    // masked seeding lays a grid for n_eff ~ n over the window; rounding of the step can add ~(1 + 1/S)^2
    long long cap = (long long)(n * 1.5 + 64.0) * (long long)ntx * nty + 16 + extra_ids;
    if (cap > 0x7ffffff0LL) { set_error("too many segments for int32 ids"); return OBIA_E_INVALID; }
    S.H = H; S.W = W; S.C = C; S.Hg = Hg; S.row0 = row0; S.img = img; S.inmask = mask; S.G = labels; S.tp = *tp; S.sp = *sp;
    S.id_cap = (int)cap; S.next_id = 1;
    S.seg_size = A.get<unsigned>((size_t)cap);
    S.inside = A.get<unsigned>((size_t)cap);
    S.alive = A.get<uint8_t>((size_t)cap);
    S.newid = A.get<int>((size_t)cap);
    S.d_total = A.get<long long>(1);
    if (!S.seg_size || !S.inside || !S.alive || !S.newid || !S.d_total) return OBIA_E_NOMEM;
    OBIA_HIP_TRY(hipMemsetAsync(S.seg_size, 0, sizeof(unsigned) * (size_t)cap, ctx->stream));
    OBIA_HIP_TRY(hipMemsetAsync(S.alive, 0, (size_t)cap, ctx->stream));
    // corner squares: side buffer/2 in MAP units (tiling.py:189), i.e. buffer/2/pixel_size pixels; a pixel is
    // inside when its centre is (rasterize default all_touched=False)
    S.clx = S.cly = S.clx_in = S.cly_in = S.clx_any = S.cly_any = 0;
    if (B > 0) {
        const double cl = (double)B / 2.0;
        // pixel k of an axis (counted from the square's outer edge) spans [k * px, (k + 1) * px) map units; the three counts are the
        // numbers of k with  centre (k + 0.5) * px < cl,  far edge (k + 1) * px <= cl,  near edge k * px < cl  -- evaluated as written,
        // in doubles, so that a quotient like 3.5 / 0.7 landing a hair beside a whole number cannot move a count
        auto count = [](double px, double cl, int kind) {
            if (!(px > 0.0)) px = 1.0;
            long long k = (long long)std::floor(cl / px) + 2;
            if (k < 0) k = 0;
            auto holds = [&](long long i) { return kind == 0 ? (i + 0.5) * px < cl : (kind == 1 ? (i + 1.0) * px <= cl : i * px < cl); };
            while (k > 0 && !holds(k - 1)) --k;     // k = number of pixels 0 .. k-1 for which the statement holds (it is monotone in k)
            return (int)std::min<long long>(k, 0x3fffffff);
        };
        S.clx = count(tp->pixel_width, cl, 0);  S.cly = count(tp->pixel_height, cl, 0);
        S.clx_in = count(tp->pixel_width, cl, 1);  S.cly_in = count(tp->pixel_height, cl, 1);
        S.clx_any = count(tp->pixel_width, cl, 2);  S.cly_any = count(tp->pixel_height, cl, 2);
    }
    return OBIA_OK;
}

// window of the white tile (tj, ti): grown by `buffer`, clamped to the GLOBAL raster, in local rows
static TileWin white_window(const TileState &S, int tj, int ti) {
    const int T = S.tp.tile_size, B = S.tp.buffer;
    const int y0 = std::max(0, tj * T - B), y1 = std::min(S.Hg, tj * T + T + B);
    const int x0 = std::max(0, ti * T - B), x1 = std::min(S.W, ti * T + T + B);
    const int h = y1 - y0, w = x1 - x0;
    return TileWin{y0 - S.row0, x0, h, w, 0, std::min(S.cly, h), std::min(S.clx, w), std::min(S.cly_in, h), std::min(S.clx_in, w),
                   std::min(S.cly_any, h), std::min(S.clx_any, w), 0};
}

// The feature pass of ALL white tiles runs as ONE batch before the white rows (it depends on the raster only), in the
// order in which the white batches will consume the windows: one large launch instead of one per tile row, one read-back
// instead of eight.  plan: windows and buffers (allocated before any batch scope of the arena); launch; fetch: the keys.
// (Running it on a second stream beside the black sweeps was measured: the HBM-bound pass and the VALU-bound sweeps slow
// each other down by more than the overlap gains -- 66.5 vs 63.2 ms per step -- so it stays on the main stream.)
static int prefetch_white_plan(obia_ctx *ctx, TileState &S, int white_order) {
    const int T = S.tp.tile_size;
    const int ntx = cdiv(S.W, T), nty = cdiv(S.Hg, T);
    TileState::PreFeat &pf = S.pf;
    pf = TileState::PreFeat();
    if (S.sp.sigma_zyx[0] > 0.0 || S.sp.sigma_zyx[1] > 0.0 || S.sp.sigma_zyx[2] > 0.0) return OBIA_OK;   // Gaussian pre-smoothing: per-batch features (slic_prepare_features holds the smoothing passes)
    for (int cls = 0; cls < (white_order == 1 ? 2 : 1); ++cls)
        for (int tj = 0; tj < nty; ++tj) {
            if (white_order == 1 && (tj & 1) != cls) continue;
            for (int ti = 0; ti < ntx; ++ti) {
                if ((ti + tj) % 2 == 0) continue;
                const TileWin t = white_window(S, tj, ti);
                if (t.h > 0 && t.w > 0) pf.wins.push_back(t);
            }
        }
    const size_t NP = pf.wins.size();
    if (NP == 0) return OBIA_OK;
    long long off = 0, foff = 0, boff = 0;
    const int CP = (S.C + 3) & ~3;
    pf.windows.resize(NP);
    for (size_t p = 0; p < NP; ++p) {
        const TileWin &t = pf.wins[p];
        if (t.y0 < 0 || t.y0 + t.h > S.H) { pf = TileState::PreFeat(); return OBIA_OK; }   // the batches report the halo error
        pf.windows[p] = SrcWindow{t.y0, t.x0, t.h, t.w, off, foff, boff};
        off += (long long)t.h * t.w;
        foff += feat_block_f4(t.h, t.w, CP);
        boff += feat_boxes(t.h, t.w);
        if (t.h > pf.maxh) pf.maxh = t.h;
    }
    if ((double)foff * 16.0 > 32.0 * 1024 * 1024 * 1024) { pf = TileState::PreFeat(); return OBIA_OK; }   // too big to hold: per-batch features
    Arena &A = ctx->arena;
    const size_t ntot = NP * (size_t)S.C * 2 + 2 * NP;
    pf.d_windows = A.get<SrcWindow>(NP);
    pf.d_keys = A.get<unsigned>(ntot);
    pf.d_feat = A.get<float>(4 * (size_t)foff);
    if (!pf.d_windows || !pf.d_keys || !pf.d_feat) return OBIA_E_NOMEM;
    if (slic_use_colour_bound((float)(1.0 / S.sp.compactness), S.C == 3 && S.sp.convert2lab != 0) && !S.sp.slic_zero && !S.sp.exit_on_fixed_point &&
        (float)S.sp.spacing_zyx[1] == 1.0f && (float)S.sp.spacing_zyx[2] == 1.0f) {
        pf.d_fbox = A.get<float>((size_t)boff * 2 * CP);
        if (!pf.d_fbox) return OBIA_E_NOMEM;
    }
    pf.host.resize(ntot);
    return OBIA_OK;
}

static int prefetch_white_launch(obia_ctx *ctx, TileState &S, bool beside) {
    TileState::PreFeat &pf = S.pf;
    if (pf.launched || !pf.d_feat) return OBIA_OK;
    pf.launched = true;
    const size_t NP = pf.wins.size();
    const int CP = (S.C + 3) & ~3;
    OBIA_TRY(upload_async(ctx, pf.d_windows, pf.windows.data(), sizeof(SrcWindow) * NP));
    const int to_lab = (S.C == 3 && S.sp.convert2lab != 0) ? 1 : 0;
    if (beside) {
        // `beside`: on a side stream, forked here and joined in prefetch_white_fetch -- the caller queues the black batch's
        // spatial pre-pass next, a kernel that is bound by vector issue and moves no data, beside this pass, which is bound by HBM
        OBIA_TRY(side_streams(ctx, obia_ctx::MAX_SIDE));
        hipStream_t side = ctx->side[obia_ctx::MAX_SIDE - 1];
        OBIA_HIP_TRY(hipEventRecord(ctx->aux_fork, ctx->stream));
        OBIA_HIP_TRY(hipStreamWaitEvent(side, ctx->aux_fork, 0));
        OBIA_TRY(slic_features_launch(side, S.C, CP, (int)NP, pf.d_windows, pf.maxh, S.img, S.W, 1, to_lab,
                                      (float)(1.0 / S.sp.compactness) * slic_prescale((float)(1.0 / S.sp.compactness), 1, to_lab, S.sp.slic_zero != 0), pf.d_feat, pf.d_keys, true, pf.d_fbox));
        OBIA_HIP_TRY(hipEventRecord(ctx->aux_join, side));
        pf.on_side = true;
        return OBIA_OK;
    }
    ScopedSpan span(ctx, T_FEAT);
    OBIA_TRY(slic_features_launch(ctx->stream, S.C, CP, (int)NP, pf.d_windows, pf.maxh, S.img, S.W, 1, to_lab,
                                  (float)(1.0 / S.sp.compactness) * slic_prescale((float)(1.0 / S.sp.compactness), 1, to_lab, S.sp.slic_zero != 0), pf.d_feat, pf.d_keys, true, pf.d_fbox));
    return OBIA_OK;
}

static int prefetch_white_fetch(obia_ctx *ctx, TileState &S) {
    if (!S.pf.d_feat) return OBIA_OK;
    OBIA_TRY(prefetch_white_launch(ctx, S));
    if (S.pf.on_side) { OBIA_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->aux_join, 0)); S.pf.on_side = false; }
    OBIA_TRY(read_back(ctx, S.pf.host.data(), S.pf.d_keys, S.pf.host.size() * sizeof(unsigned)));
    S.pf.ready = true;
    S.pf.cursor = 0;
    return OBIA_OK;
}

// Tiles of GLOBAL tile rows [tr_lo, tr_hi) with (tr % 2 == parity) when parity is 0 or 1.  Black tiles: exact
// windows, one batch for the whole range.  White tiles: windows grown by `buffer` and clamped to the GLOBAL
// raster, one batch per tile-row (windows of one tile-row never overlap; those of adjacent rows overlap at
// corners).  Every window must lie inside the local rows.
static int tiler_run(obia_ctx *ctx, TileState &S, bool white, int tr_lo, int tr_hi, int parity) {
    const int T = S.tp.tile_size, B = S.tp.buffer;
    const int ntx = cdiv(S.W, T), nty = cdiv(S.Hg, T);
    if (tr_lo < 0) tr_lo = 0;
    if (tr_hi > nty) tr_hi = nty;
    // white tile rows of ONE parity class do not see each other (their windows are T - 2B rows apart): the whole class is
    // one batch; in raster order every row sees the corner overlaps of the row before it: one batch per row
    const bool per_row = !(parity >= 0 && T > 2 * B);
    // tile_size <= 2 * buffer: the grown windows of two white tiles of ONE tile row overlap (they are T - 2B columns apart), so
    // even a row cannot be one batch -- every white tile runs on its own, in the reference's order (tiling.py:156-287)
    const bool per_tile = white && T <= 2 * B;
    std::vector<TileWin> wins;
    auto flush = [&]() -> int {
        for (auto &t : wins)
            if (t.y0 < 0 || t.y0 + t.h > S.H) { set_error("tile window rows [%d,%d) (local) outside the %d local rows: halo too small", t.y0, t.y0 + t.h, S.H); return OBIA_E_INVALID; }
        int rc = run_tile_batch(ctx, S, wins, white);
        wins.clear();
        return rc;
    };
    for (int tj = tr_lo; tj < tr_hi; ++tj) {
        if (parity >= 0 && (tj & 1) != parity) continue;
        for (int ti = 0; ti < ntx; ++ti) {
            const bool is_white = (ti + tj) % 2 != 0;
            if (is_white != white) continue;
            if (!white) {
                TileWin t{tj * T - S.row0, ti * T, std::min(T, S.Hg - tj * T), std::min(T, S.W - ti * T), 0, 0, 0, 0, 0, 0, 0, 0};
                if (t.h > 0 && t.w > 0) wins.push_back(t);
            } else {
                const TileWin t = white_window(S, tj, ti);
                if (t.h > 0 && t.w > 0) wins.push_back(t);
                if (per_tile && !wins.empty()) OBIA_TRY(flush());
            }
        }
        if (white && per_row) OBIA_TRY(flush());
    }
    if (!wins.empty()) OBIA_TRY(flush());
    return OBIA_OK;
}

// ids 1..N: black survivors first, then white, each in creation order (tiling.py:289-290)
static int tiler_finalize(obia_ctx *ctx, TileState &S, int64_t *n_segments_out) {
    const int nchunks = cdiv(S.next_id, IDS_CHUNK);
    int *d_partial = ctx->arena.get<int>((size_t)nchunks);
    if (!d_partial) return OBIA_E_NOMEM;
    hipLaunchKernelGGL(ids_count_kernel, dim3(nchunks), dim3(1024), 0, ctx->stream, S.alive, S.next_id, d_partial);
    hipLaunchKernelGGL(ids_scan_kernel, dim3(nchunks), dim3(1024), 0, ctx->stream, S.alive, S.next_id, d_partial, S.newid, S.d_total);
    int g = cdiv((long long)S.H * S.W, 256 * 16);
    if (g > 65535) g = 65535;
    hipLaunchKernelGGL(ids_apply_kernel, dim3(g), dim3(256), 0, ctx->stream, S.G, (long long)S.H * S.W, S.newid);
    OBIA_HIP_TRY(hipGetLastError());
    long long total = 0;
    OBIA_TRY(read_back(ctx, &total, S.d_total, sizeof(long long)));
    if (n_segments_out) *n_segments_out = total;
    return OBIA_OK;
}

static int tiled_slic_dev(obia_ctx *ctx, const float *img, const uint8_t *mask, int H, int W, int C,
                          const obia_tiling_params *tp, const obia_slic_params *sp, int32_t *labels_out,
                          int64_t *n_segments_out) {
    TileState S;
    OBIA_TRY(tiler_init(ctx, S, img, mask, H, W, C, H, 0, tp, sp, labels_out, 0));
    OBIA_HIP_TRY(hipMemsetAsync(labels_out, 0, sizeof(int32_t) * (size_t)H * W, ctx->stream));
    const int nty = cdiv(H, tp->tile_size);
    OBIA_TRY(prefetch_white_plan(ctx, S, tp->white_order));               // features of all white tiles: one batch
    OBIA_TRY(tiler_run(ctx, S, false, 0, nty, -1));                       // pass 1: black tiles
    if (std::getenv("OBIA_DEBUG_FAIL_AFTER_BLACK")) {   // test hook (tests/test_gpu_edge_cases.py): an error while the white feature pass is still on its side stream
        set_error("debug: forced failure after the black pass");
        return OBIA_E_INVALID;
    }
    OBIA_TRY(prefetch_white_fetch(ctx, S));
    if (tp->white_order == 1) {                                           // pass 2, two parity classes of tile rows
        OBIA_TRY(tiler_run(ctx, S, true, 0, nty, 0));
        OBIA_TRY(tiler_run(ctx, S, true, 0, nty, 1));
    } else {
        OBIA_TRY(tiler_run(ctx, S, true, 0, nty, -1));                    // pass 2, the reference's raster order
    }
    return tiler_finalize(ctx, S, n_segments_out);
}

}  // namespace obia

struct obia_tiler {
    obia_ctx *ctx;
    obia::TileState S;
    int *seam_buf = nullptr;      // scratch of obia_tiler_import_seam: two counters + the list of new ids
    size_t seam_cap = 0;
};

// Error path: nothing this library queued may still be running when the caller gets the error back (the white tiles'
// feature pass runs on a side stream beside the black batch and is joined only when the white pass starts -- a failure in
// between returned while that pass could still read the caller's raster and write arena memory the next call hands out again).
static int fail_quiesced(obia_ctx *ctx, int rc) {
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < obia_ctx::MAX_SIDE; ++i)
        if (ctx->side[i]) (void)hipStreamSynchronize(ctx->side[i]);
    return rc;
}

using namespace obia;

extern "C" {

obia_tiler *obia_tiler_create(obia_ctx *ctx, const float *img_local, const uint8_t *mask_local, int H_local, int W, int C,
                              int H_global, int row0, const obia_tiling_params *tiling, const obia_slic_params *params,
                              int32_t *labels_local, int extra_ids) {
    if (!ctx) { set_error("null context"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    ctx->arena.reset();
    begin_timing(ctx);
    obia_tiler *t = new obia_tiler();
    t->ctx = ctx;
    if (tiler_init(ctx, t->S, img_local, mask_local, H_local, W, C, H_global, row0, tiling, params, labels_local,
                   extra_ids < 0 ? 0 : extra_ids) != OBIA_OK) { delete t; return nullptr; }
    return t;
}

void obia_tiler_destroy(obia_tiler *t) {
    if (!t) return;
    (void)fail_quiesced(t->ctx, 0);   // (a session dropped half-way may have left the white feature pass on its side stream)
    if (t->seam_buf) (void)hipFree(t->seam_buf);
    resolve_timing(t->ctx);
    delete t;
}

int obia_tiler_run(obia_tiler *t, int white, int tile_row_lo, int tile_row_hi, int row_parity) {
    if (!t) { set_error("null tiler"); return OBIA_E_INVALID; }
    if (hipSetDevice(t->ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return OBIA_E_HIP; }
    const int rc = tiler_run(t->ctx, t->S, white != 0, tile_row_lo, tile_row_hi, row_parity);
    if (rc != OBIA_OK) return fail_quiesced(t->ctx, rc);
    OBIA_HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    return OBIA_OK;
}

int obia_tiler_next_id(obia_tiler *t) { return t ? t->S.next_id : -1; }

int obia_tiler_set_segments(obia_tiler *t, int first_id, int count, const uint32_t *sizes_dev) {
    if (!t || !sizes_dev || first_id < 1 || count < 0) { set_error("bad arguments"); return OBIA_E_INVALID; }
    if (first_id + count > t->S.id_cap) { set_error("segment id capacity exceeded (%d + %d > %d)", first_id, count, t->S.id_cap); return OBIA_E_NOMEM; }
    if (count == 0) return OBIA_OK;
    OBIA_HIP_TRY(hipMemcpyAsync(t->S.seg_size + first_id, sizes_dev, sizeof(uint32_t) * (size_t)count, hipMemcpyDeviceToDevice, t->ctx->stream));
    OBIA_HIP_TRY(hipMemsetAsync(t->S.alive + first_id, 1, (size_t)count, t->ctx->stream));
    if (first_id + count > t->S.next_id) t->S.next_id = first_id + count;
    OBIA_HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    return OBIA_OK;
}

int obia_tiler_get_alive(obia_tiler *t, uint8_t *alive_out_dev, int count) {
    if (!t || !alive_out_dev || count < 0 || count > t->S.id_cap) { set_error("bad arguments"); return OBIA_E_INVALID; }
    OBIA_HIP_TRY(hipMemcpyAsync(alive_out_dev, t->S.alive, (size_t)count, hipMemcpyDeviceToDevice, t->ctx->stream));
    OBIA_HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    return OBIA_OK;
}

int obia_tiler_set_alive(obia_tiler *t, const uint8_t *alive_in_dev, int count) {
    if (!t || !alive_in_dev || count < 0 || count > t->S.id_cap) { set_error("bad arguments"); return OBIA_E_INVALID; }
    OBIA_HIP_TRY(hipMemcpyAsync(t->S.alive, alive_in_dev, (size_t)count, hipMemcpyDeviceToDevice, t->ctx->stream));
    OBIA_HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    return OBIA_OK;
}

int obia_tiler_import_seam(obia_tiler *t, const int32_t *codes_dev, int n, int my_rank, int owner_rank,
                           int32_t *fmap_dev, int fmap_cap, int32_t *code_of_dev, int32_t *ids_out_dev,
                           int *first_new_out, int *n_new_out, int *max_owner_id_out) {
    if (!t || !codes_dev || !fmap_dev || !code_of_dev || !ids_out_dev || n < 0 || fmap_cap < 1 || my_rank < 0 || owner_rank < 0 ||
        my_rank > 126 || owner_rank > 126 || !first_new_out || !n_new_out || !max_owner_id_out) { set_error("bad arguments"); return OBIA_E_INVALID; }
    obia_ctx *ctx = t->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return OBIA_E_HIP; }
    *first_new_out = t->S.next_id; *n_new_out = 0; *max_owner_id_out = 0;
    if (n == 0) return OBIA_OK;
    // scratch outside the arena (the session's arena holds the tiler's persistent state): one small allocation kept by the session
    if (t->seam_cap < (size_t)n + 2) {
        if (t->seam_buf) (void)hipFree(t->seam_buf);
        t->seam_buf = nullptr; t->seam_cap = 0;
        if (hipMalloc(&t->seam_buf, sizeof(int) * ((size_t)n + 2)) != hipSuccess) { set_error("seam scratch allocation failed"); return OBIA_E_NOMEM; }
        t->seam_cap = (size_t)n + 2;
    }
    int *ctr = t->seam_buf, *new_list = t->seam_buf + 2;
    OBIA_HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(int) * 2, ctx->stream));
    const int blocks = std::min(cdiv(n, 256), 2048);
    hipLaunchKernelGGL(seam_mark_kernel, dim3(blocks), dim3(256), 0, ctx->stream, codes_dev, n, my_rank, owner_rank, fmap_dev, fmap_cap, new_list, ctr,
                       ids_out_dev);
    int h[2] = {0, 0};
    OBIA_TRY(read_back(ctx, h, ctr, sizeof(h)));   // the one read-back of an import: how many ids are new, the largest id on the seam
    *max_owner_id_out = h[1];
    const int n_new = h[0];
    if (t->S.next_id + n_new > t->S.id_cap) { set_error("segment id capacity exceeded (%d + %d > %d)", t->S.next_id, n_new, t->S.id_cap); return OBIA_E_NOMEM; }
    if (n_new > 0) {
        if (n_new <= SEAM_SORT)
            hipLaunchKernelGGL(seam_assign_sort_kernel, dim3(1), dim3(1024), 0, ctx->stream, new_list, n_new, t->S.next_id, owner_rank, fmap_dev, code_of_dev);
        else
            hipLaunchKernelGGL(seam_assign_count_kernel, dim3(256), dim3(256), 0, ctx->stream, new_list, n_new, t->S.next_id, owner_rank, fmap_dev, code_of_dev);
        hipLaunchKernelGGL(seam_translate_kernel, dim3(blocks), dim3(256), 0, ctx->stream, codes_dev, n, owner_rank, fmap_dev, fmap_cap, ids_out_dev);
        // registered like obia_tiler_set_segments(first, n_new, 0xffffffff ...): sizes follow from the caller's halo
        hipLaunchKernelGGL(seam_register_kernel, dim3(cdiv(n_new, 256)), dim3(256), 0, ctx->stream, t->S.seg_size, t->S.alive, t->S.next_id, n_new);
        t->S.next_id += n_new;
        // (the session's entry points return with their work done -- the caller reads ids_out on a stream of its own; an import
        // that brought nothing new was complete at the read-back above)
        OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    *n_new_out = n_new;   // (when h[1] >= fmap_cap the ids beyond the map were not translated: the caller calls again with a larger map;
                          // what this call imported stays imported)
    OBIA_HIP_TRY(hipGetLastError());
    return OBIA_OK;
}

int obia_tiler_finalize(obia_tiler *t, int64_t *n_segments_out) {
    if (!t) { set_error("null tiler"); return OBIA_E_INVALID; }
    const int rc = tiler_finalize(t->ctx, t->S, n_segments_out);
    return rc == OBIA_OK ? rc : fail_quiesced(t->ctx, rc);
}

int obia_tiled_slic_f32_dev(obia_ctx *ctx, const float *img, const uint8_t *mask, int H, int W, int C,
                            const obia_tiling_params *tiling, const obia_slic_params *params, int32_t *labels_out,
                            int64_t *n_segments_out) {
    if (!ctx) { set_error("null context"); return OBIA_E_INVALID; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return OBIA_E_HIP; }
    ctx->arena.reset();
    begin_timing(ctx);
    int rc;
    {
        ScopedSpan total(ctx, T_TOTAL);
        rc = tiled_slic_dev(ctx, img, mask, H, W, C, tiling, params, labels_out, n_segments_out);
    }
    if (rc != OBIA_OK) return fail_quiesced(ctx, rc);
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    resolve_timing(ctx);
    return OBIA_OK;
}

int obia_tiled_slic_f32(obia_ctx *ctx, const float *img, const uint8_t *mask, int H, int W, int C,
                        const obia_tiling_params *tiling, const obia_slic_params *params, int32_t *labels_out,
                        int64_t *n_segments_out) {
    if (!ctx) { set_error("null context"); return OBIA_E_INVALID; }
    if (!img || !labels_out || H <= 0 || W <= 0 || C <= 0) { set_error("bad arguments"); return OBIA_E_INVALID; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return OBIA_E_HIP; }
    const size_t npix = (size_t)H * W;
    float *d_img = nullptr; uint8_t *d_mask = nullptr; int32_t *d_lab = nullptr;
    int rc = OBIA_OK;
    if (hipMalloc(&d_img, npix * C * sizeof(float)) != hipSuccess || hipMalloc(&d_lab, npix * sizeof(int32_t)) != hipSuccess ||
        (mask && hipMalloc(&d_mask, npix) != hipSuccess)) {
        set_error("device allocation for host-pointer call failed");
        rc = OBIA_E_NOMEM;
    }
    if (rc == OBIA_OK && hipMemcpyAsync(d_img, img, npix * C * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = OBIA_E_HIP;
    if (rc == OBIA_OK && mask && hipMemcpyAsync(d_mask, mask, npix, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = OBIA_E_HIP;
    if (rc == OBIA_E_HIP) set_error("host->device copy failed in obia_tiled_slic_f32");
    if (rc == OBIA_OK) rc = obia_tiled_slic_f32_dev(ctx, d_img, d_mask, H, W, C, tiling, params, d_lab, n_segments_out);
    if (rc == OBIA_OK && hipMemcpyAsync(labels_out, d_lab, npix * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
        set_error("device->host copy failed in obia_tiled_slic_f32");
        rc = OBIA_E_HIP;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_img); (void)hipFree(d_lab); (void)hipFree(d_mask);
    return rc;
}

}  // extern "C"
