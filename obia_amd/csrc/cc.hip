// cc.hip -- connectivity enforcement of a SLIC label map on gfx950.
//
// Restates _enforce_label_connectivity_cython (scikit-image _slic.pyx, called from
// slic_superpixels.py:320-328; oracle/obia_oracle.c: obia_oracle_enforce_connectivity).  The reference
// is a raster-order sequential BFS.  What is order-INDEPENDENT in it is reproduced exactly:
//   * components are the 4-connected regions of equal label (masked pixels excluded);
//   * a component keeps its own label iff size >= min_size, and surviving components are numbered
//     consecutively from start_label in raster order of their FIRST pixel
//     (union-find by minimum pixel index -> the root IS the first pixel; ranks by a prefix sum);
//   * a component smaller than min_size takes the label of `adjacent`: the LAST neighbour pixel, in
//     the BFS order of the reference (neighbour order x+1, x-1, y+1, y-1), that already carries a
//     label -- i.e. belongs to a component whose first pixel precedes this BFS's start pixel.  Small
//     components are rare and small, so one lane replays the BFS of each of them exactly; chains
//     small -> small -> ... -> survivor are resolved afterwards.
//   * a component that reaches max_size is cut the way the reference cuts it: its BFS (neighbour order x+1, x-1, y+1,
//     y-1, first-in first-out) stops at max_size pixels, the raster scan later meets the next unlabelled pixel of the
//     region and starts another capped BFS from there.  Such components are rare; one lane replays each of them
//     (cc_split_kernel) and rewrites parent / size so that every piece is a component of its own for the stages below.
// Not reproduced exactly (DESIGN.md "connectivity"): the bookkeeping of small components that find no labelled
// neighbour on their first BFS and are re-seeded from a later pixel (replayed here with the same start pixels, but
// neighbours are classified by root order only).
#include "slic.hpp"

#include <cstdlib>

namespace obia {

// dense pixel index -> problem (binary search on pix_off; used on roots / small components only)
__device__ __forceinline__ int find_prob(const CcProblem *__restrict__ probs, int nprob, long long g) {
    int lo = 0, hi = nprob - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (probs[mid].pix_off <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ int ld_agent(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int find_root(const int *parent, int i) {
    int p;
    while ((p = ld_agent(&parent[i])) != i) i = p;
    return i;
}

// link the larger root under the smaller one; lock-free (Komura / Playne-Hawick style)
__device__ __forceinline__ void unite(int *parent, int a, int b) {
    for (;;) {
        a = find_root(parent, a);
        b = find_root(parent, b);
        if (a == b) return;
        if (a > b) { int t = a; a = b; b = t; }
        const int old = atomicMin(&parent[b], a);
        if (old == b) return;
        b = old;
    }
}

// ---- tile-local union-find in LDS -------------------------------------------------------------------------------
// A workgroup resolves the components of one 64x32 pixel tile in LDS (16-bit local parents; the labels stay in registers),
// then publishes parent[pixel] = GLOBAL index of the pixel's tile-local root.  Local indices are row-major inside
// the tile, so "smaller local index" == "smaller global index": the local root is the first pixel of the tile's
// part of the component, and linking larger roots under smaller ones across tiles (cc_seam_kernel) keeps the
// global invariant root == first pixel in raster order.
constexpr int CT_W = 64, CT_H = 32, CT_N = CT_W * CT_H;

__device__ __forceinline__ int lfind(const unsigned short *par, int i) {
    int p;
    while ((p = par[i]) != i) i = p;
    return i;
}
// 16-bit atomicMin on LDS (CAS on the enclosing 32-bit word); returns the previous value of par[i]
__device__ __forceinline__ int latomic_min16(unsigned short *par, int i, int v) {
    unsigned *wptr = reinterpret_cast<unsigned *>(par) + (i >> 1);
    const int sh = (i & 1) * 16;
    unsigned seen = *wptr;
    for (;;) {
        const int cur = (int)((seen >> sh) & 0xffffu);
        if (cur <= v) return cur;
        const unsigned prev = atomicCAS(wptr, seen, (seen & ~(0xffffu << sh)) | ((unsigned)v << sh));
        if (prev == seen) return cur;
        seen = prev;
    }
}
__device__ __forceinline__ void lunite(unsigned short *par, int a, int b) {
    for (;;) {
        a = lfind(par, a);
        b = lfind(par, b);
        if (a == b) return;
        if (a > b) { int t = a; a = b; b = t; }
        const int old = latomic_min16(par, b, a);
        if (old == b) return;   // b was a root and now hangs under a
        b = old;                // b had been linked elsewhere meanwhile: unite a with where it points
    }
}

// Round 4: the labels stay in REGISTERS.  A wave owns eight consecutive rows of the tile (lane = column), so the pixel above is the
// same lane's previous row, the pixel to the left one lane away (__shfl_up), and which contacts issue a union is decided on 64-bit
// lane masks in scalar registers: no label array in LDS, no four LDS reads per pixel for the neighbour tests (the kernel is bound by
// instruction issue: 8 pixels per lane, 20 KB of LDS per workgroup -> 12 KB).
__global__ __launch_bounds__(256) void cc_tile_kernel(const CcProblem *__restrict__ probs, const int32_t *__restrict__ lab,
                                                      int *__restrict__ parent, int *__restrict__ size, int mask_label,
                                                      unsigned long long *__restrict__ lrbits) {
    constexpr int RPW = CT_H / 4;   // rows per wave
    static_assert(CT_W == 64 && CT_H % 4 == 0, "a wave is one tile row wide");
    __shared__ unsigned short s_par[CT_N];
    __shared__ int s_cnt[CT_N];
    const CcProblem P = probs[blockIdx.y];
    const int tiles_x = (P.W + CT_W - 1) / CT_W;
    const int tile = blockIdx.x;
    if (tile >= tiles_x * ((P.H + CT_H - 1) / CT_H)) return;
    const int ty0 = (tile / tiles_x) * CT_H, tx0 = (tile % tiles_x) * CT_W;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = tx0 + lane, ly0 = wv * RPW;
    const bool xin = x < P.W;
    // 1. labels: rows ly0 - 1 .. ly0 + RPW - 1 (the row above the wave's first one is another wave's -- read again, a cache hit -- or,
    // for the first wave, another tile's: that contact is cc_seam_kernel's)
    int l[RPW + 1];
#pragma unroll
    for (int j = 0; j <= RPW; ++j) {
        const int y = ty0 + ly0 + j - 1;
        l[j] = (xin && y < P.H && (j > 0 || wv > 0)) ? lab[P.pix_off + (long long)y * P.W + x] : mask_label;
    }
    // the parent of a pixel is the head of its horizontal run, found with one ballot (the highest run start at or below the lane), so
    // the chains that lfind() walks only hop between run heads
    unsigned long long starts[RPW];
    int runlen[RPW];                // pixels of the horizontal run this lane heads (0: not a head, or masked)
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int i = (ly0 + j) * CT_W + lane;
        const int lc = l[j + 1];
        const int lft = __shfl_up(lc, 1);
        const bool start = (lc == mask_label) || lane == 0 || lft != lc;
        const unsigned long long all = __ballot(start);
        starts[j] = all;
        const unsigned long long below = all & (~0ull >> (63 - lane));
        s_par[i] = (unsigned short)(i - lane + (63 - __clzll((long long)below)));
        const unsigned long long above = lane < 63 ? (all >> (lane + 1)) : 0ull;      // run starts to the right of this lane
        const int end = above ? lane + 1 + __builtin_ctzll(above) : CT_W;
        runlen[j] = (start && lc != mask_label) ? end - lane : 0;
        s_cnt[i] = 0;
    }
    __syncthreads();
    // 2. vertical contacts: only the first pixel of a horizontal contact issues the union -- a contact whose left neighbour has the
    // same label in this row (no run start here) and a vertical contact of its own is covered by that neighbour's
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const bool vert = l[j + 1] != mask_label && l[j] == l[j + 1];
        const unsigned long long vs = __ballot(vert);
        const unsigned long long issue = vs & ~(~starts[j] & (vs << 1));
        if ((issue >> lane) & 1ull) { const int i = (ly0 + j) * CT_W + lane; lunite(s_par, i, i - CT_W); }
    }
    __syncthreads();
    // 3. pixel count of every tile-local component at its local root (one LDS add per horizontal run), then publish the
    // global index of the local root
    int rloc[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int i = (ly0 + j) * CT_W + lane;
        rloc[j] = l[j + 1] != mask_label ? lfind(s_par, i) : -1;
        if (runlen[j] > 0) atomicAdd(&s_cnt[rloc[j]], runlen[j]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int i = (ly0 + j) * CT_W + lane;
        const int y = ty0 + ly0 + j;
        if (y >= P.H || !xin) continue;
        const long long g = P.pix_off + (long long)y * P.W + x;
        const int r = rloc[j];
        parent[g] = r >= 0 ? (int)(P.pix_off + (long long)(ty0 + r / CT_W) * P.W + tx0 + r % CT_W) : -1;
        // the pixel count of a tile-local component is written at its local root ONLY (a few words per tile instead of four bytes
        // per pixel written here and read again by a flatten pass), and the local roots are flagged in a bitmap: size[] of any
        // other pixel is never read (round 4)
        if (r == i) {
            size[g] = s_cnt[i];
            atomicOr(&lrbits[g >> 6], 1ull << (g & 63));
        }
    }
}

// contacts across tile borders: the pixels of the first row / first column of every tile against their upper / left
// neighbour (a few percent of the pixels), on the global parents
__global__ __launch_bounds__(256) void cc_seam_kernel(const CcProblem *__restrict__ probs, const int32_t *__restrict__ lab,
                                                      int *__restrict__ parent, int mask_label) {
    const CcProblem P = probs[blockIdx.y];
    const int W = P.W;
    const int n_hrows = (P.H - 1) / CT_H;          // horizontal seams: rows CT_H, 2*CT_H, ...
    const int n_vcols = (W - 1) / CT_W;            // vertical seams: columns CT_W, 2*CT_W, ...
    const long long n_h = (long long)n_hrows * W, n_v = (long long)n_vcols * P.H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_h + n_v; i += (long long)gridDim.x * blockDim.x) {
        int y, x, dy, dx;
        if (i < n_h) { y = (int)(i / W + 1) * CT_H; x = (int)(i % W); dy = 1; dx = 0; }
        else { const long long j = i - n_h; x = (int)(j / P.H + 1) * CT_W; y = (int)(j % P.H); dy = 0; dx = 1; }
        const long long g = P.pix_off + (long long)y * W + x;
        const int l = lab[g];
        if (l == mask_label) continue;
        const long long gn = g - (long long)dy * W - dx;
        if (lab[gn] != l) continue;
        // only the first pixel of a contact run issues the union (the previous pixel along the seam, if it carries the
        // same label on both sides, already links the same two components)
        const long long gp = dy ? g - 1 : g - W, gnp = dy ? gn - 1 : gn - W;
        // (only when that pixel lies in the same tile: its link to this pixel is then already inside the tile)
        const bool has_prev = dy ? (x % CT_W != 0) : (y % CT_H != 0);
        if (has_prev && lab[gp] == l && lab[gnp] == l) continue;
        unite(parent, (int)g, (int)gn);
    }
}

// After the seam unions only the LOCAL ROOTS need to be brought to their final root (round 4): a pixel's parent is its tile-local root
// (cc_tile_kernel), the unions re-point roots only, so once every local root points straight at the root of its component a pixel's
// root is parent[parent[pixel]] -- cc_root_of() -- and the pass over all pixels that used to flatten parent[] (4 bytes read and 4
// written per pixel) is gone.  One wave per 64 words of the local-root bitmap, a lane walks the set bits of its word (a few per
// tile): the component's pixel count moves to the root, the root bitmap of the ranking passes is the local-root map minus the
// roots that were linked away.
__global__ __launch_bounds__(64) void cc_roots_kernel(int *__restrict__ parent, int *__restrict__ size, long long n,
                                                      const unsigned long long *__restrict__ lrbits,
                                                      unsigned long long *__restrict__ rootbits) {
    const long long w = (long long)blockIdx.x * 64 + threadIdx.x, nw = (n + 63) >> 6;
    if (w >= nw) return;
    unsigned long long bits = lrbits[w], roots = 0ull;
    while (bits) {
        const int b = __builtin_ctzll(bits);
        bits &= bits - 1;
        const int lr = (int)((w << 6) + b);
        const int r = find_root(parent, lr);
        if (r == lr) { roots |= 1ull << b; continue; }
        parent[lr] = r;                       // (roots do not move any more: whoever reads this meanwhile still reaches r)
        atomicAdd(&size[r], size[lr]);        // size[] of a local root that is not the root is not read afterwards
    }
    rootbits[w] = roots;
}

// the root of a pixel's component (-1: masked pixel; other negative values: a pixel of a component that is being cut, cc_split_kernel)
__device__ __forceinline__ int cc_root_of(const int *__restrict__ parent, long long i) {
    const int p = parent[i];
    return p < 0 ? p : ld_agent(&parent[p]);
}

constexpr int SCAN_NT = 256, SCAN_PER = 16, SCAN_CHUNK = SCAN_NT * SCAN_PER;

__global__ __launch_bounds__(SCAN_NT) void cc_rank_blocksum_kernel(const CcProblem *__restrict__ probs, int nprob,
                                                                   const int *__restrict__ parent, const int *__restrict__ size,
                                                                   long long n, int *__restrict__ block_sums,
                                                                   int *__restrict__ counters /*[0]=n_small [1]=small_px*/,
                                                                   const unsigned long long *__restrict__ rootbits /* null: test parent[] */) {
    __shared__ int s_w[SCAN_NT / 64], s_ws[SCAN_NT / 64], s_wp[SCAN_NT / 64];
    const long long base = (long long)blockIdx.x * SCAN_CHUNK;
    int c = 0, nsmall = 0, spx = 0, nbig = 0;
    for (int j = 0; j < SCAN_PER; ++j) {
        const long long i = base + (long long)j * SCAN_NT + threadIdx.x;
        const bool root = i < n && (rootbits ? ((rootbits[i >> 6] >> (threadIdx.x & 63)) & 1ull) != 0ull : parent[i] == (int)i);
        if (root) {
            const int sz = size[i];
            const CcProblem &P = probs[find_prob(probs, nprob, i)];
            if (sz >= P.min_size) c += 1;
            else { nsmall += 1; spx += sz; }
            if (sz >= P.max_size && sz > 1) nbig += 1;   // reaches max_size: the reference cuts it (cc_split_kernel)
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        c += __shfl_xor(c, off); nsmall += __shfl_xor(nsmall, off); spx += __shfl_xor(spx, off); nbig += __shfl_xor(nbig, off);
    }
    if ((threadIdx.x & 63) == 0) {
        s_w[threadIdx.x >> 6] = c; s_ws[threadIdx.x >> 6] = nsmall; s_wp[threadIdx.x >> 6] = spx;
        if (nbig) atomicAdd(&counters[6], nbig);
    }
    __syncthreads();
    // three sums per block, scanned by cc_rank_scan_kernel: survivors | small components | pixels of small components.  (Small
    // components used to take their list slot and queue range with two global atomics EACH on two words: with the fragmented
    // label maps of a low compactness -- millions of small components -- that was 24 ms of a 95-ms step.)
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        block_sums[gridDim.x + blockIdx.x] = s_ws[0] + s_ws[1] + s_ws[2] + s_ws[3];
        block_sums[2 * gridDim.x + blockIdx.x] = s_wp[0] + s_wp[1] + s_wp[2] + s_wp[3];
    }
}

// ---- components that reach max_size -----------------------------------------------------------------------------------
// 0. (this rare path rewrites parents in place: it starts from a fully flattened map -- every thread writes its own entry only, and
// the entry it reads through, its local root's, already holds the root)
__global__ __launch_bounds__(256) void cc_flatten_all_kernel(int *__restrict__ parent, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int r = cc_root_of(parent, i);
        if (r >= 0) parent[i] = r;
    }
}

// 1. every root whose component reaches max_size gets an index b (big_root[b], queue offset); newlab[root] = b
__global__ __launch_bounds__(256) void cc_big_list_kernel(const CcProblem *__restrict__ probs, int nprob, const int *__restrict__ parent,
                                                          const int *__restrict__ size, long long n, int *__restrict__ newlab,
                                                          int *__restrict__ big_root, int *__restrict__ big_qoff,
                                                          int *__restrict__ big_box, int *__restrict__ counters /*[6] list cursor, [7] queue cursor*/) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || parent[i] != (int)i) return;
    const int sz = size[i];
    if (!(sz >= probs[find_prob(probs, nprob, i)].max_size && sz > 1)) return;
    const int b = atomicAdd(&counters[6], 1);
    big_root[b] = (int)i;
    big_qoff[b] = atomicAdd(&counters[7], sz);
    big_box[4 * b] = 0x7fffffff; big_box[4 * b + 1] = -1; big_box[4 * b + 2] = 0x7fffffff; big_box[4 * b + 3] = -1;
    newlab[i] = b;
}

// 2. the pixels of those components are marked parent = -2 - root ("in the region, not yet in a piece") and their bounding
// boxes collected
__global__ __launch_bounds__(256) void cc_big_mark_kernel(const CcProblem *__restrict__ probs, int nprob, int *__restrict__ parent,
                                                          const int *__restrict__ size, long long n, const int *__restrict__ newlab,
                                                          int *__restrict__ big_box) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = parent[i];   // (flat here: cc_flatten_all_kernel ran first -- this kernel overwrites parents that cc_root_of would read)
    if (r < 0) return;
    const int sz = size[r];
    const CcProblem &P = probs[find_prob(probs, nprob, r)];
    if (!(sz >= P.max_size && sz > 1)) return;
    const int b = newlab[r];
    const int y = (int)((i - P.pix_off) / P.W), x = (int)((i - P.pix_off) - (long long)y * P.W);
    atomicMin(&big_box[4 * b], y); atomicMax(&big_box[4 * b + 1], y);
    atomicMin(&big_box[4 * b + 2], x); atomicMax(&big_box[4 * b + 3], x);
    parent[i] = -2 - r;   // (-1 is the parent of masked pixels)
}

// 3. one lane per component replays the reference: raster scan of the bounding box, capped BFS from every pixel that is
// still unassigned.  parent of a piece = its start pixel (its smallest index: the scan picks the smallest unassigned
// pixel, and a BFS only takes unassigned ones), size[start] = pixels of the piece.
__global__ __launch_bounds__(64) void cc_split_kernel(const CcProblem *__restrict__ probs, int nprob, int *__restrict__ parent,
                                                      int *__restrict__ size, const int *__restrict__ big_root,
                                                      const int *__restrict__ big_qoff, const int *__restrict__ big_box, int n_big,
                                                      int *__restrict__ queue) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_big) return;
    const int R = big_root[b], region = -2 - R;
    const CcProblem P = probs[find_prob(probs, nprob, R)];
    const int H = P.H, W = P.W, base = (int)P.pix_off, cap = P.max_size;
    int *q = queue + big_qoff[b];
    const int y0 = big_box[4 * b], y1 = big_box[4 * b + 1], x0 = big_box[4 * b + 2], x1 = big_box[4 * b + 3];
    for (int sy = y0; sy <= y1; ++sy)
        for (int sx = x0; sx <= x1; ++sx) {
            const int start = base + sy * W + sx;
            if (parent[start] != region) continue;
            int cnt = 1, visited = 0;
            q[0] = start;
            parent[start] = start;
            while (visited < cnt && cnt < cap) {
                const int p = q[visited];
                const int y = (p - base) / W, x = (p - base) - y * W;
                for (int d = 0; d < 4; ++d) {
                    const int xx = x + (d == 0 ? 1 : (d == 1 ? -1 : 0));
                    const int yy = y + (d == 2 ? 1 : (d == 3 ? -1 : 0));
                    if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                    const int nb = base + yy * W + xx;
                    if (parent[nb] == region) {
                        parent[nb] = start;
                        q[cnt++] = nb;
                        if (cnt >= cap) break;
                    }
                }
                ++visited;
            }
            size[start] = cnt;
        }
}

// The same two passes on the root bitmap of cc_flatten_kernel (the usual path: no component was cut): ONE wave per chunk of
// SCAN_CHUNK = 64 x 64 pixels, lane l owns bitmap word l and walks its set bits (a root every few hundred pixels: a lane holds
// none, one or two) -- no barriers, no pass over parent[] / size[] of every pixel; sums and prefixes over the wave by shuffles.
static_assert(SCAN_CHUNK == 64 * 64, "one wave per chunk: 64 bitmap words");
__global__ __launch_bounds__(64) void cc_rank_blocksum_bits_kernel(const CcProblem *__restrict__ probs, int nprob,
                                                                    const int *__restrict__ size, long long n,
                                                                    int *__restrict__ block_sums, int *__restrict__ counters,
                                                                    const unsigned long long *__restrict__ rootbits) {
    const int lane = threadIdx.x;
    const long long w = (long long)blockIdx.x * 64 + lane, nw = (n + 63) >> 6;
    unsigned long long bits = w < nw ? rootbits[w] : 0ull;
    int c = 0, nsmall = 0, spx = 0, nbig = 0;
    while (bits) {
        const int b = __builtin_ctzll(bits);
        bits &= bits - 1;
        const long long i = (w << 6) + b;
        const int sz = size[i];
        const CcProblem &P = probs[find_prob(probs, nprob, i)];
        if (sz >= P.min_size) c += 1;
        else { nsmall += 1; spx += sz; }
        if (sz >= P.max_size && sz > 1) nbig += 1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c += __shfl_xor(c, off); nsmall += __shfl_xor(nsmall, off); spx += __shfl_xor(spx, off); nbig += __shfl_xor(nbig, off);
    }
    if (lane == 0) {
        block_sums[blockIdx.x] = c;
        block_sums[gridDim.x + blockIdx.x] = nsmall;
        block_sums[2 * gridDim.x + blockIdx.x] = spx;
        if (nbig) atomicAdd(&counters[6], nbig);
    }
}

__global__ __launch_bounds__(64) void cc_rank_apply_bits_kernel(const CcProblem *__restrict__ probs, int nprob,
                                                                 const int *__restrict__ size, long long n,
                                                                 const int *__restrict__ block_sums, int *__restrict__ newlab,
                                                                 int *__restrict__ small_list, int *__restrict__ small_qoff,
                                                                 const unsigned long long *__restrict__ rootbits) {
    const int lane = threadIdx.x;
    const long long w = (long long)blockIdx.x * 64 + lane, nw = (n + 63) >> 6;
    const unsigned long long word = w < nw ? rootbits[w] : 0ull;
    // first walk: this lane's counts; exclusive prefixes over the lanes; second walk: the labels / slots in raster order
    int c = 0, ns = 0, px = 0;
    for (unsigned long long bits = word; bits; bits &= bits - 1) {
        const long long i = (w << 6) + __builtin_ctzll(bits);
        const int sz = size[i];
        if (sz >= probs[find_prob(probs, nprob, i)].min_size) c += 1;
        else { ns += 1; px += sz; }
    }
    int pc = c, pn = ns, pp = px;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int tc = __shfl_up(pc, off), tn = __shfl_up(pn, off), tp = __shfl_up(pp, off);
        if (lane >= off) { pc += tc; pn += tn; pp += tp; }
    }
    int run = block_sums[blockIdx.x] + pc - c, srun = block_sums[gridDim.x + blockIdx.x] + pn - ns,
        prun = block_sums[2 * gridDim.x + blockIdx.x] + pp - px;
    for (unsigned long long bits = word; bits; bits &= bits - 1) {
        const long long i = (w << 6) + __builtin_ctzll(bits);
        const int sz = size[i];
        if (sz >= probs[find_prob(probs, nprob, i)].min_size) {
            newlab[i] = run++;
        } else {
            small_list[srun] = (int)i;
            small_qoff[srun] = prun;
            newlab[i] = -(srun + 2);
            srun += 1; prun += sz;
        }
    }
}

// exclusive scan of block_sums in place, single workgroup; total -> counters[2]
__global__ __launch_bounds__(1024) void cc_rank_scan_kernel(int *__restrict__ block_sums_all, int nb, int *__restrict__ counters) {
    __shared__ int s_part[1024];
    const int tid = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, nb);
    // segment 0: survivors -> counters[2]; 1: small components -> counters[0]; 2: their pixels -> counters[1]
    // (one workgroup per segment: grid = 3)
    {
        const int seg = blockIdx.x;
        int *block_sums = block_sums_all + (size_t)seg * nb;
        int s = 0;
        for (int i = lo; i < hi; ++i) s += block_sums[i];
        s_part[tid] = s;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int v = (tid >= off) ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += v;
            __syncthreads();
        }
        int run = s_part[tid] - s;
        for (int i = lo; i < hi; ++i) { const int v = block_sums[i]; block_sums[i] = run; run += v; }
        if (tid == 1023) counters[seg == 0 ? 2 : seg - 1] = s_part[1023];
        __syncthreads();
    }
}

// newlab[root] = rank (>= 0) for survivors, -(index+2) for small components (collected in small_list)
__global__ __launch_bounds__(SCAN_NT) void cc_rank_apply_kernel(const CcProblem *__restrict__ probs, int nprob,
                                                                const int *__restrict__ parent, const int *__restrict__ size,
                                                                long long n, const int *__restrict__ block_sums,
                                                                int *__restrict__ newlab, int *__restrict__ small_list,
                                                                int *__restrict__ small_qoff, int *__restrict__ counters,
                                                                const unsigned long long *__restrict__ rootbits /* null: test parent[] */) {
    (void)counters;
    __shared__ int s_w[SCAN_NT / 64], s_ws[SCAN_NT / 64], s_wp[SCAN_NT / 64];
    __shared__ int s_run, s_srun, s_prun;   // next survivor label, next slot of the small list, next queue offset
    const long long base = (long long)blockIdx.x * SCAN_CHUNK;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        s_run = block_sums[blockIdx.x];
        s_srun = block_sums[gridDim.x + blockIdx.x];
        s_prun = block_sums[2 * gridDim.x + blockIdx.x];
    }
    __syncthreads();
    for (int j = 0; j < SCAN_PER; ++j) {
        const long long i = base + (long long)j * SCAN_NT + threadIdx.x;
        int flag = 0;
        bool small = false;
        int sz = 0;
        if (i < n && (rootbits ? ((rootbits[i >> 6] >> lane) & 1ull) != 0ull : parent[i] == (int)i)) {
            sz = size[i];
            flag = sz >= probs[find_prob(probs, nprob, i)].min_size;
            small = !flag;
        }
        const unsigned long long bal = __ballot(flag), sbal = __ballot(small);
        // pixels of the small components in front of this lane inside the wave (inclusive scan by shuffles, then exclusive)
        int pin = small ? sz : 0;
        if (sbal) {   // wave-uniform
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(pin, off); if (lane >= off) pin += t; }
        }
        if (lane == 63) s_wp[wv] = pin;
        if (lane == 0) { s_w[wv] = __popcll(bal); s_ws[wv] = __popcll(sbal); }
        __syncthreads();
        int before = 0, sbefore = 0, pbefore = 0;
        for (int w = 0; w < wv; ++w) { before += s_w[w]; sbefore += s_ws[w]; pbefore += s_wp[w]; }
        const int total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        const int stotal = s_ws[0] + s_ws[1] + s_ws[2] + s_ws[3], ptotal = s_wp[0] + s_wp[1] + s_wp[2] + s_wp[3];
        if (flag) newlab[i] = s_run + before + __popcll(bal & ((1ull << lane) - 1ull));
        if (small) {   // slots and queue ranges in raster order of the roots: no atomics
            const int idx = s_srun + sbefore + __popcll(sbal & ((1ull << lane) - 1ull));
            small_list[idx] = (int)i;
            small_qoff[idx] = s_prun + pbefore + pin - sz;
            newlab[i] = -(idx + 2);
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_run += total; s_srun += stotal; s_prun += ptotal; }
        __syncthreads();
    }
}

// code[p]: which component pixel p belongs to, in the form the replay needs -- -1 masked, r >= 0 for a pixel of the SURVIVING
// component rooted at r (r is also the time at which that component is labelled), -(s + 2) for a pixel of small component s.  One
// dense pass; the replay then reads ONE word per neighbour where it used to chase parent[] -> newlab[] (two dependent gathers).
__global__ __launch_bounds__(256) void cc_code_kernel(const int *__restrict__ parent, const int *__restrict__ newlab, long long n,
                                                      int *__restrict__ code) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int r = cc_root_of(parent, i);
        int c = -1;
        if (r >= 0) { const int nl = newlab[r]; c = nl >= 0 ? r : nl; }
        code[i] = c;
    }
}

// BFS of the reference over small component s (code mine = -(s + 2)), started at `start`; neighbour order
// (x+1, x-1, y+1, y-1).  A neighbour pixel of another component counts as "already labelled" when that
// component was labelled before this BFS started: a surviving component as soon as the scan reached its
// first pixel (root < start); a small component once one of its own BFS attempts found a labelled
// neighbour (settle < start; a small component that finds none is written back as 0 == unset when
// start_label is 1).  Returns the last labelled neighbour met; *min_since = the earliest time at which any
// neighbouring component is labelled (INT_MAX: never / no neighbour).
// The kernel is bound by the latency of its gathers (1 555 load instructions per wave, 10 % of the wave cycles spent on anything
// else: profiles/r03_notes.md), so a popped pixel issues the four code words of its neighbours TOGETHER, then -- together again --
// what each of them needs next (the visited mark of an own pixel, the settle time of a small neighbour): two round trips per
// pixel where the branchy form took up to twelve.  NBR (nullable): the small neighbours met, for the caller's enqueue.
// CODE = false: no code[] array (few small components: the dense pass would cost more than it saves); the word is derived from
// parent[] and newlab[] -- two dependent gathers, still issued for the four neighbours together.
struct NbrList { int *buf; int n; int cap; bool overflow; };
template <bool CODE>
__device__ int replay_bfs(const int *__restrict__ code, const int *__restrict__ newlab, const int *settle, int mine, int start, int H, int W, int base,
                          int *__restrict__ q, int32_t *__restrict__ out, int *n_out, int *min_since, NbrList *nbrs) {
    int head = 0, tail = 1, adjacent = -1, ms = 0x7fffffff;
    q[0] = start;
    out[start] = mine;
    while (head < tail) {
        const int p = q[head++];
        const int y = (p - base) / W, x = (p - base) - y * W;
        int nb[4], c[4], v2[4];
        bool ok[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int xx = x + (d == 0 ? 1 : (d == 1 ? -1 : 0));
            const int yy = y + (d == 2 ? 1 : (d == 3 ? -1 : 0));
            ok[d] = !(xx < 0 || xx >= W || yy < 0 || yy >= H);
            nb[d] = ok[d] ? base + yy * W + xx : p;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) c[d] = code[nb[d]];                 // four loads in flight (CODE = false: `code` is parent[])
        if (!CODE) {
#pragma unroll
            for (int d = 0; d < 4; ++d) c[d] = c[d] >= 0 ? code[c[d]] : c[d];   // local root -> root (cc_root_of), four in flight again
            int nl[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) nl[d] = newlab[c[d] >= 0 ? c[d] : 0];
#pragma unroll
            for (int d = 0; d < 4; ++d) c[d] = c[d] < 0 ? -1 : (nl[d] >= 0 ? c[d] : nl[d]);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {                                   // ... and the four that depend on them
            const bool own = c[d] == mine;
            const int *src = own ? (const int *)out + nb[d] : settle + ((!own && c[d] <= -2) ? -c[d] - 2 : 0);
            v2[d] = (ok[d] && (own || c[d] <= -2)) ? *src : 0;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {                                   // the reference's order
            if (!ok[d]) continue;
            if (c[d] == mine) {
                if (v2[d] != mine) { out[nb[d]] = mine; q[tail++] = nb[d]; }
            } else if (c[d] != -1) {
                const int since = c[d] >= 0 ? c[d] : v2[d];
                ms = since < ms ? since : ms;
                if (since < start) adjacent = nb[d];                    // the LAST labelled neighbour met wins
                if (nbrs && c[d] <= -2) {                               // a small neighbour: remembered for the enqueue
                    const int t = -c[d] - 2;
                    if (nbrs->n == 0 || nbrs->buf[nbrs->n - 1] != t) {
                        if (nbrs->n < nbrs->cap) nbrs->buf[nbrs->n++] = t; else nbrs->overflow = true;
                    }
                }
            }
        }
    }
    *n_out = tail;
    *min_since = ms;
    return adjacent;
}

// One lane replays the reference's treatment of one small component: BFS from its first pixel; if no
// labelled neighbour is found and start_label is 1 the component is written back as 0 (== unset), the
// raster scan meets it again at its next pixel and the BFS is replayed from there -- and so on, pixel by pixel in raster order,
// until an attempt meets a labelled neighbour.  An attempt at pixel p succeeds exactly when some neighbouring component is
// labelled before p, i.e. when p > m = the earliest labelling time among the neighbours (the first BFS has met them all): the
// successful attempt is the one at the first pixel of the component after m -- found by one pass over the component's pixels,
// then replayed once for its `adjacent` (round 2 replayed every failed attempt and sorted the pixels by insertion: quadratic in
// the component's size, which is unbounded through the stage-level entry point).
// settle_out[s] = start pixel of the attempt that found a neighbour (INT_MAX if none), target[s] = that neighbour pixel.
// `out` doubles as the visited map (rewritten by the final relabel pass).
// The settle times are the fixed point of a monotone map (a neighbour that settles later can only make this component
// settle later), iterated IN PLACE from one side on a work list -- every small component in round 0,
// afterwards the small neighbours of the components whose settle time moved in the round before (a component meets all its
// neighbours in its own BFS, adjacency is symmetric, so the one that moves enqueues those that depend on it; `tag` keeps a
// component from being enqueued twice in a round).  Reads of a neighbour's time may be stale inside a round: every value read
// lies between the start and the fixed point, a component whose neighbour moved is evaluated again in the next round on values
// at least as new as the end of this one, and the rounds end when nothing moved -- the same fixed point as synchronous (Jacobi)
// rounds over all components, at the cost of the frontier instead of the whole list per round.
constexpr int BFS_PUSH = 12;   // list entries a lane collects before it falls back to one atomic per entry
template <bool CODE>
__device__ int small_component_eval(const CcProblem *__restrict__ probs, int nprob,
                                    const int *__restrict__ code, const int *__restrict__ newlab,
                                    const int *__restrict__ small_list, const int *__restrict__ small_qoff,
                                    int start_label, int *__restrict__ settle,
                                    int *__restrict__ queue, int32_t *__restrict__ out,
                                    int *__restrict__ target, const int *__restrict__ work_in, int i,
                                    int *__restrict__ work_out, int *__restrict__ work_cnt,
                                    int *__restrict__ tag, int round, int (*s_push)[BFS_PUSH + 1]) {
    const int s = work_in ? work_in[i] : i;
    const int r = small_list[s];
    const CcProblem P = probs[find_prob(probs, nprob, r)];
    const int H = P.H, W = P.W, base = (int)P.pix_off;
    const int mine = -(s + 2);
    int *q = queue + small_qoff[s];
    int csize = 0, m = 0x7fffffff;
    int start = r;
    NbrList nbrs{&s_push[threadIdx.x][0], 0, BFS_PUSH, false};   // the small neighbours met by the first walk (it meets them all)
    int adjacent = replay_bfs<CODE>(code, newlab, settle, mine, r, H, W, base, q, out, &csize, &m, &nbrs);
    if (adjacent < 0 && start_label == 1) {
        start = 0x7fffffff;
        if (m != 0x7fffffff) {
            int st = 0x7fffffff;                         // the first pixel of the component after m, in raster order
            for (int i = 0; i < csize; ++i) { const int p = q[i]; if (p > m && p < st) st = p; }
            if (st != 0x7fffffff) {
                for (int i = 0; i < csize; ++i) out[q[i]] = 0;   // clear the visited marks of the first attempt
                int n2 = 0, m2 = 0;
                adjacent = replay_bfs<CODE>(code, newlab, settle, mine, st, H, W, base, q, out, &n2, &m2, nullptr);
                // (a neighbour's time may have moved between the two walks -- another lane of this round: whatever is read lies
                // between the start and the fixed point, and this component is evaluated again when a neighbour moved)
                start = adjacent >= 0 ? st : 0x7fffffff;
            }
        }
    }
    for (int i = 0; i < csize; ++i) out[q[i]] = 0;
    target[s] = adjacent;
    // moved: the small components around this one are evaluated again in the next round.  Their list indices are collected per
    // lane in LDS and appended with ONE atomic per wave: every enqueue used to add 1 to the same global word, and atomics on one
    // word run at a few nanoseconds EACH, device-wide.  The first walk has met every neighbour: its list (consecutive repeats
    // dropped, `tag` drops the rest) is the enqueue -- unless it overflowed, then the component's pixels are walked once more.
    if (settle[s] == start) return 0;
    settle[s] = start;
    int npush = 0;
    if (!nbrs.overflow) {
        for (int j = 0; j < nbrs.n; ++j) {
            const int t = s_push[threadIdx.x][j];
            if (atomicExch(&tag[t], round + 1) != round + 1) s_push[threadIdx.x][npush++] = t;   // (npush <= j: in place)
        }
        return npush;
    }
    for (int i = 0; i < csize; ++i) {
        const int p = q[i];
        const int y = (p - base) / W, x = (p - base) - y * W;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int xx = x + (d == 0 ? 1 : (d == 1 ? -1 : 0));
            const int yy = y + (d == 2 ? 1 : (d == 3 ? -1 : 0));
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            int c = CODE ? code[base + yy * W + xx] : cc_root_of(code, base + yy * W + xx);
            if (!CODE) c = c < 0 ? -1 : (newlab[c] >= 0 ? c : newlab[c]);
            if (c > -2 || c == mine) continue;
            const int t = -c - 2;
            if (atomicExch(&tag[t], round + 1) != round + 1) {
                if (npush < BFS_PUSH) s_push[threadIdx.x][npush++] = t;
                else work_out[atomicAdd(work_cnt, 1)] = t;          // (more than BFS_PUSH new neighbours)
            }
        }
    }
    return npush;
}

// (the wave meets again here: one reservation for all its lanes)
template <bool CODE>
__global__ __launch_bounds__(64) void cc_small_bfs_kernel(const CcProblem *__restrict__ probs, int nprob,
                                                          const int *__restrict__ code, const int *__restrict__ newlab,
                                                          const int *__restrict__ small_list, const int *__restrict__ small_qoff,
                                                          int start_label, int *__restrict__ settle,
                                                          int *__restrict__ queue, int32_t *__restrict__ out,
                                                          int *__restrict__ target, const int *__restrict__ work_in, int n_items,
                                                          int *__restrict__ work_out, int *__restrict__ work_cnt,
                                                          int *__restrict__ tag, int round) {
    __shared__ int s_push[64][BFS_PUSH + 1];   // (+1: the lanes' rows start in different banks)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int npush = 0;
    if (i < n_items)
        npush = small_component_eval<CODE>(probs, nprob, code, newlab, small_list, small_qoff, start_label, settle, queue, out, target,
                                     work_in, i, work_out, work_cnt, tag, round, s_push);
    // inclusive prefix of the lanes' counts, one atomic for the wave
    int inc = npush;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if ((int)threadIdx.x >= off) inc += v; }
    const int total = __shfl(inc, 63);
    if (total == 0) return;
    int base_slot = 0;
    if (threadIdx.x == 63) base_slot = atomicAdd(work_cnt, total);
    base_slot = __shfl(base_slot, 63);
    for (int j = 0; j < npush; ++j) work_out[base_slot + inc - npush + j] = s_push[threadIdx.x][j];
}

// ---- settle rounds on the pixels of the small components (round 4) ---------------------------------------------------------
// Where small components are many (a noisy 3-band raster: one per 33 pixels, 6.5 M per step) the work-list rounds above spend their
// time walking components one lane each -- up to two serial BFS walks per component and evaluation, with a queue and a visited map,
// bound by the latency of their gathers.  But a settle time needs no walk:
//     t(S) = min { p in S : p > m(S) },   m(S) = min over the neighbours N of S of t(N)
// (t(N) = root of N for a surviving N; the formula covers the first attempt too: m(S) < first pixel of S  =>  t(S) = first pixel).
// Both minima are order-free.  One dense pass lists the pixels of the small components (a tenth of the pixels in that regime),
// grouped by component -- a component's pixels go to its slice of the BFS queue's layout (small_qoff) through a per-component
// cursor -- with what never changes: the pixel and the codes of its four neighbours (-1: masked, outside, or the same component).
// An evaluation is then two short loops over the component's slice (no queue, no visited map, nothing written but the time), and
// the work list carries on as before: round 0 evaluates every small component, a later round the small neighbours of those whose
// time moved.  Only the LAST step is order-dependent -- the reference keeps the last labelled neighbour its BFS meets -- and runs
// once per component, from its final settle time (cc_small_target_kernel): one walk instead of one or two per evaluation.
// (Tried on the way, `bench.py --bands 3`, connectivity per step: work list with walks 31.9 ms of small-component kernels; synchronous
// rounds as two passes over ALL pixels 42 ms; over the list 17 ms + a list build that appended through ONE cursor, 48 ms:
// profiles/r04_notes.md.)
constexpr int T_NEVER = 0x7fffffff;
// grid = (rows, problems)
__global__ __launch_bounds__(256) void cc_small_pixels_kernel(const CcProblem *__restrict__ probs, const int *__restrict__ code,
                                                              const int *__restrict__ small_qoff, int *__restrict__ ccur,
                                                              int *__restrict__ px_g, int4 *__restrict__ px_cn, int cap) {
    const CcProblem P = probs[blockIdx.y];
    for (int y = blockIdx.x; y < P.H; y += gridDim.x) {
        const long long row = P.pix_off + (long long)y * P.W;
        for (int x = threadIdx.x; x < P.W; x += 256) {
            const int c = code[row + x];
            if (c > -2) continue;
            int4 cn;
            cn.x = x + 1 < P.W ? code[row + x + 1] : -1;
            cn.y = x > 0 ? code[row + x - 1] : -1;
            cn.z = y + 1 < P.H ? code[row + x + P.W] : -1;
            cn.w = y > 0 ? code[row + x - P.W] : -1;
            if (cn.x == c) cn.x = -1;
            if (cn.y == c) cn.y = -1;
            if (cn.z == c) cn.z = -1;
            if (cn.w == c) cn.w = -1;
            const int s = -c - 2;
            const int idx = small_qoff[s] + atomicAdd(&ccur[s], 1);     // (ccur[s] ends as the component's pixel count)
            if (idx < cap) { px_g[idx] = (int)(row + x); px_cn[idx] = cn; }
        }
    }
}
// one lane evaluates one small component; moved: its small neighbours join the next round's list (as in cc_small_bfs_kernel).
// Components are 3.7 pixels on average with a tail up to min_size: among the 64 of a wave there is nearly always one of dozens of
// pixels, and the wave would wait for that lane -- components above SETTLE_COOP pixels are left out of the per-lane part and
// evaluated by the whole wave afterwards, 64 entries per step.
constexpr int SETTLE_COOP = 12;
__device__ __forceinline__ int settle_entry_min(const int4 cn, const int *__restrict__ settle) {
    if ((cn.x & cn.y & cn.z & cn.w) == -1) return T_NEVER;           // no foreign neighbour
    const int a = cn.x >= 0 ? cn.x : (cn.x == -1 ? T_NEVER : ld_agent(&settle[-cn.x - 2]));
    const int b = cn.y >= 0 ? cn.y : (cn.y == -1 ? T_NEVER : ld_agent(&settle[-cn.y - 2]));
    const int c = cn.z >= 0 ? cn.z : (cn.z == -1 ? T_NEVER : ld_agent(&settle[-cn.z - 2]));
    const int d = cn.w >= 0 ? cn.w : (cn.w == -1 ? T_NEVER : ld_agent(&settle[-cn.w - 2]));
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    return v;
}
__global__ __launch_bounds__(64) void cc_settle_eval_kernel(const int *__restrict__ small_qoff, const int *__restrict__ ccur,
                                                            const int *__restrict__ px_g, const int4 *__restrict__ px_cn, int cap,
                                                            int *__restrict__ settle, const int *__restrict__ work_in, int n_items,
                                                            int *__restrict__ work_out, int *__restrict__ work_cnt,
                                                            int *__restrict__ tag, int round) {
    __shared__ int s_push[64][BFS_PUSH + 1];
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x;
    int npush = 0;
    auto push_neighbours = [&](const int4 cn) {
        const int nb[4] = {cn.x, cn.y, cn.z, cn.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (nb[d] > -2) continue;
            const int tt = -nb[d] - 2;
            if (atomicExch(&tag[tt], round + 1) != round + 1) {
                if (npush < BFS_PUSH) s_push[lane][npush++] = tt;
                else work_out[atomicAdd(work_cnt, 1)] = tt;          // (more than BFS_PUSH new neighbours)
            }
        }
    };
    int s = 0, o = 0, n = 0;
    if (i < n_items) {
        s = work_in ? work_in[i] : i;
        o = small_qoff[s];
        n = ccur[s];
        if (o + n > cap) n = cap - o;
    }
    if (n > 0 && n <= SETTLE_COOP) {
        int m = T_NEVER;
        for (int e = 0; e < n; ++e) m = min(m, settle_entry_min(px_cn[o + e], settle));
        int t = T_NEVER;
        if (m != T_NEVER)
            for (int e = 0; e < n; ++e) { const int g = px_g[o + e]; if (g > m && g < t) t = g; }
        if (settle[s] != t) {
            settle[s] = t;
            for (int e = 0; e < n; ++e) push_neighbours(px_cn[o + e]);
        }
    }
    unsigned long long coop = __ballot(n > SETTLE_COOP);
    while (coop) {                                                   // (wave-uniform)
        const int src = __builtin_ctzll(coop);
        coop &= coop - 1;
        const int sb = __shfl(s, src), ob = __shfl(o, src), nb = __shfl(n, src);
        int m = T_NEVER;
        for (int e = lane; e < nb; e += 64) m = min(m, settle_entry_min(px_cn[ob + e], settle));
        m = wave_min_i32(m);
        int t = T_NEVER;
        if (m != T_NEVER)
            for (int e = lane; e < nb; e += 64) { const int g = px_g[ob + e]; if (g > m && g < t) t = g; }
        t = wave_min_i32(t);
        const int old = settle[sb];
        if (old != t) {
            if (lane == 0) settle[sb] = t;
            for (int e = lane; e < nb; e += 64) push_neighbours(px_cn[ob + e]);
        }
    }
    // inclusive prefix of the lanes' counts, one atomic for the wave
    int inc = npush;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if ((int)threadIdx.x >= off) inc += v; }
    const int total = __shfl(inc, 63);
    if (total == 0) return;
    int base_slot = 0;
    if (threadIdx.x == 63) base_slot = atomicAdd(work_cnt, total);
    base_slot = __shfl(base_slot, 63);
    for (int j = 0; j < npush; ++j) work_out[base_slot + inc - npush + j] = s_push[threadIdx.x][j];
}
// the one order-dependent step: the BFS of the reference from the component's settle time, for the last labelled neighbour it meets
__global__ __launch_bounds__(64) void cc_small_target_kernel(const CcProblem *__restrict__ probs, int nprob, const int *__restrict__ code,
                                                             const int *__restrict__ newlab, const int *__restrict__ small_list,
                                                             const int *__restrict__ small_qoff, const int *__restrict__ settle,
                                                             int *__restrict__ queue, int32_t *__restrict__ out, int *__restrict__ target, int n_small) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_small) return;
    const int st = settle[s];
    if (st == T_NEVER) { target[s] = -1; return; }
    const CcProblem P = probs[find_prob(probs, nprob, small_list[s])];
    int *q = queue + small_qoff[s];
    int csize = 0, ms = 0;
    const int adjacent = replay_bfs<true>(code, newlab, settle, -(s + 2), st, P.H, P.W, (int)P.pix_off, q, out, &csize, &ms, nullptr);
    for (int i = 0; i < csize; ++i) out[q[i]] = 0;
    target[s] = adjacent;
}

// from_above = 0: the optimistic start (every small component labelled at its first pixel); 1: the pessimistic one (never)
__global__ void cc_settle_init_kernel(const int *__restrict__ small_list, int n_small, int *__restrict__ settle, int from_above) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_small) settle[s] = from_above ? 0x7fffffff : small_list[s];
}

// the label a small component ends with: it follows its adjacency chain (a chain only leads to components that settled EARLIER: it
// is acyclic and at most n_small long) to a surviving component's rank, or to nothing (-1)
__global__ void cc_small_final_kernel(const int *__restrict__ parent, const int *__restrict__ newlab, const int *__restrict__ target,
                                      int n_small, int max_hops, int *__restrict__ small_final) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_small) return;
    int nl = -(s + 2), hops = 0;
    while (nl < 0) {
        const int t = target[-nl - 2];
        if (t < 0 || ++hops > max_hops) { nl = -1; break; }
        nl = newlab[parent[parent[t]]];
    }
    small_final[s] = nl;
}

// final labels: survivors get rank + start_label; small components the label their adjacency chain ends in
__global__ __launch_bounds__(256) void cc_relabel_kernel(const CcResolve R, long long n, int32_t *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = cc_resolve_label(R, i);
}

int enforce_connectivity_batch(obia_ctx *ctx, const std::vector<CcProblem> &probs, const int32_t *labels_in,
                               long long total_pix, int start_label, int32_t *labels_out, int *h_n_labels_out,
                               CcResolve *deferred) {
    ScopedSpan span(ctx, T_CC);
    Arena &A = ctx->arena;
    const long long n = total_pix;
    const int np = (int)probs.size();
    if (np <= 0 || n <= 0 || n > 0x7fffffffLL) { set_error("label batch of %lld pixels not supported", n); return OBIA_E_INVALID; }
    const int mask_label = start_label - 1;
    CcProblem *d_probs = A.get<CcProblem>(np);
    int *parent = A.get<int>(n), *size = A.get<int>(n), *newlab = A.get<int>(n);
    const int nb = cdiv(n, SCAN_CHUNK);
    int *block_sums = A.get<int>(3 * (size_t)nb);   // survivors | small components | their pixels (cc_rank_blocksum_kernel)
    unsigned long long *rootbits = A.get<unsigned long long>((size_t)((n + 63) / 64));   // one bit per pixel: root (cc_flatten_kernel)
    if (!rootbits) return OBIA_E_NOMEM;
    // one bit per pixel: tile-local root (cc_tile_kernel); the eight counters sit in front of the map so that ONE fill clears both
    unsigned long long *lrbits_block = A.get<unsigned long long>((size_t)((n + 63) / 64) + 4);
    if (!lrbits_block) return OBIA_E_NOMEM;
    unsigned long long *lrbits = lrbits_block + 4;
    const unsigned long long *rb = rootbits;   // the map the ranking passes read; null once the size cut has changed the roots
    int *counters = reinterpret_cast<int *>(lrbits_block);
    if (!d_probs || !parent || !size || !newlab || !block_sums) return OBIA_E_NOMEM;
    OBIA_TRY(upload_async(ctx, d_probs, probs.data(), sizeof(CcProblem) * np));
    OBIA_HIP_TRY(hipMemsetAsync(lrbits_block, 0, sizeof(unsigned long long) * ((size_t)((n + 63) / 64) + 4), ctx->stream));
    int gs = cdiv(n, 256 * 4);
    if (gs > 65535 * 4) gs = 65535 * 4;
    {
        int max_tiles = 1;
        long long max_seam = 1;
        for (auto &P : probs) {
            const int t = cdiv(P.W, CT_W) * cdiv(P.H, CT_H);
            if (t > max_tiles) max_tiles = t;
            const long long sm = (long long)((P.H - 1) / CT_H) * P.W + (long long)((P.W - 1) / CT_W) * P.H;
            if (sm > max_seam) max_seam = sm;
        }
        hipLaunchKernelGGL(cc_tile_kernel, dim3(max_tiles, np), dim3(256), 0, ctx->stream, d_probs, labels_in, parent, size, mask_label, lrbits);
        int sg = cdiv(max_seam, 256);
        if (sg > 65535) sg = 65535;
        hipLaunchKernelGGL(cc_seam_kernel, dim3(sg, np), dim3(256), 0, ctx->stream, d_probs, labels_in, parent, mask_label);
    }
    debug_sync(ctx, "cc: tile + seam");
    hipLaunchKernelGGL(cc_roots_kernel, dim3(cdiv((n + 63) / 64, 64)), dim3(64), 0, ctx->stream, parent, size, n, lrbits, rootbits);
    debug_sync(ctx, "cc: roots");
    hipLaunchKernelGGL(cc_rank_blocksum_bits_kernel, dim3(nb), dim3(64), 0, ctx->stream, d_probs, np, size, n, block_sums, counters, rootbits);
    hipLaunchKernelGGL(cc_rank_scan_kernel, dim3(3), dim3(1024), 0, ctx->stream, block_sums, nb, counters);
    int hc[8];
    OBIA_TRY(read_back(ctx, hc, counters, sizeof(hc)));   // also orders the pageable `probs` upload
    if (hc[6] > 0) {
        // some component reaches max_size: cut it the way the reference's capped BFS does, then count again
        const int n_big = hc[6];
        int *big_root = A.get<int>(n_big), *big_qoff = A.get<int>(n_big), *big_box = A.get<int>(4 * (size_t)n_big);
        if (!big_root || !big_qoff || !big_box) return OBIA_E_NOMEM;
        OBIA_HIP_TRY(hipMemsetAsync(counters, 0, sizeof(int) * 8, ctx->stream));
        hipLaunchKernelGGL(cc_flatten_all_kernel, dim3(gs), dim3(256), 0, ctx->stream, parent, n);
        hipLaunchKernelGGL(cc_big_list_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, d_probs, np, parent, size, n, newlab,
                           big_root, big_qoff, big_box, counters);
        int hb[8];
        OBIA_TRY(read_back(ctx, hb, counters, sizeof(hb)));
        int *bqueue = A.get<int>(hb[7] > 0 ? hb[7] : 1);
        if (!bqueue) return OBIA_E_NOMEM;
        hipLaunchKernelGGL(cc_big_mark_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, d_probs, np, parent, size, n, newlab, big_box);
        hipLaunchKernelGGL(cc_split_kernel, dim3(cdiv(n_big, 64)), dim3(64), 0, ctx->stream, d_probs, np, parent, size, big_root, big_qoff,
                           big_box, n_big, bqueue);
        OBIA_HIP_TRY(hipMemsetAsync(counters, 0, sizeof(int) * 8, ctx->stream));
        rb = nullptr;   // cc_split_kernel made new roots: the ranking passes test parent[] again
        hipLaunchKernelGGL(cc_rank_blocksum_kernel, dim3(nb), dim3(SCAN_NT), 0, ctx->stream, d_probs, np, parent, size, n, block_sums, counters, rb);
        hipLaunchKernelGGL(cc_rank_scan_kernel, dim3(3), dim3(1024), 0, ctx->stream, block_sums, nb, counters);
        OBIA_TRY(read_back(ctx, hc, counters, sizeof(hc)));
    }
    const int n_small = hc[0], small_px = hc[1], n_surv = hc[2];
    int *small_list = A.get<int>(n_small > 0 ? n_small : 1);
    int *small_qoff = A.get<int>(n_small > 0 ? n_small : 1);
    int *target = A.get<int>(n_small > 0 ? n_small : 1);
    int *queue = A.get<int>(small_px > 0 ? small_px : 1);
    if (!small_list || !small_qoff || !target || !queue) return OBIA_E_NOMEM;
    debug_sync(ctx, "cc: rank counts");
    if (rb)
        hipLaunchKernelGGL(cc_rank_apply_bits_kernel, dim3(nb), dim3(64), 0, ctx->stream, d_probs, np, size, n, block_sums, newlab,
                           small_list, small_qoff, rb);
    else
        hipLaunchKernelGGL(cc_rank_apply_kernel, dim3(nb), dim3(SCAN_NT), 0, ctx->stream, d_probs, np, parent, size, n, block_sums,
                           newlab, small_list, small_qoff, counters, rb);
    int *settle_arr = nullptr;
    int32_t *visited = nullptr;
    if (n_small > 0) {
        int *settle = settle_arr = A.get<int>(n_small), *work_a = A.get<int>(n_small), *work_b = A.get<int>(n_small), *tag = A.get<int>(n_small);
        // the dense code[] pass pays when small components are many (one per 64 pixels or more: `bench.py --bands 3` has one per 33,
        // compactness 0.25 on eight bands one per ~170, where the pass cost 0.5 ms per step more than it saved)
        // (developer switch OBIA_CC_CODE_DIV: pixels per small component below which the dense code[] pass and the pixel lists are used)
        static const long long code_div = std::getenv("OBIA_CC_CODE_DIV") ? atoll(std::getenv("OBIA_CC_CODE_DIV")) : 64;
        const bool use_code = (long long)n_small * code_div >= n;
        // the rounds' "moved" counters: a ring of words cleared once (a fill per round was a launch and a gap per round)
        constexpr int RING = 64;
        int *ring = A.get<int>(RING);
        if (!ring) return OBIA_E_NOMEM;
        OBIA_HIP_TRY(hipMemsetAsync(ring, 0, sizeof(int) * RING, ctx->stream));
        long long ring_pos = 0;
        int *code = use_code ? A.get<int>(n) : parent;
        if (!settle || !work_a || !work_b || !tag || !code) return OBIA_E_NOMEM;
        // the walks' visited map: all zero between calls (every walk clears its marks): no 4-byte-per-pixel fill per batch (round 4)
        if ((size_t)n > ctx->cc_visited_px || ctx->cc_visited_dirty) {
            if ((size_t)n > ctx->cc_visited_px) {
                OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
                if (ctx->cc_visited) (void)hipFree(ctx->cc_visited);
                ctx->cc_visited = nullptr; ctx->cc_visited_px = 0;
                const size_t want = (size_t)n + (size_t)n / 8;
                if (hipMalloc(&ctx->cc_visited, want * sizeof(int32_t)) != hipSuccess) { (void)hipGetLastError(); set_error("out of device memory (visited map of %zu pixels)", want); return OBIA_E_NOMEM; }
                ctx->cc_visited_px = want;
            }
            OBIA_HIP_TRY(hipMemsetAsync(ctx->cc_visited, 0, sizeof(int32_t) * ctx->cc_visited_px, ctx->stream));
        }
        ctx->cc_visited_dirty = true;     // until this call has queued its last walk
        visited = ctx->cc_visited;
        if (use_code) hipLaunchKernelGGL(cc_code_kernel, dim3(gs), dim3(256), 0, ctx->stream, parent, newlab, n, code);
        // The settle times solve  t(S) = first pixel of S after min over its neighbours N of t(N)  (t(N) = first pixel of N for a
        // surviving N; "never" when S has no later pixel).  A time is decided by strictly EARLIER times, so the system has exactly
        // one solution, and a monotone iteration reaches it from either side:
        //   from below (optimistic: every small component labelled at its first pixel) -- one round when small components sit
        //     between surviving ones, which is what a SLIC sweep leaves; but two small components that only see each other
        //     leapfrog one pixel per round, and a map in which nearly every component is small needs thousands of rounds
        //     (8.4 s per step of `bench.py --bands 3` in round 2);
        //   from above (pessimistic: never) -- a round carries the labels one component further away from the surviving
        //     components: a few rounds more than from below in the SLIC regime, a few dozen where from below needs thousands,
        //     none at all where nothing survives (numpy model, profiles/r03_notes.md: 1165 -> 1, 856 -> 25 rounds).
        // So: from below for a few rounds; if that has not converged, start again from above (also at once when small components
        // outnumber the surviving ones eight to one).  Work list as before: round 0 evaluates every small component, a later round
        // the small neighbours of the components whose time moved; reads of a neighbour's time may be stale inside a round (every
        // value read lies between the start and the solution); never cut short: an unconverged round would write label 0.
        constexpr int ROUNDS_FROM_BELOW = 5;
        bool converged = false;
        // many small components and the reference's "written back as unset" rule in force (start_label 1): dense rounds for the times,
        // one walk per component for its neighbour (see "dense settle rounds"); OBIA_CC_WORKLIST: developer switch, the work list
        const bool dense = use_code && start_label == 1 && !std::getenv("OBIA_CC_WORKLIST");
        if (dense) {
            int *px_g = A.get<int>(small_px), *ccur = A.get<int>(n_small);
            int4 *px_cn = A.get<int4>(small_px);
            if (!px_g || !ccur || !px_cn) return OBIA_E_NOMEM;
            int maxh = 1;
            for (auto &P : probs) maxh = std::max(maxh, P.H);
            OBIA_HIP_TRY(hipMemsetAsync(ccur, 0, sizeof(int) * (size_t)n_small, ctx->stream));
            hipLaunchKernelGGL(cc_small_pixels_kernel, dim3(maxh, np), dim3(256), 0, ctx->stream, d_probs, code, small_qoff, ccur, px_g, px_cn, small_px);
            for (int side = (n_small > 8 * (long long)n_surv ? 1 : 0); side < 2 && !converged; ++side) {
                OBIA_HIP_TRY(hipMemsetAsync(tag, 0, sizeof(int) * (size_t)n_small, ctx->stream));
                hipLaunchKernelGGL(cc_settle_init_kernel, dim3(cdiv(n_small, 256)), dim3(256), 0, ctx->stream, small_list, n_small, settle, side);
                int n_items = n_small;
                const int *work_in = nullptr;
                const long long max_rounds = side == 0 ? ROUNDS_FROM_BELOW : (long long)small_px + 2;
                for (long long round = 0; round <= max_rounds; ++round) {
                    if (std::getenv("OBIA_DEBUG_CC")) fprintf(stderr, "[obia cc]   (lists) side %d round %lld: %d items\n", side, round, n_items);
                    int *cnt = ring + (ring_pos++ % RING);
                    if (ring_pos > RING && (ring_pos - 1) % RING == 0) OBIA_HIP_TRY(hipMemsetAsync(ring, 0, sizeof(int) * RING, ctx->stream));
                    hipLaunchKernelGGL(cc_settle_eval_kernel, dim3(cdiv(n_items, 64)), dim3(64), 0, ctx->stream, small_qoff, ccur, px_g, px_cn, small_px,
                                       settle, work_in, n_items, work_a, cnt, tag, (int)(round & 0x3fffffff));
                    int n_next = 0;
                    OBIA_TRY(read_back(ctx, &n_next, cnt, sizeof(int)));
                    if (n_next == 0) { converged = true; break; }
                    work_in = work_a;
                    std::swap(work_a, work_b);
                    n_items = n_next;
                }
            }
            if (converged)
                hipLaunchKernelGGL(cc_small_target_kernel, dim3(cdiv(n_small, 64)), dim3(64), 0, ctx->stream, d_probs, np, code, newlab, small_list,
                                   small_qoff, settle, queue, visited, target, n_small);
        }
        for (int side = (n_small > 8 * (long long)n_surv ? 1 : 0); !dense && side < 2 && !converged; ++side) {
            OBIA_HIP_TRY(hipMemsetAsync(tag, 0, sizeof(int) * (size_t)n_small, ctx->stream));
            hipLaunchKernelGGL(cc_settle_init_kernel, dim3(cdiv(n_small, 256)), dim3(256), 0, ctx->stream, small_list, n_small, settle, side);
            int n_items = n_small;
            const int *work_in = nullptr;
            // (times only move one way and a component's time is one of its own pixels or "never": at most small_px + 1 rounds
            // move something -- the bound of the second side)
            const long long max_rounds = side == 0 ? ROUNDS_FROM_BELOW : (long long)small_px + 2;
            for (long long round = 0; round <= max_rounds; ++round) {
                if (std::getenv("OBIA_DEBUG_CC")) fprintf(stderr, "[obia cc]   side %d round %lld: %d items\n", side, round, n_items);
                int *cnt = ring + (ring_pos++ % RING);
                if (ring_pos > RING && (ring_pos - 1) % RING == 0) OBIA_HIP_TRY(hipMemsetAsync(ring, 0, sizeof(int) * RING, ctx->stream));
                if (use_code)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(cc_small_bfs_kernel<true>), dim3(cdiv(n_items, 64)), dim3(64), 0, ctx->stream, d_probs, np, code,
                                       newlab, small_list, small_qoff, start_label, settle, queue, visited, target, work_in, n_items, work_a,
                                       cnt, tag, (int)(round & 0x3fffffff));
                else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(cc_small_bfs_kernel<false>), dim3(cdiv(n_items, 64)), dim3(64), 0, ctx->stream, d_probs, np, code,
                                       newlab, small_list, small_qoff, start_label, settle, queue, visited, target, work_in, n_items, work_a,
                                       cnt, tag, (int)(round & 0x3fffffff));
                int n_next = 0;
                OBIA_TRY(read_back(ctx, &n_next, cnt, sizeof(int)));
                if (n_next == 0) { converged = true; break; }
                work_in = work_a;
                std::swap(work_a, work_b);
                n_items = n_next;
            }
        }
        if (!converged) { set_error("connectivity enforcement: settle rounds did not converge (%d small components)", n_small); return OBIA_E_INVALID; }
        ctx->cc_visited_dirty = false;
    }
    if (std::getenv("OBIA_DEBUG_CC"))   // developer aid: the regime of this batch
        fprintf(stderr, "[obia cc] %lld px, %d problems: %d surviving, %d small components (%d px)\n", n, np, n_surv, n_small, small_px);
    debug_sync(ctx, "cc: rank apply + small components");
    // (`settle` is free again: it takes the components' final labels)
    if (n_small > 0)
        hipLaunchKernelGGL(cc_small_final_kernel, dim3(cdiv(n_small, 256)), dim3(256), 0, ctx->stream, parent, newlab, target, n_small, n_small + 1, settle_arr);
    const CcResolve R{parent, newlab, settle_arr, start_label, mask_label};
    if (deferred) *deferred = R;
    else hipLaunchKernelGGL(cc_relabel_kernel, dim3(gs), dim3(256), 0, ctx->stream, R, n, labels_out);
    OBIA_HIP_TRY(hipGetLastError());
    if (h_n_labels_out) *h_n_labels_out = n_surv;
    return OBIA_OK;
}

int enforce_connectivity_dev(obia_ctx *ctx, const int32_t *labels_in, int H, int W, int min_size, int max_size,
                             int start_label, int32_t *labels_out, int *h_n_labels_out) {
    std::vector<CcProblem> probs(1, CcProblem{H, W, 0, min_size, max_size > 0 ? max_size : 1});
    return enforce_connectivity_batch(ctx, probs, labels_in, (long long)H * W, start_label, labels_out, h_n_labels_out);
}

}  // namespace obia
