// slic_sweep.hip -- the per-sweep kernels of the SLIC engine: centroid finalise + binning, and the
// pixel-centric assign sweep with the centroid update fused in.
//
// Restates one iteration of _slic_cython (scikit-image _slic.pyx 0.18.3, reached from
// obia/segmentation/segment_boundaries.py:51; oracle/obia_oracle.c: obia_oracle_slic_core):
//   reference: for k ascending: scatter d(k, pixel) into the (4S+1)^2 window of centroid k, keep it where
//              `distance > d` (ties stay with the lowest k); then sum pixel coordinates and colours per label.
//   here:      for each pixel: lexicographic minimum of (d, k) over the centroids whose window contains it.
//              Same candidate set (windows are computed with the reference's float expressions and
//              truncations), same float32 operation order for d (compiled with -ffp-contract=off), same tie
//              rule; the sums are exact integers (coordinates) / 64-bit fixed point (colours).
//
// Kernel shape (gfx950): one 256-thread workgroup per 64x64 pixel tile (SWEEP_TW x SWEEP_TH, slic.hpp).
//   1. the workgroup stages into LDS the header of every centroid whose window intersects the tile (lanes walk the per-bin
//      linked lists built by slic_prep_kernel) and ranks the slots by centroid index k (rank table: broadcast LDS reads, no
//      serial chain), so that the reference's tie rule (lowest k wins) becomes a comparison of ranks;
//   2. each wave walks four 16x16 footprints (one 16-row band of the tile); a lane owns a 1x4 vertical strip = one quad row of
//      the feature layout (slic.hpp: feat_block_f4): ONE 16-byte load per channel brings that channel of its four pixels.
//      Lanes first score the staged candidates in parallel (one candidate per lane): window-intersects-footprint and a lower
//      bound `lb` of the spatial term over the footprint (plus, at low compactness, a lower bound of the colour term from the
//      footprint's colour box); candidates are then visited in ascending lb and the walk stops when lb exceeds the largest
//      current best distance in the wave (d >= lb, float add/mul are monotone, so nothing that is skipped could have won or
//      tied).  A visit reads the candidate's record through the scalar unit (SGPR operands, no vector / LDS instruction) and
//      evaluates the lane's four pixels as two packed-f32 pairs.  A pixel's state is ONE 64-bit key
//      float_bits(d) << 32 | rank : distances are non-negative floats, whose bit patterns order like their values, so
//      `new_key < key` is exactly the lexicographic (d, k) comparison of the reference -- no separate tie path;
//   3. the centroid update is fused: every feature is converted once to 32-bit fixed point (a power-of-two scale:
//      exact for all but the smallest 1/64 of the value range), per-lane run sums are plain int32, they are
//      transposed through a conflict-free LDS scratch so that 8 lanes x CP fields fold the wave's 64 strips
//      sequentially, and only the few resulting (centroid, field) partials touch the workgroup's LDS
//      accumulators (64-bit integer atomics); one packed global atomic record per (tile, centroid) at the end.
//      Integer sums do not depend on the order of the atomics: the segmentation is bit-reproducible.
// A tile that meets more candidates than fit in LDS (tiny S, clustered centroids) takes slow_tile(), which
// reads the bins directly; correctness never depends on the LDS capacity.
#include "slic.hpp"

#include <type_traits>
#include <hip/hip_ext.h>

#include <cstdlib>

namespace obia {

#ifdef OBIA_STAMP
// Diagnostic build only (tools/build_variant.sh tl -DOBIA_STAMP, tools/timeline_run.py): per-wave timeline, no atomics.
// Record of a wave (24 qwords, waves of tile t at 4t .. 4t+3): [0..11] phase ticks (s_memtime), [12] start, [13] end,
// [14] / [15] start / end on the constant 100-MHz clock, [16..21] counters.
__device__ unsigned long long g_tl[65536 * 4 * 24];
#define STAMP_DECL unsigned long long st_t = clock64(), st_t0 = st_t, st_r0 = __builtin_amdgcn_s_memrealtime(), st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_cnt[6] = {0, 0, 0, 0, 0, 0};
#define STAMP_COUNT(i, v) st_cnt[i] += (v);
#define STAMP(i) { const unsigned long long st_n = clock64(); st_acc[i] += st_n - st_t; st_t = st_n; }
#ifndef OBIA_STAMP_KIND
#define OBIA_STAMP_KIND 0   /* 0: the kernels that fold colours, 1: the lean pre-pass kernel (TL_MASK=1) */
#endif
#define STAMP_FLUSH if (accumulate && (LEAN ? 1 : 0) == OBIA_STAMP_KIND && (threadIdx.x & 63) == 0 && gtile < 65536) { unsigned long long *st_o = g_tl + ((size_t)gtile * 4 + (threadIdx.x >> 6)) * 24; for (int st_i = 0; st_i < 12; ++st_i) st_o[st_i] = st_acc[st_i]; st_o[12] = st_t0; st_o[13] = clock64(); st_o[14] = st_r0; st_o[15] = __builtin_amdgcn_s_memrealtime(); for (int st_i = 0; st_i < 6; ++st_i) st_o[16 + st_i] = st_cnt[st_i]; }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_COUNT(i, v)
#define STAMP_FLUSH
#endif

typedef float v2f __attribute__((ext_vector_type(2)));   // two pixels of a lane's strip: v_pk_{add,mul}_f32 operate on both at once
__device__ __forceinline__ v2f splat(float s) { return (v2f){s, s}; }

// one feature of one pixel in the quad-row plane layout (slic.hpp: feat_block_f4) -- rare paths only
__device__ __forceinline__ float feat_at(const float *__restrict__ feat, const SlicProblem &P, int CP, int y, int x, int ch) {
    return feat[((P.feat_off + (((long long)(y >> 2) * P.XB + (x >> 4)) * CP + ch) * 16 + (x & 15)) << 2) + (y & 3)];
}

constexpr int NT = 256;
constexpr int FB = 16;          // wave footprint side
constexpr int PPT = 4;          // pixels per lane (vertical strip)
#ifndef LEAN_WAVES
#define LEAN_WAVES 8
#endif
#ifndef OBIA_SWEEP_GROUPS_DEFAULT
#define OBIA_SWEEP_GROUPS_DEFAULT 1
#endif
#ifndef ASSIGN_WAVES
#define ASSIGN_WAVES 6
#endif
#ifndef OBIA_XCD_GROUP
#define OBIA_XCD_GROUP 2
#endif
constexpr int MAXC = SWEEP_MAXC;   // LDS candidate slots of a tile; must stay <= 128: the slot number rides in the low 7 bits of the
                                // scoring keys (and below 255: a byte of the footprint lists)

// ---- candidate lists (round 4) ----------------------------------------------------------------------------------------
// Which centroids can reach a tile changes little from sweep to sweep (a converging centroid moves by hundredths of a pixel),
// but finding them -- bin heads -> list nodes -> links, an atomic slot counter, an O(n^2) rank table -- was a quarter of a
// wave's lifetime in every one of the twenty sweeps of a batch.  A tile therefore keeps its candidates as a LIST: the centroid
// indices in ascending order (a candidate's slot IS its rank under the reference's tie rule, lowest k wins) and, per thread, the
// slot it scores in each of its wave's four footprints (at most 64 candidates per footprint: ONE scoring round, one key
// register).  The list is a SUPERSET that stays valid while centroids move:
//   (A) every centroid stays within LIST_M pixels (per axis) of a reference position `ref`; the centroid step (slic_prep_*)
//       checks that, and a centroid that left its margin takes its new position as reference and asks every tile its window,
//       grown by LIST_M + 1, meets to rebuild (tl_req[tile] = sweep);
//   (B) a tile's list holds every centroid whose CURRENT window, grown by 2 * LIST_M + 2, met the tile (or the footprint)
//       when the list was built.
// With (A) the window of a centroid lies inside its reference window grown by LIST_M + 1 (the window bounds are monotone in
// the position and truncated to integers), and the reference window inside the window at build time grown by LIST_M + 1 more:
// (B) therefore contains every centroid whose window meets the tile now.  Whether a listed candidate's window really meets a
// footprint is decided by the exact test on its freshly loaded header, as before: labels are bit-identical with and without lists.
// A tile whose grown candidate set does not fit (dense centroids) is staged from the bins with the exact windows in every
// sweep, as in rounds 1-3; beyond MAXC exact candidates slow_tile() takes over.
#ifndef OBIA_LIST_MARGIN
#define OBIA_LIST_MARGIN 2
#endif
constexpr int LIST_M = OBIA_LIST_MARGIN;
constexpr int LIST_GM = LIST_M + 1;
constexpr int LIST_G2 = 2 * LIST_M + 2;
#ifndef OBIA_LIST_GIVE_UP
#define OBIA_LIST_GIVE_UP 2
#endif
constexpr int LIST_GIVE_UP = OBIA_LIST_GIVE_UP;   // short-lived builds in a row after which a tile stays unlisted

// (A): called by the centroid step for a centroid at (cy, cx) with the window [y0, y1) x [x0, x1) it just computed
__device__ __forceinline__ void list_margin_check(const SlicProblem &P, int k, float cy, float cx, int y0, int y1, int x0, int x1,
                                                  bool first, float *__restrict__ ref, int *__restrict__ tl_req, int sweep_id) {
    if (!ref) return;
    float2 *rp = reinterpret_cast<float2 *>(ref) + k;
    if (first) { *rp = make_float2(cy, cx); return; }   // (every tile builds its list in the first sweep)
    const float2 r = *rp;
    if (fabsf(cy - r.x) < (float)LIST_M && fabsf(cx - r.y) < (float)LIST_M) return;
    *rp = make_float2(cy, cx);
    int ty_lo = (y0 - LIST_GM) / SWEEP_TH; if (y0 - LIST_GM < 0) ty_lo = 0;
    int ty_hi = (y1 - 1 + LIST_GM) / SWEEP_TH; if (ty_hi > P.tiles_y - 1) ty_hi = P.tiles_y - 1;
    int tx_lo = (x0 - LIST_GM) / SWEEP_TW; if (x0 - LIST_GM < 0) tx_lo = 0;
    int tx_hi = (x1 - 1 + LIST_GM) / SWEEP_TW; if (tx_hi > P.tiles_x - 1) tx_hi = P.tiles_x - 1;
    for (int ty = ty_lo; ty <= ty_hi; ++ty)
        for (int tx = tx_lo; tx <= tx_hi; ++tx) tl_req[P.tile_off + ty * P.tiles_x + tx] = sweep_id;
}

// K3 + binning: finalise centroids from the accumulator records (or the seeds on the very first sweep),
// write the centroid record {cy, cx, y0, y1, x0, x1, link, -, colour[CP]} (link: next centroid of the bin's list) and push the centroid on the
// linked list of the bin that holds its current position.
// G = RQ lanes per centroid: lane q of a group owns qword q of the 128-B (256-B) accumulator record, so the record is
// read and cleared with one coalesced access per wave; n and the coordinate sums reach the group by shuffle.
template <int G>
__global__ __launch_bounds__(256) void slic_prep_kernel(const SlicProblem *__restrict__ probs,
                                                        const int *__restrict__ cent_prob, int total_cent, int CP,
                                                        int first, int slic_zero, const float *__restrict__ seed,
                                                        unsigned long long *__restrict__ acc, double inv_fscale,
                                                        float *__restrict__ cent, int *__restrict__ head,
                                                        int *__restrict__ head_other,
                                                        int total_cells, int *__restrict__ bin_stamp, int sweep_id,
                                                        int k_base, int cell_base, float *__restrict__ ref, int *__restrict__ tl_req) {
    // exit_on_fixed_point: a centroid whose record differs from the previous sweep's stamps the bin it leaves and the
    // bin it enters with the sweep number; the sweep kernel skips a tile none of whose bins was stamped since the tile
    // was last evaluated (same candidate records => same labels, same partial sums, replayed from the tile's cache).
    // the bin heads are double-buffered: while this sweep fills `head`, the buffer of the NEXT sweep is reset here
    // (saves one memset launch per sweep)
    // (a launch covers the centroids [k_base, total_cent) and the bins [cell_base, total_cells): one group of problems)
    for (int i = cell_base + blockIdx.x * blockDim.x + threadIdx.x; i < total_cells; i += gridDim.x * blockDim.x) head_other[i] = -1;
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = k_base + gt / G, q = gt % G;
    const int lane = threadIdx.x & 63, gl0 = lane - q;       // first lane of the group inside the wave
    const bool in_range = k < total_cent;
    const int p = in_range ? cent_prob[k] : -1;
    // everything whose address depends on k alone is requested before the problem descriptor (which depends on cent_prob[k]) is
    // waited for: the kernel is a chain of dependent round trips (4 -> 3), 180 launches per step of the tiler
    const int RS = CENT_REC + CP;
    float *rec = cent + (size_t)(in_range ? k : 0) * RS;
    const float old_cy_l = in_range ? rec[0] : 0.0f, old_cx_l = in_range ? rec[1] : 0.0f;
    const float oldc_l = (in_range && !first && q < CP) ? rec[CENT_REC + q] : 0.0f;
    const unsigned long long aq_l = (in_range && !first) ? acc[(size_t)k * G + q] : 0ull;
    SlicProblem P;
    bool live = p >= 0;
    if (live) { P = probs[p]; live = (k - P.cent_off) < P.K; }
    float cy = 0.0f, cx = 0.0f;
    bool moved = false;
    const float old_cy = live ? old_cy_l : 0.0f, old_cx = live ? old_cx_l : 0.0f;
    if (first) {
        if (live) {
            cy = seed[2 * (size_t)k];
            cx = seed[2 * (size_t)k + 1];
            if (q < CP) rec[CENT_REC + q] = 0.0f;   // initial centroid colour is zero (slic_superpixels.py:298-300)
        }
        moved = true;
    } else {
        unsigned long long *a = acc + (size_t)(live ? k : 0) * G;
        const unsigned long long aq = live ? aq_l : 0ull;
        const float oldc = (live && q < CP) ? oldc_l : 0.0f;
        if (live) a[q] = 0ull;
        // every lane takes part in the shuffles (dead groups carry zeros).  n, sum_y and sum_x are full 64-bit words:
        // sum_y reaches 2^32 as soon as (pixels of a cluster) x (row) does -- a coarse segmentation of a big raster
        const unsigned long long nq = __shfl(aq, gl0 + CP);
        const unsigned long long syq = __shfl(aq, gl0 + CP + 1);
        const unsigned long long sxq = __shfl(aq, gl0 + CP + 2);
        const float fn = (float)nq;
        // segments[k, c] /= n  in float32; n == 0 -> 0/0 = NaN centroid, as in the reference
        cy = (float)syq / fn;
        cx = (float)sxq / fn;
        bool mq = false;
        if (live && q < CP) {
            const float sum = (float)((double)(long long)aq * inv_fscale);
            const float v = sum / fn;
            mq = __float_as_uint(v) != __float_as_uint(oldc);
            rec[CENT_REC + q] = v;
        }
        const unsigned long long mb = __ballot(mq);
        const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << gl0);
        moved = (mb & gmask) != 0ull;
        moved |= __float_as_uint(cy) != __float_as_uint(old_cy) || __float_as_uint(cx) != __float_as_uint(old_cx);
    }
    if (!live || q != 0) return;
    // SLIC-zero: slot 7 of the record carries max_dist_color[k] from sweep to sweep (1 before the first sweep; raised by
    // slic_maxdist_kernel after every centroid update); otherwise the slot is unused
    const float mdc = slic_zero ? ((first || slic_zero == 2) ? 1.0f : rec[7]) : 0.0f;   // 2: first sweep of the colour pass
    if (bin_stamp && moved && !first && old_cy == old_cy && old_cx == old_cx) {   // the bin it leaves
        int oby = (int)(old_cy / (float)P.sy), obx = (int)(old_cx / (float)P.sx);
        oby = oby < 0 ? 0 : (oby >= P.ncy ? P.ncy - 1 : oby);
        obx = obx < 0 ? 0 : (obx >= P.ncx ? P.ncx - 1 : obx);
        bin_stamp[P.cell_off + oby * P.ncx + obx] = sweep_id;
    }
    float4 *hrec = reinterpret_cast<float4 *>(rec);   // records are 16-byte aligned (RS is a multiple of 4)
    if (!(cy == cy) || !(cx == cx)) {   // NaN centroid: its window is empty, it is never binned
        hrec[0] = make_float4(cy, cx, 0.0f, 0.0f);
        hrec[1] = make_float4(0.0f, 0.0f, __int_as_float(-1), mdc);
        return;
    }
    // z/y/x window of _slic_cython: (ssize_t)max(c - 2*step, 0) .. (ssize_t)min(c + 2*step + 1, size)
    float fy0 = cy - (float)(2 * P.sy); fy0 = (0.0f > fy0) ? 0.0f : fy0;
    float fy1 = (cy + (float)(2 * P.sy)) + 1.0f; fy1 = ((float)P.H < fy1) ? (float)P.H : fy1;
    float fx0 = cx - (float)(2 * P.sx); fx0 = (0.0f > fx0) ? 0.0f : fx0;
    float fx1 = (cx + (float)(2 * P.sx)) + 1.0f; fx1 = ((float)P.W < fx1) ? (float)P.W : fx1;
    int by = (int)(cy / (float)P.sy), bx = (int)(cx / (float)P.sx);
    by = by < 0 ? 0 : (by >= P.ncy ? P.ncy - 1 : by);
    bx = bx < 0 ? 0 : (bx >= P.ncx ? P.ncx - 1 : bx);
    // the link to the next centroid of the bin's list rides in slot 6 of the record (a centroid's index is its record's index):
    // a list node is ONE 32-byte read, not a read plus a lone dword from a second array
    const int link = atomicExch(&head[P.cell_off + by * P.ncx + bx], k);
    hrec[0] = make_float4(cy, cx, __int_as_float((int)fy0), __int_as_float((int)fy1));
    hrec[1] = make_float4(__int_as_float((int)fx0), __int_as_float((int)fx1), __int_as_float(link), mdc);
    if (bin_stamp && moved) bin_stamp[P.cell_off + by * P.ncx + bx] = sweep_id;   // the bin it enters (or changed in)
    list_margin_check(P, k, cy, cx, (int)fy0, (int)fy1, (int)fx0, (int)fx1, first != 0, ref, tl_req, sweep_id);
}

// The same step with ONE LANE per centroid (round 3).  The grouped kernel above spends 16 lanes on a centroid: 22 000 waves per
// launch of a bench batch, each alive for 3.5 us, on a chip that is a quarter full for the 17-20 us the launch lasts (PMC: 2 200
// resident waves on average) -- the launch is bound by wave dispatch, and there are 180 such launches per step of the tiler.  Here a
// lane reads its centroid's accumulator record (16-byte loads, all in flight together), clears it, and writes the centroid record:
// sixteen times fewer waves, and one dependent round trip less (the problem comes with the block index, not through cent_prob[k]).
// Same arithmetic, operation for operation.
template <int CP>
__global__ __launch_bounds__(64) void slic_prep_lane_kernel(const SlicProblem *__restrict__ probs, int p_base,
                                                             int first, int slic_zero, const float *__restrict__ seed,
                                                             unsigned long long *__restrict__ acc, int RQ, double inv_fscale,
                                                             float *__restrict__ cent, int *__restrict__ head,
                                                             int *__restrict__ head_other,
                                                             int total_cells, int *__restrict__ bin_stamp, int sweep_id,
                                                             int cell_base, float *__restrict__ ref, int *__restrict__ tl_req) {
    constexpr int RS = CENT_REC + CP;
    constexpr int NQ = (CP + 3 + 1) / 2;   // 16-byte pairs of the accumulator record that are in use: colours | n | sum_y | sum_x
    {   // the bins of the NEXT sweep (see slic_prep_kernel)
        const int nthr = gridDim.x * gridDim.y * blockDim.x;
        for (int i = cell_base + (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < total_cells; i += nthr) head_other[i] = -1;
    }
    // grid = (blocks of 64 centroids, problems of the group): the problem descriptor is workgroup-uniform (scalar loads that depend
    // on nothing), so the chain of dependent round trips is descriptor -> records -> bin head
    const SlicProblem P = probs[p_base + blockIdx.y];
    const int kl = blockIdx.x * blockDim.x + threadIdx.x;
    if (kl >= P.K) return;
    const int k = P.cent_off + kl;
    float4 *hrec = reinterpret_cast<float4 *>(cent + (size_t)k * RS);   // records are 16-byte aligned (RS is a multiple of 4)
    ulonglong2 *a2 = reinterpret_cast<ulonglong2 *>(acc + (size_t)k * RQ);
    const float4 h0 = hrec[0], h1 = hrec[1];
    float4 oc[CP / 4];
    ulonglong2 aq[NQ];
    if (!first) {
        // (the old colours are only compared with the new ones, for the stamps of exit_on_fixed_point: not read otherwise)
        if (bin_stamp) {
#pragma unroll
            for (int i = 0; i < CP / 4; ++i) oc[i] = hrec[2 + i];
        } else {
#pragma unroll
            for (int i = 0; i < CP / 4; ++i) oc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) aq[i] = a2[i];
    }
    const float old_cy = h0.x, old_cx = h0.y;
    float cy, cx;
    bool moved;
    if (first) {
        cy = seed[2 * (size_t)k];
        cx = seed[2 * (size_t)k + 1];
#pragma unroll
        for (int i = 0; i < CP / 4; ++i) hrec[2 + i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // initial centroid colour is zero (slic_superpixels.py:298-300)
        moved = true;
    } else {
        unsigned long long q[2 * NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) { q[2 * i] = aq[i].x; q[2 * i + 1] = aq[i].y; a2[i] = make_ulonglong2(0ull, 0ull); }
        // n, sum_y and sum_x are full 64-bit words (sum_y reaches 2^32 as soon as (pixels of a cluster) x (row) does)
        // uint64 -> float by way of double (exact below 2^53, which a count or a coordinate sum never reaches, so the one rounding is
        // the same): the direct conversion expands to a 64-bit shift by a register, and when that register is the last one the kernel
        // allocates the MI355X reads a wrong shift amount (tools/check_shift64.py -- the build fails on such an instruction)
        const float fn = (float)(double)q[CP];
        // segments[k, c] /= n  in float32; n == 0 -> 0/0 = NaN centroid, as in the reference
        cy = (float)(double)q[CP + 1] / fn;
        cx = (float)(double)q[CP + 2] / fn;
        moved = false;
        float nc[CP];
#pragma unroll
        for (int ch = 0; ch < CP; ++ch) {
            const float sum = (float)((double)(long long)q[ch] * inv_fscale);
            nc[ch] = sum / fn;
        }
#pragma unroll
        for (int i = 0; i < CP / 4; ++i) {
            moved |= __float_as_uint(nc[4 * i]) != __float_as_uint(oc[i].x) || __float_as_uint(nc[4 * i + 1]) != __float_as_uint(oc[i].y)
                  || __float_as_uint(nc[4 * i + 2]) != __float_as_uint(oc[i].z) || __float_as_uint(nc[4 * i + 3]) != __float_as_uint(oc[i].w);
            hrec[2 + i] = make_float4(nc[4 * i], nc[4 * i + 1], nc[4 * i + 2], nc[4 * i + 3]);
        }
        moved |= __float_as_uint(cy) != __float_as_uint(old_cy) || __float_as_uint(cx) != __float_as_uint(old_cx);
    }
    // SLIC-zero: slot 7 of the record carries max_dist_color[k] from sweep to sweep (see slic_prep_kernel)
    const float mdc = slic_zero ? ((first || slic_zero == 2) ? 1.0f : h1.w) : 0.0f;
    if (bin_stamp && moved && !first && old_cy == old_cy && old_cx == old_cx) {   // the bin it leaves
        int oby = (int)(old_cy / (float)P.sy), obx = (int)(old_cx / (float)P.sx);
        oby = oby < 0 ? 0 : (oby >= P.ncy ? P.ncy - 1 : oby);
        obx = obx < 0 ? 0 : (obx >= P.ncx ? P.ncx - 1 : obx);
        bin_stamp[P.cell_off + oby * P.ncx + obx] = sweep_id;
    }
    if (!(cy == cy) || !(cx == cx)) {   // NaN centroid: its window is empty, it is never binned
        hrec[0] = make_float4(cy, cx, 0.0f, 0.0f);
        hrec[1] = make_float4(0.0f, 0.0f, __int_as_float(-1), mdc);
        return;
    }
    // z/y/x window of _slic_cython: (ssize_t)max(c - 2*step, 0) .. (ssize_t)min(c + 2*step + 1, size)
    float fy0 = cy - (float)(2 * P.sy); fy0 = (0.0f > fy0) ? 0.0f : fy0;
    float fy1 = (cy + (float)(2 * P.sy)) + 1.0f; fy1 = ((float)P.H < fy1) ? (float)P.H : fy1;
    float fx0 = cx - (float)(2 * P.sx); fx0 = (0.0f > fx0) ? 0.0f : fx0;
    float fx1 = (cx + (float)(2 * P.sx)) + 1.0f; fx1 = ((float)P.W < fx1) ? (float)P.W : fx1;
    int by = (int)(cy / (float)P.sy), bx = (int)(cx / (float)P.sx);
    by = by < 0 ? 0 : (by >= P.ncy ? P.ncy - 1 : by);
    bx = bx < 0 ? 0 : (bx >= P.ncx ? P.ncx - 1 : bx);
    const int link = atomicExch(&head[P.cell_off + by * P.ncx + bx], k);
    hrec[0] = make_float4(cy, cx, __int_as_float((int)fy0), __int_as_float((int)fy1));
    hrec[1] = make_float4(__int_as_float((int)fx0), __int_as_float((int)fx1), __int_as_float(link), mdc);
    if (bin_stamp && moved) bin_stamp[P.cell_off + by * P.ncx + bx] = sweep_id;   // the bin it enters (or changed in)
    list_margin_check(P, k, cy, cx, (int)fy0, (int)fy1, (int)fx0, (int)fx1, first != 0, ref, tl_req, sweep_id);
}

// float feature -> 32-bit fixed point.  fs is a power of two chosen in slic_features_finish so that |f * fs| < 2^29:
// the product is exact, the conversion truncates only what lies below 2^-29 of the largest feature (floats above
// 1/64 of it are integers after the scaling), four of them add up in an int32 and every total stays below 2^62.
__device__ __forceinline__ int to_fixed32(float f, float fs) { return (int)(f * fs); }

// ---- wave-wide reductions on DPP (no LDS traffic) ---------------------------------------------------
// non-negative floats order like their bit patterns, so min/max run on unsigned integers.  Written as
// inline asm so that every butterfly step is ONE v_{min,max}_u32 with a DPP operand (hipcc lowers the
// update_dpp builtin to v_mov + v_mov_dpp + v_min, three VALU issues per step).  A DPP operand written by
// the previous VALU instruction needs two wait states: the s_nop 1 in front of every step.
#define OBIA_WAVE_REDUCE(OP)                                                                      \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
                 "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"      \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"           \
                 "s_nop 1"                                                                        \
                 : "+v"(v))
__device__ __forceinline__ unsigned wave_umin(unsigned v) {
    OBIA_WAVE_REDUCE("v_min_u32_dpp");   // every lane now holds the minimum of its row of 16
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    unsigned r;   // the four row minima meet on the scalar unit (hipcc would move them back to vector registers for a v_min3)
    asm("s_min_u32 %0, %1, %2\n\ts_min_u32 %0, %0, %3\n\ts_min_u32 %0, %0, %4" : "=&s"(r) : "s"(a), "s"(b), "s"(c), "s"(d) : "scc");
    return r;
}

// LDS written by some lanes of a wave and read by other lanes of the same wave: the LDS pipe executes a
// wave's instructions in order, so only the COMPILER must be kept from moving accesses across this point.
// (A workgroup-scope fence would also emit s_waitcnt vmcnt(0) and stall on the label stores in flight.)
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// the lane's index inside its wave from the hardware's bit counts: two instructions wherever it is needed, so that NO register
// carries the thread index across the footprint loop (round 4: with the list state in a register the allocator spilled the
// thread index, and its reload at the head of every footprint waited for the feature loads in flight)
__device__ __forceinline__ int lane_now() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// one pixel straight into the global accumulator record (rare paths only): colours | n | sum_y | sum_x
template <int CP>
__device__ __forceinline__ void global_accumulate(unsigned long long *__restrict__ acc, int RQ, int k, unsigned y, unsigned x,
                                                  const float *f, float fs) {
    unsigned long long *a = acc + (size_t)k * RQ;
#pragma unroll
    for (int ch = 0; ch < CP; ++ch) atomicAdd(&a[ch], (unsigned long long)(long long)to_fixed32(f[ch], fs));
    atomicAdd(&a[CP], 1ull);
    atomicAdd(&a[CP + 1], (unsigned long long)y);
    atomicAdd(&a[CP + 2], (unsigned long long)x);
}

// store_labels: 0 = this sweep does not store its labels, 1 = it does but earlier sweeps of the batch did not (the very last
// sweep of a batch whose labels are only stored at the end), 2 = every sweep of the batch stores.  A valid pixel that no window
// reaches ("orphan") keeps the label of the sweep before: unless every sweep stored, that label is not in memory -- the pixel
// would come out with the fill value where the reference keeps the label of sweep N-1 -- so the flag is raised and the host
// repeats the batch with every sweep storing.  (Sweep 1 has no sweep before it: the fill value IS what the reference keeps.)
__device__ __forceinline__ bool orphan_needs_repeat(int store_labels, int sweep_id) {
    return store_labels == 0 || (store_labels == 1 && sweep_id > 1);
}

// Fallback for a tile whose candidate set does not fit the LDS slots: every lane scans the bins around
// each of its pixels directly in global memory.  Same arithmetic, no staging.
template <int CP, bool MASKED, bool IGNORE_COLOR, bool SLICZERO>
__device__ void slow_tile(const SlicProblem &P, int ty0, int tx0, const float *__restrict__ feat,
                          const uint8_t *__restrict__ mask, const float *__restrict__ cent,
                          const int *__restrict__ head, int32_t *__restrict__ labels,
                          unsigned long long *__restrict__ acc, int RQ, int accumulate, int accum_color, int start_label,
                          float fs, int store_labels, int *__restrict__ orphan_flag, int sweep_id, int nch) {
    constexpr int RS = CENT_REC + CP;
    const float w = P.spatial_w;
    for (int i = threadIdx.x; i < SWEEP_TW * SWEEP_TH; i += NT) {
        const int y = ty0 + i / SWEEP_TW, x = tx0 + i % SWEEP_TW;
        if (y >= P.H || x >= P.W) continue;
        const long long pix = P.pix_off + (long long)y * P.W + x;
        if (MASKED && mask[pix] == 0) { labels[pix] = start_label - 1; continue; }
        float f[CP];
#pragma unroll
        for (int ch = 0; ch < CP; ++ch) f[ch] = ch < nch ? feat_at(feat, P, CP, y, x, ch) : 0.0f;   // (padded planes are not written)
        int by_lo = (y - 2 * P.sy - 2) / P.sy; if (y - 2 * P.sy - 2 < 0) by_lo = 0;
        int by_hi = (y + 2 * P.sy + 2) / P.sy; if (by_hi > P.ncy - 1) by_hi = P.ncy - 1;
        int bx_lo = (x - 2 * P.sx - 2) / P.sx; if (x - 2 * P.sx - 2 < 0) bx_lo = 0;
        int bx_hi = (x + 2 * P.sx + 2) / P.sx; if (bx_hi > P.ncx - 1) bx_hi = P.ncx - 1;
        float best = INFINITY;
        int bk = -1;
        for (int by = by_lo; by <= by_hi; ++by)
            for (int bx = bx_lo; bx <= bx_hi; ++bx)
                for (int cur = head[P.cell_off + by * P.ncx + bx]; cur >= 0; cur = reinterpret_cast<const int *>(cent + (size_t)cur * RS)[6]) {
                    const float *rec = cent + (size_t)cur * RS;
                    const int *irec = reinterpret_cast<const int *>(rec);
                    if (!(y >= irec[2] && y < irec[3] && x >= irec[4] && x < irec[5])) continue;
                    // (`spacing`: the differences are scaled before they are squared, _slic.pyx; (1, 1) multiplies by 1.0f -- exact)
                    const float tyv = P.sp_y * (rec[0] - (float)y), txv = P.sp_x * (rec[1] - (float)x);
                    const float dy2 = tyv * tyv, dx2 = txv * txv;
                    float d = (dy2 + dx2) * w;
                    if (!IGNORE_COLOR) {
                        float dc = 0.0f;
#pragma unroll
                        for (int ch = 0; ch < CP; ++ch) { const float t = f[ch] - rec[CENT_REC + ch]; dc += t * t; }
                        d += SLICZERO ? dc / rec[7] : dc;
                    }
                    if (d < best || (d == best && cur < bk)) { best = d; bk = cur; }
                }
        int k = bk;
        if (k < 0) {   // `nearest` keeps the previous sweep's value
            if (orphan_needs_repeat(store_labels, sweep_id)) *orphan_flag = 1;   // ... which was not stored: the host repeats the batch with every sweep storing
            // (the previous label is in memory only when every sweep stores -- otherwise labels[] holds whatever the arena held, the
            // flag above has the batch repeated, and this sweep's sums are thrown away with it)
            if (store_labels == 2) {
                const int prev = labels[pix];
                if (prev >= start_label) k = prev - start_label + P.cent_off;
            }
        } else {
            labels[pix] = k - P.cent_off + start_label;
        }
        if (accumulate && k >= 0) {
            if (!accum_color) {
#pragma unroll
                for (int ch = 0; ch < CP; ++ch) f[ch] = 0.0f;
            }
            global_accumulate<CP>(acc, RQ, k, (unsigned)y, (unsigned)x, f, fs);
        }
    }
}

// K2: the sweep.  grid = (max tiles per problem, nprob).
// LEAN: a spatial-only pre-pass sweep that folds no colours (nine of the ten pre-pass sweeps): no feature registers, no
// colour scratch in LDS -- the same code with those parts compiled out, launched at a higher occupancy.
// NCH: the channels that exist (C <= CP).  Planes, records and accumulators come in groups of four channels; the padded ones
// hold zeros in the features and in every centroid, so a body compiled with NCH < CP neither loads them nor adds their
// (0 - 0)^2 = +0 to the colour distance: same bits, a quarter less traffic and colour arithmetic for 9 bands run as 12.
template <int CP, bool MASKED, bool IGNORE_COLOR, bool FIXPT, bool SLICZERO, bool LEAN, bool COLLB, int NCH = CP>
__device__ __forceinline__ void slic_assign_body(
    const SlicProblem *__restrict__ probs, const float *__restrict__ feat, const uint8_t *__restrict__ mask,
    const unsigned *__restrict__ mask4, const float *__restrict__ cent, const int *__restrict__ head,
    int32_t *__restrict__ labels, unsigned long long *__restrict__ acc, int RQ, int accumulate, int store_labels,
    int start_label, float fs, const int *__restrict__ bin_stamp, int *__restrict__ tile_lp,
    int *__restrict__ cache_k, unsigned long long *__restrict__ cache_q, int sweep_id, int use_cache,
    unsigned long long *__restrict__ px_counter, const int *__restrict__ tile_prob, int total_tiles_all,
    int *__restrict__ orphan_flag, int tiles_per_prob, const float *__restrict__ fbox, int tile_base, int nch_arg,
    int *__restrict__ tl_k, unsigned *__restrict__ tl_fp, int *__restrict__ tl_meta, const int *__restrict__ tl_req) {
    // channels that exist: a compile-time constant in the variants compiled per padding (NCH < CP), the launch argument in the
    // others (NCH == CP: the last pre-pass sweep, SLIC-zero, the fixed-point variant).  The planes of the padded channels are NOT
    // written by the feature pass since round 3 (nine bands: 36 instead of 48 bytes per pixel): nobody may read them.
    const int nch_rt = (NCH < CP) ? NCH : nch_arg;
    // COLLB (low compactness): the scoring adds a lower bound of the COLOUR term to the spatial one -- the distance of the
    // candidate's colour to the box of the footprint's features (slic.hpp: feat_boxes) -- and the visits drop from 13 to 7 per
    // footprint at compactness 0.25.  Valid because every step is monotone: |f - c| >= max(lo - c, c - hi, 0) per channel for
    // every pixel of the footprint, products of non-negatives and the sequential sums keep the order under rounding.
    // colours are folded by every sweep that runs this body without LEAN (the colour sweeps and the last pre-pass sweep)
    // and by none that runs it with LEAN: a compile-time constant either way
    constexpr int accum_color = LEAN ? 0 : 1;
#ifdef OBIA_ABL_NOACC
    accumulate = 0;   // ablation build: no centroid update at all
#endif
    // accumulate: fold this sweep's assignment into the accumulator records (off on the very last sweep);
    // accum_color: also fold the colours (off on the spatial-only pre-pass sweeps whose colour means are never
    // read: only the LAST pre-pass sweep seeds the colours of the main pass, slic_superpixels.py:310-318)
    // One-dimensional grid over all tiles of the batch, XCD-aware: workgroups are dealt round-robin over the 8 XCDs (blocks b
    // and b + 8 share one, each XCD has its own L2), so XCD x = blockIdx.x % 8 takes the x-th contiguous EIGHTH of the tiles
    // (problem by problem, raster order) instead of every eighth tile: neighbouring tiles -- which stage the same centroid
    // records and touch the same mask lines -- meet in one L2 instead of fetching them from HBM once per XCD.
#ifndef OBIA_XCD_GROUP
#define OBIA_XCD_GROUP 2
#endif
    constexpr int XG = OBIA_XCD_GROUP;   // consecutive tiles that share an XCD
    // (a launch covers the tiles [tile_base, total_tiles_all) of the batch: one group of problems, see slic_run_sweeps)
    const int gtile = tile_base + (((int)(blockIdx.x >> 3) / XG) * 8 + (int)(blockIdx.x & 7)) * XG + (int)(blockIdx.x >> 3) % XG;
    if (gtile >= total_tiles_all) return;
    // the tile's list state (scalar loads that depend on the block index alone: in flight beside the problem descriptor)
    // (meta: {entries | -1 no list yet | -2 not listed,  sweep of the build | short-lived builds in a row << 16})
    const int l_n = tl_meta[2 * (size_t)gtile], l_bw = tl_meta[2 * (size_t)gtile + 1], l_req = tl_req[gtile];
    const int l_built = l_bw < 0 ? -1 : (l_bw & 0xffff), l_streak = l_bw < 0 ? 0 : (l_bw >> 16);
    STAMP_DECL
    constexpr int RS = CENT_REC + CP;
    // (batches of equally sized problems -- the tiler's -- need no table look-up in front of the descriptor load: one
    // dependent scalar round trip less at the head of every workgroup)
    const int prob_i = tiles_per_prob > 0 ? gtile / tiles_per_prob : tile_prob[gtile];
    const SlicProblem P = probs[prob_i];
    const int tile = gtile - P.tile_off;
    constexpr int AQ = LEAN ? 1 : CP + 1;       // qwords of an LDS accumulator: colours (not in the lean kernel), then one packed word
    constexpr int PWI = LEAN ? 0 : CP;          // index of the packed word
                                                //   n | sum(y - ty0) << 16 | sum(x - tx0) << 40   (a 64x64 tile: n <= 4096 < 2^16,
                                                //   sums <= 4096 * 63 < 2^18 in fields of 24 bits: no field can carry into the next)
    constexpr int NPASS = (CP + 7) / 8;         // the transposed fold handles 8 colour fields per pass
    // fs: the fixed-point scale of the colour sums, a power of two.  It comes in as a float kernel argument (a scalar register): converted
    // from a double inside the kernel it sat in a vector register that the low-compactness kernel spilled and reloaded inside the
    // footprint loop (round 4)

    __shared__ __attribute__((aligned(16))) float s_hdr[MAXC][CENT_REC];
    // candidates sit in ascending k: slot = rank (the reference's tie rule, lowest k wins, becomes a comparison of slots: the slot
    // rides in the low word of the pixel keys and indexes the accumulators)
    __shared__ __attribute__((aligned(16))) float s_col[COLLB ? MAXC : 1][COLLB ? CP : 4];   // COLLB: colours of the staged candidates (scoring)
    __shared__ __attribute__((aligned(16))) int s_k[MAXC];   // slot -> k, ascending: a candidate's slot is its rank
    __shared__ unsigned char s_fls[NT / 64][64];             // list building: the footprint list of a wave before the lanes pick it up
    __shared__ unsigned long long s_acc[MAXC][AQ];
    // transposed scratch of the fold: [wave][layer = first / second run of a lane][field = colours, packed integer word][lane]
    constexpr int NF = LEAN ? 1 : CP + 1;
#ifndef OBIA_FOLD_LAYERS
#define OBIA_FOLD_LAYERS 1
#endif
    // runs of a lane's strip that go through the fold: the spatial-only pre-pass kernel is bound by its arithmetic, not by the
    // LDS -- its one packed word per run is added directly
    constexpr int FOLD_LAYERS = LEAN ? 0 : OBIA_FOLD_LAYERS;
    constexpr int FLD = FOLD_LAYERS > 0 ? FOLD_LAYERS : 1;
    __shared__ int s_tf[LEAN ? 1 : NT / 64][FLD][NF][LEAN ? 1 : 65];    // 65: row stride that keeps the transposed reads conflict-free
    __shared__ int s_tkey[LEAN ? 1 : NT / 64][FLD][LEAN ? 1 : 64];
    __shared__ unsigned s_orph[SWEEP_TH * SWEEP_TW / 32];   // valid pixels no window reached (rare): handled after the footprints
    __shared__ int s_cnt, s_uncacheable;   // s_cnt: slot counter of the build path
#ifdef OBIA_ABL_LDSPAD
    __shared__ int s_pad[OBIA_ABL_LDSPAD / 4];   // ablation build: LDS ballast that limits the workgroups per CU
    if (start_label == 12345) s_pad[threadIdx.x] = 1;
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: everything derived from it lives in scalar registers
    const int ty0 = (tile / P.tiles_x) * SWEEP_TH, tx0 = (tile % P.tiles_x) * SWEEP_TW;
    const int ty1 = min(ty0 + SWEEP_TH, P.H), tx1 = min(tx0 + SWEEP_TW, P.W);
    if (P.direct) {   // anisotropic `spacing`: every tile takes the direct path (same arithmetic with the scaled differences; the staged
                      // path's bounds and packed distance code assume unit spacing).  Workgroup-uniform.
        slow_tile<CP, MASKED, IGNORE_COLOR, SLICZERO>(P, ty0, tx0, feat, mask, cent, head, labels, acc, RQ, accumulate,
                                                      accum_color, start_label, fs, store_labels, orphan_flag, sweep_id, nch_rt);
        return;
    }

    // ---- wave geometry; the features of the wave's FIRST footprint are requested before staging, so their HBM
    // latency overlaps the dependent bin -> record loads of the staging phase ---------------------------------------
    const float w = P.spatial_w;
    const int fy0 = ty0 + FB * wv;
    const int fy0_o = fy0;
    const bool wave_active = fy0 < P.H;   // a wave below the bottom edge only helps with the final flush
    const int fy1 = min(fy0 + FB, P.H);
    const int fy1_o = fy1;
    const bool want_feat = !IGNORE_COLOR || accum_color;
    v2f f2[LEAN ? 1 : CP][PPT / 2];   // [channel][row pair]: pixels (yb, yb+1) and (yb+2, yb+3) of the lane's strip
    unsigned mbw;   // mask bytes of the lane's four pixels, packed (SlicBatch::d_mask4: ONE load, one register; turned into `valid` at the
                    // label stage: nothing waits for them earlier)
    // Addresses: one wave-uniform 64-bit base per footprint (scalar registers) plus a 32-bit lane offset -- a pixel's
    // address costs one or two vector instructions instead of a 64-bit multiply-add chain (16 rows x W x 32 B fits 32
    // bits: slic_run_sweeps limits W).  The mask bytes and the features of the four pixels are INDEPENDENT loads (all
    // twelve in flight at once; a masked pixel's features are fetched and never used): a feature load that waits for
    // its mask byte costs a second full memory round trip per pixel.
    // The loads of footprint b + 1 are issued in iteration b, right after the last use of the feature registers (the
    // centroid update) -- one call site inside the loop, so the registers are loop-carried without copies, and the
    // fold, the next scoring and the first selection run under the memory latency.  `real` = false (no next footprint):
    // every lane reads the first pixel of the current footprint -- one cache line, no branch around the loads.
    const bool all_valid = (long long)P.n_valid == (long long)P.H * (long long)P.W;   // wave-uniform (scalar registers)
    auto fetch = [&](int fx0, int yb, int lane_o, bool real) {   // (the lane's row comes in as an opaque per-footprint copy)
#ifdef OBIA_ABL_NOLOAD
        real = false;   // ablation build: every lane reads the footprint's first pixel (no HBM traffic for features / mask)
#endif
        const int xx = fx0 + (lane_o & 15);
        // a problem whose mask hides nothing (every interior tile of the tiler: n_valid == H * W, counted by
        // count_valid_kernel) reads no mask bytes (wave-uniform branch); the others read the packed mask: the four rows of the
        // lane's strip are the four bytes of one dword (rows past H hold zeros)
        if (MASKED && !all_valid) {
            const unsigned *mbase = mask4 + ((long long)P.m4_off + (long long)(fy0 >> 2) * P.W + fx0);      // wave-uniform
            const unsigned moff = (real && (yb < P.H) && (xx < P.W)) ? (unsigned)(lane_o >> 4) * (unsigned)P.W + (unsigned)(lane_o & 15) : 0u;
            mbw = mbase[moff];
        } else {
            mbw = 0x01010101u;
        }
        if (!LEAN && want_feat) {
            // quad-row blocks (slic.hpp): ONE 16-byte load per channel brings that channel of the lane's four pixels, the
            // channels are 256 bytes apart (immediate offsets), a quarter wave reads 256 contiguous bytes and the footprint's
            // quad row is one contiguous block (fy0 and fx0 are multiples of 16: the lane's strip is one quad row of one block)
            const float4 *pb = reinterpret_cast<const float4 *>(feat) +
                               (P.feat_off + ((long long)(fy0 >> 2) * P.XB + (fx0 >> 4)) * ((LEAN ? 0 : CP) * 16));   // wave-uniform
            const unsigned fob = (real && (yb < P.H) && (xx < P.W))
                                     ? ((unsigned)(lane_o >> 4) * (unsigned)((LEAN ? 0 : CP) * 16 * P.XB) + (unsigned)(lane_o & 15)) * 16u : 0u;
#pragma unroll
            for (int ch = 0; ch < (LEAN ? 1 : CP); ++ch) {
                if (ch >= nch_rt) { f2[ch][0] = splat(0.0f); f2[ch][1] = splat(0.0f); continue; }   // padded channel: zeros, not read (wave-uniform)
                // (non-temporal: a feature line is read by one wave, once per sweep -- common.hpp: ld_stream_f4)
                const float4 t = ld_stream_f4(reinterpret_cast<const char *>(pb + ch * 16) + (size_t)fob);
                f2[ch][0] = (v2f){t.x, t.y};
                f2[ch][1] = (v2f){t.z, t.w};
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < (LEAN ? 1 : CP); ++ch) { f2[ch][0] = splat(0.0f); f2[ch][1] = splat(0.0f); }
        }
    };
    // (first footprint: the lane's row and offset are rebuilt from the thread index at the call site -- computed once at the head of
    // the kernel they were spilled across the staging paths and reloaded, with a full wait, in front of the loads)
    auto fetch_first = [&]() {
        const int lane_i = lane_now();
        fetch(tx0, fy0 + PPT * (lane_i >> 4), lane_i, true);
    };
    constexpr int GQ = CP + 3;   // global record / cache entry: colours, n, sum_y, sum_x
    const int tile_id = P.tile_off + tile;
    // bins whose centroids can reach the tile: candidate <=> y0_k < ty1 && y1_k > ty0 (same in x); with
    // y0 = trunc(max(cy-2sy,0)), y1 = trunc(min(cy+2sy+1,H)) that needs cy in (ty0 - 2sy - 2, ty1 + 2sy + 1): one pixel
    // of slack covers float rounding of the binning.
    int by_lo = (ty0 - 2 * P.sy - 2) / P.sy; if (ty0 - 2 * P.sy - 2 < 0) by_lo = 0;
    int by_hi = (ty1 + 2 * P.sy + 1) / P.sy; if (by_hi > P.ncy - 1) by_hi = P.ncy - 1;
    int bx_lo = (tx0 - 2 * P.sx - 2) / P.sx; if (tx0 - 2 * P.sx - 2 < 0) bx_lo = 0;
    int bx_hi = (tx1 + 2 * P.sx + 1) / P.sx; if (bx_hi > P.ncx - 1) bx_hi = P.ncx - 1;
    const int nbw = bx_hi - bx_lo + 1;
    const int nbins = (by_hi - by_lo + 1) * nbw;
    if (FIXPT) {
        // exit_on_fixed_point: has any bin that can feed this tile been stamped since the tile was last evaluated?
        const int lp = use_cache ? tile_lp[tile_id] : 0;
        int dirty = lp <= 0;
        if (!dirty)
            for (int bi = tid; bi < nbins; bi += NT)
                dirty |= bin_stamp[P.cell_off + (by_lo + bi / nbw) * P.ncx + bx_lo + bi % nbw] > lp;
        if (!__syncthreads_or(dirty)) {
            // same candidate records as when the cache was written: same labels (already in place), same partial
            // sums -- replay them (cache_k[.][0] holds the slot count, -1 marks a slot nothing landed on)
            if (accumulate) {
                const int *ck = cache_k + (size_t)tile_id * (MAXC + 1);
                const unsigned long long *cq = cache_q + (size_t)tile_id * MAXC * GQ;
                const int ne = ck[0];
                for (int i = tid; i < ne * GQ; i += NT) {
                    const int e = i / GQ, q = i - e * GQ;
                    const int k = ck[1 + e];
                    if (k < 0 || (q < CP && !accum_color)) continue;
                    atomicAdd(&acc[(size_t)k * RQ + q], cq[(size_t)e * GQ + q]);
                }
            }
            return;
        }
        if (px_counter && tid == 0) atomicAdd(&px_counter[tile & 255], (unsigned long long)(ty1 - ty0) * (unsigned long long)(tx1 - tx0));
    } else if (px_counter && tile == 0 && tid == 0) {
        atomicAdd(px_counter, (unsigned long long)P.H * (unsigned long long)P.W);
    }
    // ---- 1. the candidates of the tile ("candidate lists" at the top of the file) ------------------------------------------------
    // listed: the tile reads its list -- ONE round trip to the centroid indices (issued at the head of the kernel, beside the
    // problem descriptor) and one to the headers; no bin walk, no slot counter, no ranking, one barrier.
    // The low-compactness kernel keeps no lists: where the colour term decides, centroids wander by pixels per sweep on noisy bands,
    // lists are rebuilt every other sweep and a build costs twenty uses (measured, compactness 0.25: sweep 0.294 ms per launch with
    // lists, 0.280 without, 0.278 in round 3 -- profiles/r04_notes.md).  Its tiles are staged from the bins in every sweep.
#ifdef OBIA_NO_LISTS   /* developer build: every sweep stages from the bins with the exact windows (A/B of the list caching) */
    constexpr bool USE_LISTS = false;
#else
    constexpr bool USE_LISTS = !COLLB;
#endif
    const bool listed = USE_LISTS && l_n >= 0 && l_req <= l_built;    // workgroup-uniform (scalar registers)
    unsigned myc4 = 0xffffffffu;   // the thread's list slot in each of its wave's four footprints, a byte each (0xff: none)
    auto clear_tables = [&]() {
        for (int i = tid; i < MAXC * AQ; i += NT) (&s_acc[0][0])[i] = 0ull;
        for (int i = tid; i < SWEEP_TH * SWEEP_TW / 32; i += NT) s_orph[i] = 0u;
        if (tid == 0) s_uncacheable = 0;
    };
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, rcol[CP / 4];
#pragma unroll
    for (int q = 0; q < CP / 4; ++q) rcol[q] = r0;
    int nxt = -1;
    auto load_node = [&](int c) {   // header, link (and colours for the colour-box bound) of one centroid record: one round trip
        const float4 *src = reinterpret_cast<const float4 *>(cent + (size_t)c * RS);
        r0 = src[0]; r1 = src[1];
        if (COLLB) {
#pragma unroll
            for (int q = 0; q < CP / 4; ++q) rcol[q] = src[2 + q];
        }
        nxt = __float_as_int(r1.z);   // the link rides in the record
    };
    auto put_slot = [&](int slot, int k) {   // the staged copy of a record: header (slot 6: k), k table, colours
        float4 *dh = reinterpret_cast<float4 *>(&s_hdr[slot][0]);
        dh[0] = r0; dh[1] = make_float4(r1.x, r1.y, __int_as_float(k), r1.w);
        s_k[slot] = k;
        if (COLLB) {
#pragma unroll
            for (int q = 0; q < CP / 4; ++q) *reinterpret_cast<float4 *>(&s_col[slot][4 * q]) = rcol[q];
        }
    };
    int nc;
    if (listed) {
        // (the list loads live inside this branch: issued at the head of the kernel their results were spilled across the other one)
        int lk = -1;
        if (tid < l_n) lk = tl_k[(size_t)gtile * MAXC + tid];
        myc4 = tl_fp[(size_t)gtile * NT + tid];
        clear_tables();   // under the latency of the list
        if (lk >= 0) load_node(lk);
        // (the features of the wave's FIRST footprint: requested behind the headers, so that the staging below waits for the
        // headers only; their HBM latency runs under the staging and the scoring)
        if (wave_active) fetch_first();
        if (lk >= 0) put_slot(tid, lk);
        nc = l_n;
        __syncthreads();
        STAMP(1)   // prologue + staging (up to its barrier)
    } else {
        // build: lanes walk the lists of the bins whose centroids can reach the tile -- candidates are the centroids whose
        // window, grown by g, meets it (g = LIST_G2: the superset that is kept as the tile's list; g = 0: the exact set, for
        // a tile whose superset does not fit) -- rank them by k, put them in that order and make the footprint lists.
        clear_tables();
        // A list that is rebuilt in the sweep after it was built was never used: where centroids keep moving by pixels per sweep
        // (low compactness on noisy bands) building lists costs more than the bin walk with the exact windows.  Two such builds
        // in a row and the tile stays unlisted for the rest of the batch (it is staged like in rounds 1-3).
        const int streak = (l_n >= 0 && sweep_id - l_built <= 1) ? l_streak + 1 : 0;
        int g = (!USE_LISTS || l_n == -2 || streak >= LIST_GIVE_UP || sweep_id >= 0xffff) ? 0 : LIST_G2;
        bool listable = g > 0;
        for (;;) {   // (second round with g = 0 when the superset overflows: workgroup-uniform)
            __syncthreads();
            if (tid < MAXC) s_k[tid] = 0x7fffffff;   // sentinel: the rank loop reads whole groups of eight entries without bound checks
            if (tid == 0) s_cnt = 0;
            __syncthreads();
            // bins whose centroids can reach the tile: candidate <=> y0_k - g < ty1 && y1_k + g > ty0 (same in x); with
            // y0 = trunc(max(cy-2sy,0)), y1 = trunc(min(cy+2sy+1,H)) that needs cy in (ty0 - 2sy - 2 - g, ty1 + 2sy + 1 + g): one
            // pixel of slack covers float rounding of the binning.
            int gy_lo = (ty0 - 2 * P.sy - 2 - g) / P.sy; if (ty0 - 2 * P.sy - 2 - g < 0) gy_lo = 0;
            int gy_hi = (ty1 + 2 * P.sy + 1 + g) / P.sy; if (gy_hi > P.ncy - 1) gy_hi = P.ncy - 1;
            int gx_lo = (tx0 - 2 * P.sx - 2 - g) / P.sx; if (tx0 - 2 * P.sx - 2 - g < 0) gx_lo = 0;
            int gx_hi = (tx1 + 2 * P.sx + 1 + g) / P.sx; if (gx_hi > P.ncx - 1) gx_hi = P.ncx - 1;
            const int gbw = gx_hi - gx_lo + 1;
            const int gbins = (gy_hi - gy_lo + 1) * gbw;
            for (int bi = tid; bi < gbins; bi += NT) {
                int cur = head[P.cell_off + (gy_lo + bi / gbw) * P.ncx + gx_lo + bi % gbw];
                if (cur >= 0) load_node(cur);
                while (cur >= 0) {
                    const int y0 = __float_as_int(r0.z), y1 = __float_as_int(r0.w);
                    const int x0 = __float_as_int(r1.x), x1 = __float_as_int(r1.y);
                    if (y0 - g < ty1 && y1 + g > ty0 && x0 - g < tx1 && x1 + g > tx0) {
                        const int slot = atomicAdd(&s_cnt, 1);
                        if (slot < MAXC) put_slot(slot, cur);
                    }
                    cur = nxt;
                    if (cur >= 0) load_node(cur);
                }
            }
            __syncthreads();
            nc = s_cnt;
            if (nc > MAXC) {   // workgroup-uniform
                if (g > 0) { g = 0; listable = false; continue; }
                if (tid == 0) { tl_meta[2 * (size_t)gtile] = -2; tl_meta[2 * (size_t)gtile + 1] = sweep_id & 0xffff; }
                slow_tile<CP, MASKED, IGNORE_COLOR, SLICZERO>(P, ty0, tx0, feat, mask, cent, head, labels, acc, RQ, accumulate,
                                                              accum_color, start_label, fs, store_labels, orphan_flag, sweep_id, nch_rt);
                return;
            }
            // rank of every slot = number of staged candidates with a smaller k (all distinct).  Thread t ranks slot t against the
            // whole k table with broadcast reads of four entries each -- independent LDS reads, no serial chain -- and moves its
            // record to the slot of its rank.
            {
                int myk = 0, rk = 0;
                if (tid < nc) {
                    myk = s_k[tid];
#pragma unroll 1
                    for (int i = 0; i < nc; i += 8) {   // eight entries per step: two independent reads in flight (unstaged slots hold INT_MAX)
                        const int4 ka = *reinterpret_cast<const int4 *>(&s_k[i]), kb = *reinterpret_cast<const int4 *>(&s_k[i + 4]);
                        rk += (ka.x < myk) + (ka.y < myk) + (ka.z < myk) + (ka.w < myk) + (kb.x < myk) + (kb.y < myk) + (kb.z < myk) + (kb.w < myk);
                    }
                    r0 = *reinterpret_cast<const float4 *>(&s_hdr[tid][0]);
                    r1 = *reinterpret_cast<const float4 *>(&s_hdr[tid][4]);
                    if (COLLB) {
#pragma unroll
                        for (int q = 0; q < CP / 4; ++q) rcol[q] = *reinterpret_cast<const float4 *>(&s_col[tid][4 * q]);
                    }
                }
                __syncthreads();
                if (tid < nc) put_slot(rk, myk);
                __syncthreads();
            }
            // footprint lists: the candidates whose (grown) window meets each of the wave's four footprints, at most 64 of them.
            // A tile with at most 64 candidates needs none: lane l scores slot l in every footprint (the exact window test of
            // the scoring sorts out the rest) -- the usual case of a tile that is staged from the bins in every sweep.
            bool ovf = false;
            myc4 = (nc <= 64 && lane < nc) ? (unsigned)lane * 0x01010101u : 0xffffffffu;
            for (int bxi = 0; nc > 64 && wave_active && bxi < SWEEP_TW / FB; ++bxi) {
                const int fx0 = tx0 + FB * bxi;
                if (fx0 >= P.W) break;   // wave-uniform
                const int fx1 = min(fx0 + FB, P.W);
                int base = 0;
                for (int r = 0; r < (MAXC + 63) / 64; ++r) {
                    const int c = 64 * r + lane;
                    bool in = false;
                    if (c < nc) {
                        const float4 h0 = *reinterpret_cast<const float4 *>(&s_hdr[c][0]);
                        const float4 h1 = *reinterpret_cast<const float4 *>(&s_hdr[c][4]);
                        in = (__float_as_int(h0.z) - g < fy1) && (__float_as_int(h0.w) + g > fy0) &&
                             (__float_as_int(h1.x) - g < fx1) && (__float_as_int(h1.y) + g > fx0);
                    }
                    const unsigned long long m = __ballot(in);
                    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                    if (in && pos < 64) s_fls[wv][pos] = (unsigned char)c;
                    base += __popcll(m);
                }
                ovf |= base > 64;
                wave_lds_sync();
                const unsigned mine = (lane < base) ? (unsigned)s_fls[wv][lane] : 0xffu;
                myc4 = (myc4 & ~(0xffu << (8 * bxi))) | (mine << (8 * bxi));
                wave_lds_sync();
            }
            if (__syncthreads_or(ovf ? 1 : 0)) {   // (a footprint meets more than 64 candidates: clustered centroids)
                if (g > 0) { g = 0; listable = false; continue; }
                if (tid == 0) { tl_meta[2 * (size_t)gtile] = -2; tl_meta[2 * (size_t)gtile + 1] = sweep_id & 0xffff; }
                slow_tile<CP, MASKED, IGNORE_COLOR, SLICZERO>(P, ty0, tx0, feat, mask, cent, head, labels, acc, RQ, accumulate,
                                                              accum_color, start_label, fs, store_labels, orphan_flag, sweep_id, nch_rt);
                return;
            }
            break;
        }
        if (listable) {
            if (tid < nc) tl_k[(size_t)gtile * MAXC + tid] = s_k[tid];
            tl_fp[(size_t)gtile * NT + tid] = myc4;
        }
        if (tid == 0) { tl_meta[2 * (size_t)gtile] = listable ? nc : -2; tl_meta[2 * (size_t)gtile + 1] = (sweep_id & 0xffff) | (streak << 16); }
        // (the features of the first footprint are requested only now: registers that are live across the build path -- the
        // fallback to slow_tile() sits in it -- were spilled in the footprint loop)
        if (wave_active) fetch_first();
        STAMP(1)   // prologue + staging + ranking + lists
    }
    STAMP(0)   // (sort: part of the build path since round 4)

    constexpr unsigned INF_BITS = 0x7f800000u;
    // ---- 2. per wave: four 16x16 footprints (one 16-row band of the 64x64 tile) ---------------------------------------
    for (int bxi = 0; wave_active && bxi < SWEEP_TW / FB; ++bxi) {
        const int fx0 = tx0 + FB * bxi;
        if (fx0 >= P.W) break;   // wave-uniform
        // values derived from the lane's rows are the same in all four footprints; hoisted out of this loop they would sit in
        // ~25 registers for the whole kernel (and spill): opaque copies keep them one or two instructions away instead
        // (rebuilt from the thread index, the one vector register that is live anyway: kept across the loop, lane, yb and lrow
        // would hold three registers for the whole kernel)
        const int lane_i = lane_now();
        const int yb_i = fy0 + PPT * (lane_i >> 4);
        const unsigned lrow_i = (unsigned)(PPT * (lane_i >> 4)) * (unsigned)P.W + (unsigned)(lane_i & 15);
        const int fx1 = min(fx0 + FB, P.W);
        const int x = fx0 + (lane_i & 15);
        const float fx = (float)x;
        // pixel state: key = float_bits(best distance) << 32 | slot of the best candidate.  Every pixel starts at
        // (+inf, 0): nothing with d = inf is ever smaller (the reference's `inf > inf` never assigns), and "assigned"
        // is d < inf.  Masked / outside pixels are evaluated like the others (their features are zeros or unused values)
        // and discarded at the label stage: the visits never wait for the mask bytes.
        unsigned long long bk[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) bk[j] = (unsigned long long)INF_BITS << 32;
#define BK_D(j) __uint_as_float((unsigned)(bk[j] >> 32))

        // ---- score the footprint's candidates, one per lane (the footprint list holds at most 64) ----------------------
        // lb = the reference's spatial expression evaluated at the footprint point nearest to the centroid: every
        // operation is monotone, so lb <= spatial(pixel) <= d(pixel) for every pixel of the footprint.  The scoring key
        // is lb with its low 7 mantissa bits replaced by the slot number: still a lower bound (rounded DOWN), unique in
        // the wave, and the wave minimum names its slot without a ballot.
        unsigned key = 0xffffffffu;
        int kkv = 0;         // global centroid index of the lane's candidate
        float clbv = 0.0f;   // COLLB: its colour-box bound
        float blo[COLLB ? CP : 1], bhi[COLLB ? CP : 1];
        if (COLLB) {   // the box of this footprint: wave-uniform address, scalar loads
            const float4 *bx = reinterpret_cast<const float4 *>(fbox) + (P.fb_off + (long long)(fy0_o >> 4) * P.XB + (fx0 >> 4)) * (2 * CP / 4);
#pragma unroll
            for (int q = 0; q < CP / 4; ++q) {
                const float4 l = bx[q], h = bx[CP / 4 + q];
                blo[4 * q] = l.x; blo[4 * q + 1] = l.y; blo[4 * q + 2] = l.z; blo[4 * q + 3] = l.w;
                bhi[4 * q] = h.x; bhi[4 * q + 1] = h.y; bhi[4 * q + 2] = h.z; bhi[4 * q + 3] = h.w;
            }
        }
        {
            const unsigned c = (myc4 >> (8 * bxi)) & 0xffu;   // the lane's slot in this footprint's list
            // (the float images of the footprint's edges are rebuilt here from scalar registers: kept across the
            // footprint loop they cost vector registers the visit loop needs)
            int fy0 = fy0_o, fy1 = fy1_o;
            asm volatile("" : "+s"(fy0), "+s"(fy1));
            if (c != 0xffu) {
                const float4 h0 = *reinterpret_cast<const float4 *>(&s_hdr[c][0]);
                const float4 h1 = *reinterpret_cast<const float4 *>(&s_hdr[c][4]);
                kkv = __float_as_int(h1.z);
                const int y0 = __float_as_int(h0.z), y1 = __float_as_int(h0.w);
                const int x0 = __float_as_int(h1.x), x1 = __float_as_int(h1.y);
                if (y0 < fy1 && y1 > fy0 && x0 < fx1 && x1 > fx0) {   // the exact window of the fresh header
                    const float cy = h0.x, cx = h0.y;
                    const float ry = (cy < (float)fy0) ? (float)fy0 : ((cy > (float)(fy1 - 1)) ? (float)(fy1 - 1) : cy);
                    const float rx = (cx < (float)fx0) ? (float)fx0 : ((cx > (float)(fx1 - 1)) ? (float)(fx1 - 1) : cx);
                    const float tyv = cy - ry, txv = cx - rx;
                    float lb = (tyv * tyv + txv * txv) * w;
                    if (COLLB) {
                        // dc = sum over channels, in channel order, of (f - c)^2 >= the same sum of max(lo - c, c - hi, 0)^2
                        float cl = 0.0f;
#pragma unroll
                        for (int q = 0; q < CP / 4; ++q) {
                            const float4 cc = *reinterpret_cast<const float4 *>(&s_col[c][4 * q]);
                            const float cv[4] = {cc.x, cc.y, cc.z, cc.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float t = fmaxf(fmaxf(blo[4 * q + e] - cv[e], cv[e] - bhi[4 * q + e]), 0.0f);
                                cl += t * t;
                            }
                        }
                        clbv = cl;
                        lb += cl;
                    }
                    key = (__float_as_uint(lb) & ~127u) | c;      // lb >= 0, never NaN: below 0xffffffff
                }
            }
        }

        STAMP(2)   // scoring
        STAMP_COUNT(0, 1)   // footprints
        // ---- visit candidates in ascending lb until lb exceeds every lane's current best ---------------------------
        // a lane's largest current best distance (+inf while one of its valid pixels is unassigned, 0 when it has no
        // valid pixel): the walk stops when lb exceeds it in every lane -- one compare and a ballot, no wave reduction.
        // The kernel is bound by VALU issue first and by the dependent chain of a single wave second (a wave alone on its
        // SIMD still needs 2/3 of the time it needs with five neighbours), so an iteration is one straight-line block with
        // as few vector instructions as possible: the record is requested by the scalar unit first, the wave minimum that
        // names the NEXT candidate is computed under that latency, the per-pixel window test runs only when the window
        // does not cover the footprint, and the two row pairs run as two interleaved packed chains.
        float mybest = fmaxf(fmaxf(BK_D(0), BK_D(1)), fmaxf(BK_D(2), BK_D(3)));
#ifdef OBIA_ABL_VISITS
        int abl_visits = 0;
#endif
        unsigned mn = wave_umin(key);
        // (wave-uniform) the first visit of a footprint meets pixels that all hold (+inf, 0): no stop test, no live test, and the
        // key update is "finite distance wins" -- one 32-bit compare instead of a 64-bit one per pixel (round 4: colour sweep -0.6 %,
        // pre-pass -2.7 %, low-compactness sweep -1.5 %)
        bool firstv = true;
        for (;;) {
            if (mn == 0xffffffffu) break;
            if (!firstv && !__ballot(__uint_as_float(mn & ~127u) <= mybest)) break;   // equality must still be visited: it can tie on k
#ifdef OBIA_ABL_VISITS
            if (++abl_visits > OBIA_ABL_VISITS) break;   // ablation build: at most this many visits per footprint
#endif
            const unsigned c = mn & 127u;   // rank of the candidate: low word of the pixel keys
            // the scoring lane that holds the minimum names the candidate: its record is read from global memory at a
            // wave-uniform address -- the scalar unit loads header and colours into SGPRs, no VALU / LDS work (the kernel is
            // bound by VALU issue: every vector instruction of this loop is paid 4.4 times per footprint)
            const int sl = (int)__builtin_ctzll(__ballot(key == mn));   // (the keys are unique in the wave)
            const int kk = __builtin_amdgcn_readlane(kkv, sl);
            key = (key == mn) ? 0xffffffffu : key;
            float clb = 0.0f;   // COLLB: the candidate's colour-box bound (uniform)
            if (COLLB) clb = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(clbv), sl));
            const float4 *__restrict__ crec = reinterpret_cast<const float4 *>(cent + (size_t)kk * RS);
            const float4 h0 = crec[0], h1 = crec[1];
            // the next candidate, under the latency of the loads
            const unsigned mn_next = wave_umin(key);
            const float cy = h0.x, cx = h0.y;
            const int y0 = __float_as_int(h0.z), y1 = __float_as_int(h0.w);
            const int x0 = __float_as_int(h1.x), x1 = __float_as_int(h1.y);
            const float mdc = SLICZERO ? h1.w : 1.0f;
            mn = mn_next;
            STAMP(8)   // visit: record + next minimum
            const float tx = cx - fx;
            const float dx2 = tx * tx;
            // rows of the strip as floats, rebuilt per visit from the one integer register that is live anyway (row numbers
            // are far below 2^24: float(yb) + j is exact)
            int yb_v = yb_i;
            asm volatile("" : "+v"(yb_v));
            const float fyb = (float)yb_v;
            v2f dv2[PPT / 2];
#pragma unroll
            for (int p = 0; p < PPT / 2; ++p) {   // two pixels per instruction; each component rounds like the scalar operation
                const v2f tyv = splat(cy) - (splat(fyb) + (v2f){(float)(2 * p), (float)(2 * p + 1)});
                const v2f dy2 = tyv * tyv;
                dv2[p] = (dy2 + splat(dx2)) * splat(w);           // (dz + dy + dx) * spatial_weight, dz = 0
            }
            float dv[PPT] = {dv2[0].x, dv2[0].y, dv2[1].x, dv2[1].y};
            // a window that covers the whole footprint needs no per-pixel test; otherwise a pixel outside the window gets the
            // spatial term +inf: `inf < best` never holds, it cannot win
            if (!((y0 <= fy0) && (y1 >= fy1) && (x0 <= fx0) && (x1 >= fx1))) {   // wave-uniform, decided on the scalar unit
                const bool inx = (x >= x0) && (x < x1);
#pragma unroll
                for (int j = 0; j < PPT; ++j)
                    dv[j] = (inx && ((unsigned)(yb_v + j - y0) < (unsigned)(y1 - y0))) ? dv[j] : INFINITY;
            }
            // colour >= 0 and float add is monotone, so d >= spatial: a candidate whose spatial part already exceeds the best
            // distance of a pixel cannot win it (equality could still tie on k): no such pixel in the wave -> no colours
            // (COLLB: d = spatial + dc >= spatial + colour bound, both sums rounded the same way)
            const bool anylive = COLLB ? (!(dv[0] + clb > BK_D(0)) || !(dv[1] + clb > BK_D(1)) || !(dv[2] + clb > BK_D(2)) || !(dv[3] + clb > BK_D(3)))
                                       : (!(dv[0] > BK_D(0)) || !(dv[1] > BK_D(1)) || !(dv[2] > BK_D(2)) || !(dv[3] > BK_D(3)));
            STAMP_COUNT(1, 1)   // visits
            STAMP(9)   // visit: spatial + live test
            if (!firstv && !__ballot(anylive)) continue;
            STAMP_COUNT(2, 1)   // visits that evaluate colours
            STAMP_COUNT(3, 2)
            {
                float col[LEAN ? 1 : CP];
                if (!IGNORE_COLOR) {
#pragma unroll
                    for (int q = 0; q < (NCH + 3) / 4; ++q) {
                        const float4 t = crec[2 + q];   // wave-uniform address: scalar loads, the colours stay in SGPRs
                        col[4 * q] = t.x; col[4 * q + 1] = t.y; col[4 * q + 2] = t.z; col[4 * q + 3] = t.w;
                    }
                }
                v2f d0 = (v2f){dv[0], dv[1]}, d1 = (v2f){dv[2], dv[3]};
                if (!IGNORE_COLOR) {
                    // dc = 0; dc += t*t per channel, in channel order: 0 + t*t == t*t exactly (t*t is never -0), the other
                    // additions stay sequential.  Packed f32: both pixels of a pair in one v_pk_add / v_pk_mul, each
                    // component rounded like the scalar instruction (no contraction: -ffp-contract=off); the two pairs
                    // are independent chains in one block.
                    v2f t0 = f2[0][0] - splat(col[0]), t1 = f2[0][1] - splat(col[0]);
                    v2f dc0 = t0 * t0, dc1 = t1 * t1;
#pragma unroll
                    for (int ch = 1; ch < (LEAN ? 1 : NCH); ++ch) {   // (channels >= NCH: 0 - 0, squared, added: +0)
                        t0 = f2[LEAN ? 0 : ch][0] - splat(col[LEAN ? 0 : ch]);
                        t1 = f2[LEAN ? 0 : ch][1] - splat(col[LEAN ? 0 : ch]);
                        dc0 += t0 * t0;
                        dc1 += t1 * t1;
                    }
                    // SLIC-zero: the colour term is scaled by the largest colour distance seen in this cluster so far
                    // (_slic.pyx: dist_center += dist_color / max_dist_color[k])
                    d0 += SLICZERO ? dc0 / splat(mdc) : dc0;
                    d1 += SLICZERO ? dc1 / splat(mdc) : dc1;
                }
                // reference: ascending k with strict `distance > d`  ==  lexicographic min of (d, k)  ==  min of the keys
                const float dd[PPT] = {d0.x, d0.y, d1.x, d1.y};
                if (firstv) {
#pragma unroll
                    for (int j = 0; j < PPT; ++j) {
                        const unsigned long long nk = ((unsigned long long)__float_as_uint(dd[j]) << 32) | (unsigned long long)c;
                        bk[j] = (dd[j] < INFINITY) ? nk : bk[j];      // (+inf, 0) loses to every finite distance and to nothing else
                    }
                    firstv = false;
                } else
#pragma unroll
                for (int j = 0; j < PPT; ++j) {
                    const unsigned long long nk = ((unsigned long long)__float_as_uint(dd[j]) << 32) | (unsigned long long)c;
                    bk[j] = (nk < bk[j]) ? nk : bk[j];
                }
            }
            mybest = fmaxf(fmaxf(BK_D(0), BK_D(1)), fmaxf(BK_D(2), BK_D(3)));
            STAMP(10)   // visit: colours + keys
        }
        STAMP(3)   // visits (loop control, first minimum)

        // ---- labels ---------------------------------------------------------------------------------------------------
        int pk[PPT];   // accumulation key: LDS slot, or -1
        int32_t *lbase = labels + (P.pix_off + (long long)fy0 * P.W + fx0);   // wave-uniform
        bool orphan = false;
        // Only the labels of the LAST sweep are an output.  The labels of the other sweeps are read by nothing but the rare
        // "orphan" pixels below, so those sweeps do not store them (store_labels = 0: 4 of the 36 bytes per pixel, four
        // stores and four LDS reads per lane) -- an orphan then raises a flag and the host repeats the batch with every
        // sweep storing (slic_run_sweeps): the result is the same either way.
        bool valid[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const bool inimg = (yb_i + j < P.H) && (x < P.W);
            valid[j] = inimg && (((mbw >> (8 * j)) & 0xffu) != 0u);
            const bool assigned = valid[j] && ((unsigned)(bk[j] >> 32) < INF_BITS);
            pk[j] = assigned ? (int)(unsigned)bk[j] : -1;
            // a valid pixel no window reaches keeps the previous sweep's label (`nearest` is only initialised once,
            // before the loop): nothing is stored, the pixel is noted in the tile's bitmap and accumulated under its old
            // label after the footprints (rare: centroids that drifted away from a thin piece of the mask)
            orphan |= valid[j] && !assigned;
        }
        if (store_labels) {   // kernel argument: wave-uniform
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const bool inimg = (yb_i + j < P.H) && (x < P.W);
                const int kk = s_k[(unsigned)bk[j]];   // slot -> k (slot 0 while unassigned: unused)
                if (inimg && !(valid[j] && pk[j] < 0))
                    __builtin_nontemporal_store(pk[j] >= 0 ? kk - P.cent_off + start_label : start_label - 1,
                                                lbase + (lrow_i + (unsigned)j * (unsigned)P.W));
            }
        }
        if (__ballot(orphan)) {   // wave-uniform, rare
            if (orphan_needs_repeat(store_labels, sweep_id) && lane_i == 0) *orphan_flag = 1;   // the labels this pixel would keep were not stored: repeat the batch
#pragma unroll
            for (int j = 0; j < PPT; ++j)
                if (valid[j] && ((unsigned)(bk[j] >> 32) >= INF_BITS)) {
                    const int ry = yb_i + j - ty0, rx = x - tx0;
                    atomicOr(&s_orph[ry * (SWEEP_TW / 32) + (rx >> 5)], 1u << (rx & 31));
                    s_uncacheable = 1;   // the tile's result depends on the previous labels: never replay it (exit_on_fixed_point)
                }
        }
        STAMP(4)   // labels

        // ---- 3. fused centroid update ------------------------------------------------------------------------------------
        // LDS atomics on one address are executed one lane after the other, and a footprint only holds three or four slots:
        // sixty-four lanes adding their sums directly cost 11 % of the sweep (measured: OBIA_FOLD_LAYERS = 0 vs 1).  So the
        // FIRST run of a lane's strip -- three strips out of four hold one run only -- is written to a transposed scratch,
        // colours as 32-bit fixed point, the integer part packed as  n | sum(y - ty0) << 8 | n * (x - tx0) << 20,  and a few
        // fold lanes walk it sequentially, adding a partial to the LDS accumulators only where the slot changes.  Later runs
        // (a strip crossing a segment boundary) add directly: a second scratch layer for them was measured 1 % slower.
        if (accumulate) {
        if (FOLD_LAYERS > 0) s_tkey[wv][0][lane_i] = -1;
        if (FOLD_LAYERS > 1) s_tkey[wv][1][lane_i] = -1;
        {
            const unsigned xrel = (unsigned)(x - tx0);
            int rkey = -1, nruns = 0;
            unsigned pw = 0;     // packed integer word of the run
            int rf[LEAN ? 1 : CP];
#pragma unroll
            for (int ch = 0; ch < (LEAN ? 1 : CP); ++ch) rf[ch] = 0;
            auto close_run = [&]() {
                if (rkey < 0) return;
                if (nruns < FOLD_LAYERS) {   // the lane's column in layer `nruns` of the transposed scratch
                    s_tkey[wv][nruns][lane_i] = rkey;
                    s_tf[wv][nruns][NF - 1][lane_i] = (int)pw;
                    if (!LEAN && accum_color) {
#pragma unroll
                        for (int ch = 0; ch < CP; ++ch) s_tf[wv][nruns][LEAN ? 0 : ch][lane_i] = rf[LEAN ? 0 : ch];
                    }
                } else {
                    atomicAdd(&s_acc[rkey][PWI], (unsigned long long)(pw & 0xffu) | ((unsigned long long)((pw >> 8) & 0xfffu) << 16) |
                                                     ((unsigned long long)(pw >> 20) << 40));
                    if (!LEAN && accum_color) {
#pragma unroll
                        for (int ch = 0; ch < CP; ++ch) atomicAdd(&s_acc[rkey][LEAN ? 0 : ch], (unsigned long long)(long long)rf[LEAN ? 0 : ch]);
                    }
                }
                ++nruns;
            };
            // PRE: the feature planes already carry the fixed-point scale (slic_prescale: a power of two folded into the planes, the
            // spatial weight and the centroid colours -- every distance is the reference's times that power, every comparison the
            // same), so the conversion is the truncation alone: 32 multiplications per lane and footprint less (round 4)
            auto merge_strip = [&](auto pre_tag) {
                constexpr bool PRE = decltype(pre_tag)::value;
#pragma unroll
                for (int j = 0; j < PPT; ++j) {
                    if (pk[j] != rkey) {
                        close_run();
                        rkey = pk[j]; pw = 0;
#pragma unroll
                        for (int ch = 0; ch < (LEAN ? 1 : CP); ++ch) rf[ch] = 0;
                    }
                    if (pk[j] >= 0) {
                        pw += 1u | ((unsigned)(yb_i + j - ty0) << 8) | (xrel << 20);
                        if (!LEAN && accum_color) {
#pragma unroll
                            for (int ch = 0; ch < NCH; ++ch) {   // (padded channels stay 0)
                                const float fv = (j & 1) ? f2[LEAN ? 0 : ch][j >> 1].y : f2[LEAN ? 0 : ch][j >> 1].x;
                                rf[LEAN ? 0 : ch] += PRE ? (int)fv : to_fixed32(fv, fs);
                            }
                        }
                    }
                }
                close_run();
            };
            if (fs == 1.0f) merge_strip(std::true_type{});   // (kernel argument: a scalar branch)
            else merge_strip(std::false_type{});
        }
        }
        STAMP(5)   // run merge
        {   // the feature registers are free: request the next footprint (the ONLY call site inside the loop)
            const bool has_next = (bxi + 1 < SWEEP_TW / FB) && (fx0 + FB < P.W);   // wave-uniform
            fetch(has_next ? fx0 + FB : fx0, yb_i, lane_i, has_next);
        }
        if (!accumulate || FOLD_LAYERS == 0) continue;
        // transposed fold.  Colours: lane (fld, g) walks the strips 8g .. 8g+7 of colour field fld, layer by layer.  The packed
        // words of both layers are walked by sixteen lanes (layer, g).  The sub-fields of a packed word cannot carry into each
        // other over 8 strips (n <= 32 < 2^8, coordinate sums <= 8 * 252 < 2^12): it is summed as one integer and unpacked
        // into the 64-bit format of the LDS accumulators when a partial is added.
        wave_lds_sync();
        if (!LEAN && accum_color) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int fld = 8 * pass + (lane_i & 7), g = lane_i >> 3;
                if (fld < CP) {
#pragma unroll
                    for (int layer = 0; layer < FOLD_LAYERS; ++layer) {
                        int cur = -1;
                        long long sum = 0;
                        // all sixteen reads first (independent addresses): one LDS round trip, not eight behind the atomics
                        int tkv[8], vv[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) { tkv[i] = s_tkey[wv][layer][8 * g + i]; vv[i] = s_tf[wv][layer][LEAN ? 0 : fld][8 * g + i]; }
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int tk = tkv[i];
                            const int v = vv[i];   // stale where tk < 0: never added
                            if (tk != cur) {
                                if (cur >= 0) atomicAdd(&s_acc[cur][LEAN ? 0 : fld], (unsigned long long)sum);
                                cur = tk; sum = 0;
                            }
                            sum += (long long)v;
                        }
                        if (cur >= 0) atomicAdd(&s_acc[cur][LEAN ? 0 : fld], (unsigned long long)sum);
                    }
                }
            }
        }
        if (lane_i < 8 * FOLD_LAYERS) {
            const int layer = lane_i >> 3, g = lane_i & 7;
            int cur = -1;
            unsigned sum = 0;
            auto emit = [&]() {
                atomicAdd(&s_acc[cur][PWI], (unsigned long long)(sum & 0xffu) | ((unsigned long long)((sum >> 8) & 0xfffu) << 16) |
                                                ((unsigned long long)(sum >> 20) << 40));
            };
            int tkv[8];
            unsigned vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { tkv[i] = s_tkey[wv][layer][8 * g + i]; vv[i] = (unsigned)s_tf[wv][layer][NF - 1][8 * g + i]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int tk = tkv[i];
                const unsigned v = vv[i];
                if (tk != cur) {
                    if (cur >= 0) emit();
                    cur = tk; sum = 0;
                }
                sum += v;
            }
            if (cur >= 0) emit();
        }
        wave_lds_sync();   // the scratch is rewritten by the next footprint
        STAMP(6)   // fold
    }
#undef BK_D
    if (!accumulate) { STAMP_FLUSH return; }
    __syncthreads();
    const int tid_e = wv * 64 + lane_now();   // (the thread index, rebuilt: see lane_now)
    if (s_uncacheable && store_labels == 2) {   // workgroup-uniform, rare: the orphan pixels of the tile go straight to the global records (their previous labels are in memory only when every sweep stores)
        for (int i = tid_e; i < SWEEP_TH * SWEEP_TW; i += NT) {
            if (!((s_orph[i >> 5] >> (i & 31)) & 1u)) continue;
            const int y = ty0 + i / SWEEP_TW, x = tx0 + i % SWEEP_TW;
            const long long pix = P.pix_off + (long long)y * P.W + x;
            const int prev = labels[pix];
            if (prev < start_label) continue;
            float one[CP];
#pragma unroll
            for (int ch = 0; ch < CP; ++ch) one[ch] = (accum_color && ch < nch_rt) ? feat_at(feat, P, CP, y, x, ch) : 0.0f;
            global_accumulate<CP>(acc, RQ, prev - start_label + P.cent_off, (unsigned)y, (unsigned)x, one, fs);
        }
    }
    // ---- LDS accumulators -> global records: consecutive lanes write consecutive qwords of one 128-B record -----------
    // (with exit_on_fixed_point the same values are kept, slot by slot, as the tile's cache for the sweeps that replay them)
    const bool keep = FIXPT && !s_uncacheable;
    int *ck = keep ? cache_k + (size_t)tile_id * (MAXC + 1) : nullptr;
    unsigned long long *cq = keep ? cache_q + (size_t)tile_id * MAXC * GQ : nullptr;
    // (the lean kernel without the fixed-point cache only has the three integer words of every record to send)
    constexpr int QLO = (LEAN && !FIXPT) ? CP : 0, QN = GQ - QLO;
    for (int i = tid_e; i < nc * QN; i += NT) {
        const int slot = i / QN, q = QLO + (i - slot * QN);
        const unsigned long long pw = s_acc[slot][PWI];
        const unsigned long long n = pw & 0xffffull;
        if (FIXPT && keep && q == QLO) ck[1 + slot] = n ? s_k[slot] : -1;
        if (n == 0ull) continue;   // nothing landed on this centroid
        const int k = s_k[slot];
        unsigned long long v;
        if (q < CP) { if (!FIXPT && !accum_color) continue; v = accum_color ? s_acc[slot][LEAN ? 0 : q] : 0ull; }
        else if (q == CP) v = n;
        else if (q == CP + 1) v = ((pw >> 16) & 0xffffffull) + n * (unsigned long long)ty0;
        else v = (pw >> 40) + n * (unsigned long long)tx0;
        if (FIXPT && keep) cq[(size_t)slot * GQ + q] = v;
        if (FIXPT && q < CP && !accum_color) continue;
        atomicAdd(&acc[(size_t)k * RQ + q], v);
    }
    if (keep && tid_e == 0) { ck[0] = nc; tile_lp[tile_id] = sweep_id; }
    STAMP(7)   // barrier + flush
    STAMP_FLUSH
}

#define OBIA_ASSIGN_PARAMS                                                                                             \
    const SlicProblem *__restrict__ probs, const float *__restrict__ feat, const uint8_t *__restrict__ mask,             \
        const unsigned *__restrict__ mask4, const float *__restrict__ cent, const int *__restrict__ head,                                                    \
        int32_t *__restrict__ labels, unsigned long long *__restrict__ acc, int RQ, int accumulate, int store_labels,     \
        int start_label, float fs, const int *__restrict__ bin_stamp, int *__restrict__ tile_lp,                    \
        int *__restrict__ cache_k, unsigned long long *__restrict__ cache_q, int sweep_id, int use_cache,                 \
        unsigned long long *__restrict__ px_counter, const int *__restrict__ tile_prob, int total_tiles_all,              \
        int *__restrict__ orphan_flag, int tiles_per_prob, const float *__restrict__ fbox, int tile_base, int nch_arg,     \
        int *__restrict__ tl_k, unsigned *__restrict__ tl_fp, int *__restrict__ tl_meta, const int *__restrict__ tl_req
#define OBIA_ASSIGN_ARGS                                                                                               \
    probs, feat, mask, mask4, cent, head, labels, acc, RQ, accumulate, store_labels, start_label, fs, bin_stamp, tile_lp, \
        cache_k, cache_q, sweep_id, use_cache, px_counter, tile_prob, total_tiles_all, orphan_flag, tiles_per_prob, fbox, tile_base, nch_arg, \
        tl_k, tl_fp, tl_meta, tl_req

// the colour sweeps and the last pre-pass sweep
template <int CP, bool MASKED, bool IGNORE_COLOR, bool FIXPT, bool SLICZERO, int NCH = CP>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(ASSIGN_WAVES, ASSIGN_WAVES))) void slic_assign_kernel(OBIA_ASSIGN_PARAMS) {
    slic_assign_body<CP, MASKED, IGNORE_COLOR, FIXPT, SLICZERO, false, false, NCH>(OBIA_ASSIGN_ARGS);
}

// the colour sweeps at low compactness: with the colour-box bound (one more LDS table, a few more registers)
template <int CP, bool MASKED, int NCH = CP>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(6, 6))) void slic_assign_collb_kernel(OBIA_ASSIGN_PARAMS) {
    slic_assign_body<CP, MASKED, false, false, false, false, true, NCH>(OBIA_ASSIGN_ARGS);
}

// the pre-pass sweeps that fold no colours: no feature registers, 4 KB of LDS
template <int CP, bool MASKED, bool FIXPT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(LEAN_WAVES, LEAN_WAVES))) void slic_prepass_kernel(OBIA_ASSIGN_PARAMS) {
    slic_assign_body<CP, MASKED, true, FIXPT, false, true, false>(OBIA_ASSIGN_ARGS);
}

#ifdef OBIA_STAMP
extern "C" void obia_debug_timeline(unsigned long long *out, int nwaves) {   // 24 qwords per wave, waves of tile t at 4t .. 4t+3
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), sizeof(unsigned long long) * 24 * (size_t)nwaves);
}
#endif

// SLIC-zero: after a centroid update, max_dist_color[k] = max(max_dist_color[k], colour distance of every pixel of
// cluster k to the NEW centroid) (_slic.pyx, the loop after the centroid recomputation; oracle/obia_oracle.c:273-286).
// The value lives in slot 7 of the centroid record; distances are >= 0, so the float bit pattern orders like the value.
template <int CP, bool MASKED>
__global__ __launch_bounds__(256) void slic_maxdist_kernel(const SlicProblem *__restrict__ probs, const float *__restrict__ feat,
                                                           const uint8_t *__restrict__ mask, const int32_t *__restrict__ labels,
                                                           float *__restrict__ cent, int start_label, int nch) {
    constexpr int RS = CENT_REC + CP;
    const SlicProblem P = probs[blockIdx.y];
    for (int y = blockIdx.x; y < P.H; y += gridDim.x)
        for (int x = threadIdx.x; x < P.W; x += 256) {
            const long long pix = P.pix_off + (long long)y * P.W + x;
            if (MASKED && mask[pix] == 0) continue;
            const int l = labels[pix];
            if (l < start_label) continue;
            const int k = l - start_label + P.cent_off;
            const float *rec = cent + (size_t)k * RS;
            float dc = 0.0f;
#pragma unroll
            for (int ch = 0; ch < CP; ++ch) {
                if (ch >= nch) break;   // (padded channels: 0 - 0)
                const float t = feat_at(feat, P, CP, y, x, ch) - rec[CENT_REC + ch];
                dc += t * t;
            }
            unsigned *slot = reinterpret_cast<unsigned *>(cent + (size_t)k * RS + 7);
            if (__float_as_uint(dc) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, __float_as_uint(dc));
        }
}

// A group of consecutive problems of the batch whose prep / sweep chain runs on its own stream (slic_run_sweeps): the
// centroids [k0, k1), bins [cell0, cell1) and tiles [tile0, tile1) of the batch's tables.
struct SweepGroup { int k0, k1, cell0, cell1, tile0, tile1; long long pix0, pix1; hipStream_t stream; int p0, p1, kmax; };   // [p0, p1): the group's problems, kmax: their largest K

struct FixedPointState {   // exit_on_fixed_point bookkeeping (device pointers; null when the option is off)
    int *bin_stamp = nullptr, *tile_lp = nullptr, *cache_k = nullptr;
    unsigned long long *cache_q = nullptr;
};

template <int CP>
static void launch_assign(obia_ctx *ctx, SlicBatch &b, int ignore_color, int accumulate, int accum_color, int store_labels,
                          int *orphan_flag, const FixedPointState &fp, int sweep_id, int use_cache, unsigned long long *px_counter,
                          const KernelSpan &span, const SweepGroup &sg, const int *head_cur) {
    constexpr int XGH = OBIA_XCD_GROUP;
    if (sg.tile1 <= sg.tile0) return;
    dim3 grid(8 * XGH * (unsigned)((sg.tile1 - sg.tile0 + 8 * XGH - 1) / (8 * XGH)));   // whole groups of 8 XCDs x XG tiles (see slic_assign_body)
    const int RQ = acc_record_qwords(CP);
    int tpp = b.probs.empty() ? 0 : b.probs[0].tiles_x * b.probs[0].tiles_y;   // tiles per problem if all problems agree, else 0
    for (auto &P : b.probs) if (P.tiles_x * P.tiles_y != tpp) tpp = 0;
#define LAUNCH_K_(...)                                                                                               \
    hipExtLaunchKernelGGL(HIP_KERNEL_NAME(__VA_ARGS__), grid, dim3(NT), 0, sg.stream, span.a, span.b, 0, b.d_probs, b.d_feat,   \
                          b.d_mask, b.d_mask4, b.d_cent, head_cur, b.d_labels, b.d_acc, RQ, accumulate, store_labels, b.start_label,      \
                          (float)b.fscale, fp.bin_stamp, fp.tile_lp, fp.cache_k, fp.cache_q, sweep_id, use_cache, px_counter,          \
                          b.d_tile_prob, sg.tile1, orphan_flag, tpp, b.d_fbox, sg.tile0, b.C, b.d_tl_k, b.d_tl_fp, b.d_tl_meta, b.d_tl_req)
    // channels that exist: C of the CP = 4 * ceil(C / 4) the planes and records hold.  The two kernels that run 9 of every 10
    // sweeps come in a variant per padding (slic_assign_body: NCH); the others treat the padded channels like real ones.
    const int pad = CP - b.C;
#define LAUNCH_ASSIGN_(M, I, F, Z) LAUNCH_K_(slic_assign_kernel<CP, M, I, F, Z>)
#define LAUNCH_MAIN_(M)                                                                                              \
    do {                                                                                                             \
        if (pad == 1) LAUNCH_K_(slic_assign_kernel<CP, M, false, false, false, CP - 1>);                             \
        else if (pad == 2) LAUNCH_K_(slic_assign_kernel<CP, M, false, false, false, CP - 2>);                        \
        else if (pad == 3) LAUNCH_K_(slic_assign_kernel<CP, M, false, false, false, CP - 3>);                        \
        else LAUNCH_K_(slic_assign_kernel<CP, M, false, false, false>);                                              \
    } while (0)
    // SLIC-zero only changes the colour sweeps (the spatial pre-pass computes no colour term) and is not combined with
    // the fixed-point cache (the per-cluster scale changes after the records were compared)
#define LAUNCH_COLLB_(M)                                                                                             \
    do {                                                                                                             \
        if (pad == 1) LAUNCH_K_(slic_assign_collb_kernel<CP, M, CP - 1>);                                            \
        else if (pad == 2) LAUNCH_K_(slic_assign_collb_kernel<CP, M, CP - 2>);                                       \
        else if (pad == 3) LAUNCH_K_(slic_assign_collb_kernel<CP, M, CP - 3>);                                       \
        else LAUNCH_K_(slic_assign_collb_kernel<CP, M>);                                                             \
    } while (0)
#define LAUNCH_LEAN_(M, F) LAUNCH_K_(slic_prepass_kernel<CP, M, F>)
#define LAUNCH_ASSIGN(M, I)                                                                                          \
    do {                                                                                                             \
        if ((I) && !accum_color) { if (fp.bin_stamp) LAUNCH_LEAN_(M, true); else LAUNCH_LEAN_(M, false); }           \
        else if (b.slic_zero && !(I)) LAUNCH_ASSIGN_(M, false, false, true);                                         \
        else if (fp.bin_stamp) LAUNCH_ASSIGN_(M, I, true, false);                                                    \
        else if (b.col_lb && b.d_fbox && !(I)) LAUNCH_COLLB_(M);                                                     \
        else if (!(I)) LAUNCH_MAIN_(M);                                                                              \
        else LAUNCH_ASSIGN_(M, I, false, false);                                                                     \
    } while (0)
    if (b.masked) { if (ignore_color) LAUNCH_ASSIGN(true, true); else LAUNCH_ASSIGN(true, false); }
    else LAUNCH_ASSIGN(false, false);
#undef LAUNCH_ASSIGN
#undef LAUNCH_ASSIGN_
#undef LAUNCH_LEAN_
#undef LAUNCH_COLLB_
#undef LAUNCH_MAIN_
#undef LAUNCH_K_
}

// d_mask -> d_mask4 (slic.hpp): dword (q, x) = the mask bytes of the rows 4q .. 4q+3 at column x.  Once per batch; every sweep then
// reads a lane's strip with one load instead of four.  A thread owns four columns of one quad row: four dword loads (one per row:
// a wave reads 256 contiguous bytes of each), a 4 x 4 byte transpose, one 16-byte store.  A problem whose mask hides nothing is
// skipped: its sweeps never read the packed mask (slic_assign_body: all_valid).
__global__ __launch_bounds__(256) void mask_pack4_kernel(const SlicProblem *__restrict__ probs, const uint8_t *__restrict__ mask,
                                                         unsigned *__restrict__ mask4) {
    const SlicProblem P = probs[blockIdx.y];
    if ((long long)P.n_valid == (long long)P.H * (long long)P.W) return;
    const int nq = (P.H + 3) >> 2, W = P.W;
    const uint8_t *mb = mask + P.pix_off;
    unsigned *out = mask4 + P.m4_off;
    const bool vec = (W % 4 == 0) && (P.pix_off % 4 == 0) && (P.m4_off % 4 == 0) && (reinterpret_cast<uintptr_t>(mask) % 4 == 0) &&
                     (reinterpret_cast<uintptr_t>(mask4) % 16 == 0);
    for (int q = blockIdx.x; q < nq; q += gridDim.x) {
        if (vec) {
            for (int x4 = threadIdx.x; x4 < (W >> 2); x4 += 256) {
                unsigned r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = 4 * q + j;
                    r[j] = y < P.H ? *reinterpret_cast<const unsigned *>(mb + (long long)y * W + 4 * x4) : 0u;
                }
                uint4 o;
                unsigned *ov = &o.x;
#pragma unroll
                for (int c = 0; c < 4; ++c) {   // column c: byte c of every row, normalised to 0 / 1
                    unsigned w = 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w |= (unsigned)(((r[j] >> (8 * c)) & 0xffu) != 0u) << (8 * j);
                    ov[c] = w;
                }
                *reinterpret_cast<uint4 *>(out + (long long)q * W + 4 * x4) = o;
            }
        } else {
            for (int x = threadIdx.x; x < W; x += 256) {
                unsigned w = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = 4 * q + j;
                    if (y < P.H) w |= (unsigned)(mb[(long long)y * W + x] != 0) << (8 * j);
                }
                out[(long long)q * W + x] = w;
            }
        }
    }
}

__global__ void fill_i32_kernel(int32_t *p, long long n, int32_t v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

int slic_run_sweeps(obia_ctx *ctx, SlicBatch &b, int mode) {
    auto fill_labels = [&]() {   // nearest[:] = start_label - 1, once (before the loop of _slic_cython)
        long long n = b.total_pix;
        int blocks = cdiv(n, 256 * 8);
        if (blocks > 65535) blocks = 65535;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(fill_i32_kernel, dim3(blocks), dim3(256), 0, ctx->stream, b.d_labels, n, b.start_label - 1);
    };
    // (nearest[:] = start_label - 1 is written further down, and only when some pixel can keep that value)
    if (std::getenv("OBIA_DEBUG_SYNC"))
        for (size_t p = 0; p < b.probs.size(); ++p) {
            const SlicProblem &P = b.probs[p];
            fprintf(stderr, "[obia debug]   problem %zu: %dx%d K %d steps %d %d bins %dx%d cell_off %d cent_off %d tiles %dx%d tile_off %d n_valid %d (total_cent %d total_cells %d tiles_all %lld)\n",
                    p, P.H, P.W, P.K, P.sy, P.sx, P.ncy, P.ncx, P.cell_off, P.cent_off, P.tiles_y, P.tiles_x, P.tile_off, P.n_valid,
                    b.total_cent, b.total_cells, b.total_tiles_all);
        }
    if (b.total_tiles <= 0 || b.max_iter <= 0) { fill_labels(); return OBIA_OK; }   // no sweep: every pixel keeps the fill value
    // the sweep addresses a footprint's pixels as a 64-bit wave-uniform base plus a 32-bit lane offset (16 rows x W x 64 B)
    for (auto &P : b.probs)
        if (P.W >= (1 << 22)) { set_error("rasters / tile windows wider than 4194303 pixels are not supported (got %d)", P.W); return OBIA_E_UNSUPPORTED; }
    const int passes = b.masked ? 2 : 1;   // maskSLIC: spatial-only pre-pass first (slic_superpixels.py:310-314)
    const int RQ = acc_record_qwords(b.CP);
    Arena &A = ctx->arena;
    // pixel counters (profiling): 256 slots each for the colour sweeps and the pre-pass sweeps, summed on the host;
    // slot 512: the orphan flag of the sweeps that do not store their labels
    unsigned long long *d_px = A.get<unsigned long long>(513);
    if (!d_px) return OBIA_E_NOMEM;
    int *d_orphan = reinterpret_cast<int *>(d_px + 512);
    FixedPointState fp;
    if (b.exit_on_fixed_point) {
        const size_t nt = (size_t)b.total_tiles_all;
        fp.bin_stamp = A.get<int>((size_t)b.total_cells);
        fp.tile_lp = A.get<int>(nt);
        fp.cache_k = A.get<int>(nt * (MAXC + 1));
        fp.cache_q = A.get<unsigned long long>(nt * MAXC * (size_t)(b.CP + 3));
        if (!fp.bin_stamp || !fp.tile_lp || !fp.cache_k || !fp.cache_q) return OBIA_E_NOMEM;
        OBIA_HIP_TRY(hipMemsetAsync(fp.bin_stamp, 0, sizeof(int) * (size_t)b.total_cells, ctx->stream));
        OBIA_HIP_TRY(hipMemsetAsync(fp.tile_lp, 0, sizeof(int) * nt, ctx->stream));
    }
    {   // candidate lists of the sweep tiles (see the top of the file)
        const size_t nt = (size_t)b.total_tiles_all;
        b.d_tl_k = A.get<int>(nt * MAXC);
        b.d_tl_fp = A.get<unsigned>(nt * NT);
        b.d_tl_meta = A.get<int>(nt * 3);   // {entries, build word} per tile, then the rebuild requests: ONE fill resets all three
        b.d_tl_req = b.d_tl_meta ? b.d_tl_meta + nt * 2 : nullptr;
        b.d_ref = A.get<float>((size_t)b.total_cent * 2);
        if (!b.d_tl_k || !b.d_tl_fp || !b.d_tl_meta || !b.d_tl_req || !b.d_ref) return OBIA_E_NOMEM;
    }
    int maxh_z = 1;
    for (auto &P : b.probs) if (P.H > maxh_z) maxh_z = P.H;
    if (b.masked && b.d_mask && !b.d_mask4) {   // the packed mask of the sweeps (problems whose mask hides nothing never read it); the
                                                // tiler's mask kernel writes it on the way (tiling.hip: tile_mask_kernel<true>)
        b.d_mask4 = A.get<unsigned>((size_t)b.total_m4);
        if (!b.d_mask4) return OBIA_E_NOMEM;
        int gq = (maxh_z + 3) / 4;
        if (gq > 16384) gq = 16384;
        hipLaunchKernelGGL(mask_pack4_kernel, dim3(gq, b.nprob), dim3(256), 0, ctx->stream, b.d_probs, b.d_mask, b.d_mask4);
    }
    if (maxh_z > 4096) maxh_z = 4096;

    // Groups of problems.  The sweeps of ONE launch end with a tail (the last workgroups run on a half-empty chip), the next
    // launch starts with a ramp (every workgroup stages its candidates before anyone computes), and between them sit two launch
    // gaps and the small prep kernel: ~25 us per iteration that a white tile row (4 tiles, 130-us sweeps) cannot hide.  The
    // problems of a batch never exchange anything during the sweeps, so consecutive problems are dealt into groups whose
    // prep -> sweep chains CAN run on streams of their own (OBIA_SWEEP_GROUPS=2..4): one group's sweep fills the bubbles of the
    // others'.  Same kernels, same arithmetic, same integer accumulators per problem: the labels do not depend on the grouping
    // (tests/test_gpu_sweep_groups.py).  Measured on the headline workload: 2 groups +2 % (5715-5724 -> 5824-5858 Mpixel/s), 4
    // groups -5 %: kernels that share the chip slow each other down by most of what the hidden bubbles gain (the union of the
    // colour sweeps' run time grows from 19.3 to 20.5 ms per step), and the per-launch durations no longer mean anything -- so
    // the default is ONE group, the option stays for batches of many small problems.
    // Always one group when a step of the loop touches the whole batch (SLIC-zero's max-distance pass) or in the debug mode.
    std::vector<SweepGroup> groups;
    {
        const char *env_groups = std::getenv("OBIA_SWEEP_GROUPS");   // (read per batch: the tests switch it inside one process)
        int ng = (env_groups && atoi(env_groups) > 0) ? atoi(env_groups) : OBIA_SWEEP_GROUPS_DEFAULT;
        if (ng > 1 + obia_ctx::MAX_SIDE) ng = 1 + obia_ctx::MAX_SIDE;
        if (ng > b.nprob) ng = b.nprob;
        if (b.slic_zero || std::getenv("OBIA_DEBUG_SYNC")) ng = 1;
        if (ng < 1) ng = 1;
        if (ng > 1) OBIA_TRY(side_streams(ctx, ng - 1));
        // consecutive problems, balanced by tiles (the sweep's unit of work): group g starts at the first problem whose first tile
        // lies at or beyond g / ng of the batch's tiles
        std::vector<int> cut(1, 0);
        for (int g = 1; g < ng; ++g) {
            const long long want = b.total_tiles_all * (long long)g / ng;
            int p = cut.back();
            while (p < b.nprob && b.probs[p].tile_off < want) ++p;
            if (p > cut.back() && p < b.nprob) cut.push_back(p);
        }
        cut.push_back(b.nprob);
        for (size_t g = 0; g + 1 < cut.size(); ++g) {
            const SlicProblem &P0 = b.probs[cut[g]];
            const bool last = cut[g + 1] >= b.nprob;
            SweepGroup sg;
            sg.k0 = P0.cent_off;   sg.k1 = last ? b.total_cent : b.probs[cut[g + 1]].cent_off;
            sg.cell0 = P0.cell_off; sg.cell1 = last ? b.total_cells : b.probs[cut[g + 1]].cell_off;
            sg.tile0 = P0.tile_off; sg.tile1 = last ? (int)b.total_tiles_all : b.probs[cut[g + 1]].tile_off;
            sg.pix0 = P0.pix_off;   sg.pix1 = last ? b.total_pix : b.probs[cut[g + 1]].pix_off;
            sg.stream = g == 0 ? ctx->stream : ctx->side[g - 1];
            sg.p0 = cut[g]; sg.p1 = cut[g + 1]; sg.kmax = 1;
            for (int p = sg.p0; p < sg.p1; ++p) if (b.probs[p].K > sg.kmax) sg.kmax = b.probs[p].K;
            groups.push_back(sg);
        }
    }

    // All sweeps of the batch.  store_all = false: only the very last sweep stores its labels (the others' labels are dead
    // stores unless a valid pixel is reached by no window -- see the label stage of the sweep); true: every sweep stores, the
    // reference's literal behaviour, needed by the fixed-point replay (labels "already in place") and by SLIC-zero (its
    // max-colour-distance pass reads the last assignment).
    auto run_all = [&](bool store_all) -> int {
        OBIA_HIP_TRY(hipMemsetAsync(d_px, 0, sizeof(unsigned long long) * 513, ctx->stream));
        OBIA_HIP_TRY(hipMemsetAsync(b.d_head, 0xff, sizeof(int) * (size_t)b.total_cells, ctx->stream));   // buffer 0 only
        // no tile has a list, nobody asked for a rebuild (-1 everywhere: the first sweep builds every list)
        OBIA_HIP_TRY(hipMemsetAsync(b.d_tl_meta, 0xff, sizeof(int) * 3 * (size_t)b.total_tiles_all, ctx->stream));
        debug_sync(ctx, "sweeps: memsets");
        if (groups.size() > 1) {   // fork: the side streams start after everything queued on the context's stream so far
            OBIA_HIP_TRY(hipEventRecord(ctx->fork_ev, ctx->stream));
            for (size_t g = 1; g < groups.size(); ++g) OBIA_HIP_TRY(hipStreamWaitEvent(groups[g].stream, ctx->fork_ev, 0));
        }
        for (size_t gi = 0; gi < groups.size(); ++gi) {
        const SweepGroup &sg = groups[gi];
        bool first = true;
        int sweep_no = 0;
        for (int pass = 0; pass < passes; ++pass) {
            const int ignore_color = (b.masked && pass == 0) ? 1 : 0;
            const bool last_pass = (pass == passes - 1);
            if (pass > 0 && store_all && sg.pix1 > sg.pix0) {
                // the main pass is a second call of _slic_cython (slic_superpixels.py:310-318): `nearest` starts from the fill
                // value again, a pixel no window reaches in its first sweep does not inherit a pre-pass label.  (Only when the
                // pre-pass stored labels at all: otherwise the fill of the batch's start is still in place.)
                const long long n = sg.pix1 - sg.pix0;
                int blocks = cdiv(n, 256 * 8);
                if (blocks > 65535) blocks = 65535;
                hipLaunchKernelGGL(fill_i32_kernel, dim3(blocks), dim3(256), 0, sg.stream, b.d_labels + sg.pix0, n, b.start_label - 1);
                // (exit_on_fixed_point: no tile may replay "labels already in place" across the refill -- every tile of the group
                // is evaluated by the first sweep of the main pass)
                if (fp.tile_lp && sg.tile1 > sg.tile0)
                    OBIA_HIP_TRY(hipMemsetAsync(fp.tile_lp + sg.tile0, 0, sizeof(int) * (size_t)(sg.tile1 - sg.tile0), sg.stream));
            }
            for (int it = 0; it < b.max_iter; ++it) {
                int *head_cur = b.d_head + (size_t)(sweep_no & 1) * b.total_cells;
                int *head_nxt = b.d_head + (size_t)((sweep_no + 1) & 1) * b.total_cells;
                ++sweep_no;   // sweep ids start at 1
                // SLIC-zero: the per-cluster colour scale restarts at 1 with the colour pass and is carried afterwards
                const int zmode = (b.slic_zero && !ignore_color) ? (it == 0 ? 2 : 1) : 0;
                const long long nk = sg.k1 > sg.k0 ? sg.k1 - sg.k0 : 1;   // (at least one block: it also resets the group's bins)
                const bool prep_grouped = std::getenv("OBIA_PREP_GROUPED") != nullptr;   // developer switch (A/B timing, tests/test_gpu_prep_kernels.py): the 16-lanes-per-centroid kernel
#define LAUNCH_PREP_LANE(CPV)                                                                                         \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(slic_prep_lane_kernel<CPV>), dim3(cdiv(sg.kmax, 64), sg.p1 > sg.p0 ? sg.p1 - sg.p0 : 1), dim3(64), 0, \
                       sg.stream, b.d_probs, sg.p1 > sg.p0 ? sg.p0 : 0, first ? 1 : 0, zmode, b.d_seed, b.d_acc, RQ, 1.0 / b.fscale, b.d_cent,  \
                       head_cur, head_nxt, sg.cell1, fp.bin_stamp, sweep_no, sg.cell0, b.d_ref, b.d_tl_req)
                if (!prep_grouped) {
                    switch (b.CP) {
                        case 4: LAUNCH_PREP_LANE(4); break;
                        case 8: LAUNCH_PREP_LANE(8); break;
                        case 12: LAUNCH_PREP_LANE(12); break;
                        default: LAUNCH_PREP_LANE(16); break;
                    }
                } else if (RQ == 16)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(slic_prep_kernel<16>), dim3(cdiv(nk * 16, 256)), dim3(256), 0,
                                       sg.stream, b.d_probs, b.d_cent_prob, sg.k1, b.CP, first ? 1 : 0, zmode, b.d_seed, b.d_acc,
                                       1.0 / b.fscale, b.d_cent, head_cur, head_nxt, sg.cell1, fp.bin_stamp, sweep_no, sg.k0, sg.cell0, b.d_ref, b.d_tl_req);
                else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(slic_prep_kernel<32>), dim3(cdiv(nk * 32, 256)), dim3(256), 0,
                                       sg.stream, b.d_probs, b.d_cent_prob, sg.k1, b.CP, first ? 1 : 0, zmode, b.d_seed, b.d_acc,
                                       1.0 / b.fscale, b.d_cent, head_cur, head_nxt, sg.cell1, fp.bin_stamp, sweep_no, sg.k0, sg.cell0, b.d_ref, b.d_tl_req);
#undef LAUNCH_PREP_LANE
                b.d_head_cur = head_cur;
                debug_sync(ctx, "sweeps: prep");
                first = false;
                if (zmode == 1) {   // the centroids just moved: raise max_dist_color from the assignment of the last sweep
                    dim3 zg(maxh_z, b.nprob);
#define LAUNCH_MAXDIST(CPV)                                                                                           \
    do {                                                                                                              \
        if (b.masked) hipLaunchKernelGGL(HIP_KERNEL_NAME(slic_maxdist_kernel<CPV, true>), zg, dim3(256), 0, ctx->stream, b.d_probs, \
                                         b.d_feat, b.d_mask, b.d_labels, b.d_cent, b.start_label, b.C);            \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(slic_maxdist_kernel<CPV, false>), zg, dim3(256), 0, ctx->stream, b.d_probs, \
                                b.d_feat, b.d_mask, b.d_labels, b.d_cent, b.start_label, b.C);                     \
    } while (0)
                    switch (b.CP) {
                        case 4: LAUNCH_MAXDIST(4); break;
                        case 8: LAUNCH_MAXDIST(8); break;
                        case 12: LAUNCH_MAXDIST(12); break;
                        default: LAUNCH_MAXDIST(16); break;
                    }
#undef LAUNCH_MAXDIST
                }
                // the update after the very last sweep is never read: skip its accumulation
                const bool very_last = last_pass && it == b.max_iter - 1;
                const int accumulate = very_last ? 0 : 1;
                const int accum_color = (!ignore_color || it == b.max_iter - 1) ? 1 : 0;
                const int store_labels = store_all ? 2 : (very_last ? 1 : 0);   // (see orphan_needs_repeat)
                // the last pre-pass sweep is the only one of its pass that folds colours (they seed the main pass): the
                // caches written by the earlier pre-pass sweeps hold no colour sums, so it evaluates every tile
                const int use_cache = (ignore_color && it == b.max_iter - 1) ? 0 : 1;
                if (sg.tile1 > sg.tile0) {   // (an empty group launches nothing: no span either -- its pooled events would keep an older recording)
                    KernelSpan span(ctx, ignore_color ? T_PREPASS : T_ASSIGN);   // events bound to the dispatch
                    unsigned long long *pxc = ctx->profiling ? d_px + (ignore_color ? 256 : 0) : nullptr;
                    if (gi == 0 && ctx->profiling && !ignore_color && store_labels) ctx->timing.assign_store_px += (double)b.total_pix;
                    switch (b.CP) {
                        case 8: launch_assign<8>(ctx, b, ignore_color, accumulate, accum_color, store_labels, d_orphan, fp, sweep_no, use_cache, pxc, span, sg, head_cur); break;
#ifndef OBIA_ONLY_CP8   /* developer builds (tools/build_variant.sh ... -DOBIA_ONLY_CP8): only the 5..8-band sweep kernels are compiled */
                        case 4: launch_assign<4>(ctx, b, ignore_color, accumulate, accum_color, store_labels, d_orphan, fp, sweep_no, use_cache, pxc, span, sg, head_cur); break;
                        case 12: launch_assign<12>(ctx, b, ignore_color, accumulate, accum_color, store_labels, d_orphan, fp, sweep_no, use_cache, pxc, span, sg, head_cur); break;
                        case 16: launch_assign<16>(ctx, b, ignore_color, accumulate, accum_color, store_labels, d_orphan, fp, sweep_no, use_cache, pxc, span, sg, head_cur); break;
#endif
                        default: set_error("bad CP"); return OBIA_E_INVALID;
                    }
                }
                debug_sync(ctx, ignore_color ? "sweeps: pre-pass sweep" : "sweeps: colour sweep");
            }
        }
        }
        for (size_t g = 1; g < groups.size(); ++g) {   // join: whatever follows on the context's stream sees every group's labels
            OBIA_HIP_TRY(hipEventRecord(ctx->join_ev[g - 1], groups[g].stream));
            OBIA_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->join_ev[g - 1], 0));
        }
        OBIA_HIP_TRY(hipGetLastError());
        return OBIA_OK;
    };
    static const bool env_store_all = std::getenv("OBIA_STORE_ALL_LABELS") != nullptr;   // developer switch (A/B timing)
    const bool store_all = b.exit_on_fixed_point || b.slic_zero || env_store_all;
    // The fill value survives in exactly one kind of pixel: a valid one that no window reached ("orphan").  When only the last sweep
    // stores, such a pixel raises the flag and the batch is repeated below (with the fill) -- unless the batch has ONE sweep in all,
    // where the fill value is what the reference keeps.  Every other pixel is written by the last sweep (masked ones included): the
    // 1.2 GB fill of a bench batch is skipped on the common path.
    if (store_all || passes * b.max_iter <= 1) {
        fill_labels();
        debug_sync(ctx, "sweeps: label fill");
    }
    unsigned long long h[513];
    if (mode == 2) {   // the repeat of a batch whose deferred flag said "orphan": every sweep stores its labels
        ctx->timing.batch_repeats += 1;
        fill_labels();
        OBIA_HIP_TRY(hipMemsetAsync(b.d_acc, 0, sizeof(unsigned long long) * (size_t)b.total_cent * RQ, ctx->stream));
        OBIA_TRY(run_all(true));
        OBIA_TRY(read_back(ctx, h, d_px, sizeof(h)));
        if (ctx->profiling)
            for (int i = 0; i < 256; ++i) { ctx->timing.assign_px += (double)h[i]; ctx->timing.prepass_px += (double)h[256 + i]; }
        return OBIA_OK;
    }
    OBIA_TRY(run_all(store_all));
    if (mode == 1 && !store_all) {   // the flag and the counters travel to pinned memory behind the sweeps; looked at in slic_sweeps_settle
        if (!ctx->defer_buf) OBIA_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->defer_buf), sizeof(h), hipHostMallocDefault));
        OBIA_HIP_TRY(hipMemcpyAsync(ctx->defer_buf, d_px, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        ctx->defer_pending = true;
        return OBIA_OK;
    }
    OBIA_TRY(read_back(ctx, h, d_px, sizeof(h)));
    if (!store_all && (h[512] & 0xffffffffull) != 0ull) {
        // a pixel needed the label of an earlier sweep: repeat the batch from the seeds with every sweep storing its labels
        ctx->timing.batch_repeats += 1;
        fill_labels();
        OBIA_HIP_TRY(hipMemsetAsync(b.d_acc, 0, sizeof(unsigned long long) * (size_t)b.total_cent * RQ, ctx->stream));
        OBIA_TRY(run_all(true));
        OBIA_TRY(read_back(ctx, h, d_px, sizeof(h)));
    }
    if (ctx->profiling)
        for (int i = 0; i < 256; ++i) { ctx->timing.assign_px += (double)h[i]; ctx->timing.prepass_px += (double)h[256 + i]; }
    return OBIA_OK;
}

// (see slic.hpp) the stream has been synchronised since slic_run_sweeps(..., 1) returned
int slic_sweeps_settle(obia_ctx *ctx, SlicBatch &b, bool *repeat) {
    (void)b;
    *repeat = false;
    if (!ctx->defer_pending) return OBIA_OK;
    ctx->defer_pending = false;
    const unsigned long long *h = ctx->defer_buf;
    if ((h[512] & 0xffffffffull) != 0ull) { *repeat = true; return OBIA_OK; }   // (the repeat counts its own pixels)
    if (ctx->profiling)
        for (int i = 0; i < 256; ++i) { ctx->timing.assign_px += (double)h[i]; ctx->timing.prepass_px += (double)h[256 + i]; }
    return OBIA_OK;
}

}  // namespace obia
