// slic.hpp -- batched SLIC engine: host-side planning structures and entry points.
#pragma once
#include "common.hpp"

namespace obia {

// One independent SLIC problem (a whole image, or one tile window of the tiled driver).
// Device-visible, read through scalar loads (wave-uniform).
struct SlicProblem {
    int H, W;            // window size in pixels
    int K;               // number of centroids
    int sy, sx;          // window steps: regular_grid((1,H,W), K) (slic_superpixels.py -> _slic_cython)
    int ncy, ncx;        // bin grid: cells of sy x sx pixels
    float spatial_w;     // float(1 / step^2)
    int cent_off;        // first centroid of this problem in the batch-wide centroid arrays
    int cell_off;        // first bin of this problem in the batch-wide bin-head array
    int tile_off;        // first workgroup of this problem in the assign grid
    int tiles_x, tiles_y;
    long long pix_off;   // first pixel of this problem in the dense per-problem pixel arrays
    long long feat_off;  // first float4 of this problem in the feature planes (see feat_block_f4)
    int XB;              // 16-column blocks per row of the feature planes: ceil(W / 16)
    int n_valid;         // valid (unmasked) pixels
    long long fb_off;    // first record of this problem in the footprint colour boxes (SlicBatch::d_fbox)
    float sp_y, sp_x;    // slic()'s `spacing` of the row / column axis as float32 (1, 1 unless the caller says otherwise)
    int direct;          // 1: every tile takes the direct (unstaged) path -- set when the spacing is not (1, 1)
    int m4_off;          // first dword of this problem in the packed mask (SlicBatch::d_mask4): ceil(H / 4) * W dwords per problem
};

// Source window of a problem inside the caller's raster (feature preparation).
struct SrcWindow {
    int y0, x0, h, w;
    long long pix_off;   // pixel-major feature layout (quickshift): first pixel
    long long feat_off;  // quad-row plane layout (SLIC): first float4
    long long fb_off;    // first footprint colour box (see feat_boxes)
};


// Feature layout of the SLIC sweeps ("quad-row blocks").  Rows are grouped in fours, columns in sixteens; the unit is one
// float4 = ONE channel of the four pixels (4q .. 4q+3, x):
//   float4 index = feat_off + ((q * XB + x / 16) * CP + c) * 16 + x % 16            XB = ceil(W / 16)
// i.e. per (quad row, 16-column block): CP runs of 256 bytes, one per channel -- 4 KB for 8 bands, contiguous.
// A lane of the sweep owns a vertical 1x4 strip, so one 16-byte load brings a channel of its four pixels as two register
// pairs (packed-f32 arithmetic on two pixels per instruction), a quarter wave reads 256 contiguous bytes (whole 128-byte
// lines), the channels of a lane are 256 bytes apart (immediate offsets of one address) and a footprint's quad row is
// one contiguous block.  Rows past H inside the last quad and columns past W inside the last block hold zeros.
inline int feat_xb(int w) { return (w + 15) >> 4; }
inline long long feat_block_f4(int h, int w, int CP) { return (long long)((h + 3) / 4) * feat_xb(w) * CP * 16; }

// Footprint colour boxes (low compactness only, SlicBatch::col_lb): for every 16 x 16 footprint of the sweep -- quad rows
// 4f .. 4f+3 of one 16-column block -- the per-channel minimum and maximum of its features, 2 * CP floats {lo[CP], hi[CP]}.
// A candidate's colour distance to the box is a lower bound of its colour term for every pixel of the footprint: when the
// colour term decides (compactness below ~1) the spatial bound alone visits 13 candidates per footprint, the sum 7.
inline long long feat_boxes(int h, int w) { return (long long)(((h + 3) / 4 + 3) / 4) * feat_xb(w); }

// regular_grid((1,H,W), n) of scikit-image (util/_regular_grid.py:61-83): start/step per axis,
// step 0 == slice(None).
void regular_grid_hw(long long H, long long W, long long n, long long out[4]);

#ifndef OBIA_SWEEP_TW
#define OBIA_SWEEP_TW 64
#endif
#ifndef OBIA_SWEEP_MAXC
#define OBIA_SWEEP_MAXC 96
#endif
constexpr int SWEEP_TW = OBIA_SWEEP_TW, SWEEP_TH = 64;   // workgroup tile of the sweep kernel (pixels)
constexpr int SWEEP_MAXC = OBIA_SWEEP_MAXC;   // candidate slots of a sweep tile (LDS slots of the sweep kernel)
constexpr int CENT_REC = 8;      // header dwords of a centroid record: cy, cx, y0, y1, x0, x1, k, -
// Accumulator record of one centroid, 128-byte aligned so a tile's flush touches two 64-B lines:
//   q[0..CP)  colour sums, 64-bit fixed point     q[CP] = n | (sum_y << 32)     q[CP+1] = sum_x
inline int acc_record_qwords(int CP) { return CP <= 12 ? 16 : 32; }

struct SlicBatch {
    int nprob = 0;
    int C = 0, CP = 0;                 // bands, bands padded to a multiple of 4
    bool masked = false;
    int start_label = 1;
    int max_iter = 10;
    long long total_pix = 0;
    int total_cent = 0, total_cells = 0, total_tiles = 0;   // total_tiles: largest tile count of one problem (grid.x)
    long long total_tiles_all = 0;                          // sum over problems (per-tile state of exit_on_fixed_point)
    std::vector<SlicProblem> probs;    // host copy
    std::vector<SrcWindow> windows;
    // device arrays (arena)
    SlicProblem *d_probs = nullptr;
    SrcWindow *d_windows = nullptr;
    float *d_fbox = nullptr;           // footprint colour boxes (feat_boxes), or null
    bool col_lb = false;               // the sweeps add the colour-box bound to the spatial one (see slic_use_colour_bound)
    float *d_feat = nullptr;           // quad-row planes, 4 * total_feat_f4 floats (pixel-major [total_pix][CP] when !feat_planes)
    bool feat_planes = true;           // false: pixel-major features (quickshift reads them per pixel)
    double sigma[3] = {0.0, 0.0, 0.0};   // Gaussian pre-smoothing (z, y, x), 0 = none (obia_slic_params::sigma_zyx)
    double spacing[3] = {1.0, 1.0, 1.0}; // obia_slic_params::spacing_zyx
    long long total_feat_f4 = 0;       // float4 elements of d_feat (plane layout)
    uint8_t *d_mask = nullptr;         // [total_pix] or null
    unsigned *d_mask4 = nullptr;       // the same bytes packed for the sweeps: dword (q, x) = mask of the rows 4q .. 4q+3 at column x, a byte each
                                       // (a lane's 1x4 strip in ONE load and one register); written by slic_run_sweeps from d_mask
    long long total_m4 = 0;            // dwords of d_mask4
    int32_t *d_labels = nullptr;       // [total_pix] problem-local labels (start_label based)
    float *d_seed = nullptr;           // [total_cent][2]
    int *d_cent_prob = nullptr;        // [total_cent]
    int *d_tile_prob = nullptr;        // [total_tiles_all] problem of every sweep tile, tiles numbered problem by problem in raster order
    float *d_cent = nullptr;           // [total_cent][8 + CP] records
    int *d_head = nullptr;   // two buffers of total_cells (double-buffered per sweep); list links ride in the centroid records
    int *d_head_cur = nullptr;
    unsigned long long *d_acc = nullptr;   // [total_cent] accumulator records, see acc_record_qwords()
    // cached candidate lists of the sweep tiles (slic_sweep.hip, "candidate lists"): allocated by slic_run_sweeps
    int *d_tl_k = nullptr;             // [total_tiles_all][SWEEP_MAXC] centroid indices of a tile's list, ascending
    unsigned *d_tl_fp = nullptr;       // [total_tiles_all][256] per thread: its list slot in each of its wave's four footprints (a byte each)
    int *d_tl_meta = nullptr;          // [total_tiles_all][2] {entries (-1: no list yet, -2: the tile cannot be listed), sweep it was built in}
    int *d_tl_req = nullptr;           // [total_tiles_all] last sweep in which a centroid that left its margin asked the tile to rebuild (-1: none)
    float *d_ref = nullptr;            // [total_cent][2] position a centroid's margin is measured from
    double fscale = 1.0;
    float prescale = 1.0f;             // power of two already folded into the feature planes, the spatial weight and the centroid colours (slic_prescale); fscale is what is left
    bool exit_on_fixed_point = false;
    bool slic_zero = false;            // SLIC-zero: colour term scaled by the cluster's largest colour distance so far
};

// Feature preparation for every problem of the batch: per-band min/max of its window, then
// normalise (optional) -> Lab (optional) -> * 1/compactness into d_feat.  `src` is the caller's
// (Hs,Ws,C) raster.  Returns OBIA_E_NONFINITE for constant / non-finite bands.
// `skip` (nullable): when given, a problem whose window holds a constant or non-finite band is flagged
// skip[p] = 1 (its features are zero) instead of failing the whole batch -- the reference's tiler
// swallows the per-tile ValueError (tiling.py:149-150).
// Gaussian pre-smoothing of slic(..., sigma=...) (slic_superpixels.py: ndi.gaussian_filter between the Lab conversion and the scaling):
// sigma per axis (z, y, x; the depth axis has ONE plane and is filtered too, as scipy does), two scratch arrays of total_pix * CP
// floats, the weights of the three passes on the device (smooth_upload_weights) and a scratch word per window.
struct SmoothSpec {
    double sigma[3] = {0.0, 0.0, 0.0};
    float *tmp_a = nullptr, *tmp_b = nullptr;
    double *d_w[3] = {nullptr, nullptr, nullptr};   // [radius + 1]: centre first
    int radius[3] = {0, 0, 0};
    unsigned *d_scratch = nullptr;                  // [np] (max |feature| of the unscaled pass: not used)
    long long maxpix = 0;                           // pixels of the largest window
    bool on() const { return sigma[0] > 1e-15 || sigma[1] > 1e-15 || sigma[2] > 1e-15; }
};
int smooth_prepare(obia_ctx *ctx, SmoothSpec &sm, long long total_pix, long long maxpix, int CP, int np);
int gaussian_weights_host(double sigma, bool sigma_is_f32, std::vector<double> &w);   // returns the radius; w[0] = centre weight
int slic_features_launch(hipStream_t stream, int C, int CP, int np, const SrcWindow *d_windows, int maxh, const float *src, int Ws,
                         int normalize, int to_lab, float ratio, float *d_feat, unsigned *d_keys, bool planes = true,
                         float *d_fbox = nullptr,    // d_fbox (plane layout only): the footprints' colour boxes from the same pass
                         const SmoothSpec *smooth = nullptr);
int slic_features_finish(SlicBatch &b, const unsigned *keys, const unsigned *nonfinite, const unsigned *maxabs_bits, int normalize,
                         std::vector<int> *skip);
int slic_prepare_features(obia_ctx *ctx, SlicBatch &b, const float *src, int Hs, int Ws,
                          int normalize, int to_lab, float ratio, std::vector<int> *skip = nullptr);
// The part of the fixed-point scale of the colour sums that is known BEFORE the features are computed -- normalised bands times
// `ratio` reach exactly `ratio` -- as a power of two.  The caller multiplies it into the ratio it hands to the feature pass (exact: the
// planes hold the reference's features times 2^s) and stores it in SlicBatch::prescale; slic_plan_and_seed scales the spatial weight
// by its square, the centroid colours follow from the sums: every distance is the reference's times 2^(2s), every comparison and
// every tie the same, and the sweeps convert a feature to fixed point by truncation alone (SlicBatch::fscale, taken from the largest
// |feature| as before, comes out as 1).  1 when the range is not known beforehand (no normalisation, Lab) or SLIC-zero is on.
float slic_prescale(float ratio, int normalize, int to_lab, bool slic_zero);
// Does a batch with this image ratio (1 / compactness; to_lab: the features are Lab, ~100 units wide) use the colour-box bound?  (OBIA_COLOUR_BOUND=0/1 overrides.)
bool slic_use_colour_bound(float ratio, bool to_lab = false);

// Seeds (grid or masked grid), fills K / steps / bins in b.probs, uploads descriptors.
// n_segments[p] = requested segments of problem p.
// nvalid (nullable): valid-pixel counts already known to the caller.
// ext (nullable, single-problem batches only): externally supplied initial centroids -- scikit-image's own
// _get_mask_centroids output (slic_superpixels.py:14-68) -- instead of the grid rule; step = max(steps).
struct ExternalSeeds { const double *yx; int n; double step; };
int slic_plan_and_seed(obia_ctx *ctx, SlicBatch &b, const std::vector<int> &n_segments,
                       const std::vector<int> *nvalid = nullptr, const ExternalSeeds *ext = nullptr);
// valid (unmasked) pixels per problem: mask.sum() (tiling.py:133, slic_superpixels.py:322)
int slic_count_valid(obia_ctx *ctx, SlicBatch &b, std::vector<int> &nvalid);

// Runs the sweeps; labels (pre-connectivity) land in b.d_labels.
// mode 0: complete when it returns (one read-back: the orphan flag -- a valid pixel no window reached makes the batch run again with
//         every sweep storing its labels -- and the profiling counters).
// mode 1: the read-back is only QUEUED (round 4: one host round trip less per batch).  The caller synchronises the stream for its own
//         reasons afterwards (the connectivity stage reads its counters back) and then calls slic_sweeps_settle(); when that reports
//         `repeat`, the labels are not final: the caller runs mode 2 and whatever it had computed from the labels again.
// mode 2: the repeat itself (every sweep stores its labels), complete when it returns.
int slic_run_sweeps(obia_ctx *ctx, SlicBatch &b, int mode = 0);
int slic_sweeps_settle(obia_ctx *ctx, SlicBatch &b, bool *repeat);

// Connectivity enforcement on a batch of dense label maps laid out back to back (pix_off); labels come
// out consecutive over the whole batch, in problem order then raster order of each component's first pixel.
struct CcProblem { int H, W; long long pix_off; int min_size; int max_size; };   // component sizes: merge below min, cut at max
// The last step of the enforcement -- every pixel takes the final label of its component -- as data: a caller that moves the labels
// somewhere else anyway (the tiler scatters them into the raster's id space) asks for this instead of the dense label map and
// resolves each pixel where it consumes it (cc_resolve_label): one pass over the batch and 8 bytes per pixel less (round 4).
// The arrays live in the context's arena: valid until the caller rewinds it.
struct CcResolve { const int *parent; const int *newlab; const int *small_final; int start_label; int mask_label; };
#if defined(__HIPCC__)
__device__ __forceinline__ int cc_resolve_label(const CcResolve &R, long long i) {
    // (a pixel's parent is its tile-local root, whose parent is the root of the component: cc.hip, cc_roots_kernel)
    const int p = R.parent[i];
    if (p < 0) return R.mask_label;
    const int r = R.parent[p];
    int nl = R.newlab[r];
    // a small component took the label at the end of its adjacency chain: followed once per COMPONENT (cc_small_final_kernel), not
    // per pixel -- a pixel of a small component costs one more gather, not four per hop (round 4)
    if (nl < 0) nl = R.small_final[-nl - 2];
    return (nl >= 0) ? nl + R.start_label : 0;   // `adjacent = 0` when no labelled neighbour exists
}
#endif
// labels_out: the dense label map.  deferred (nullable): when given, the final relabel pass is NOT run -- labels_out is not
// written -- and *deferred describes how to resolve a pixel.  (The component walks mark their pixels in a map of the context's
// that is all zero between calls, obia_ctx::cc_visited.)
int enforce_connectivity_batch(obia_ctx *ctx, const std::vector<CcProblem> &probs, const int32_t *labels_in,
                               long long total_pix, int start_label, int32_t *labels_out, int *h_n_labels_out,
                               CcResolve *deferred = nullptr);

// Connectivity enforcement on one dense (H,W) label map (device pointers).
int enforce_connectivity_dev(obia_ctx *ctx, const int32_t *labels_in, int H, int W, int min_size,
                             int max_size, int start_label, int32_t *labels_out, int *d_n_labels_out);

// Zonal statistics (device pointers, outputs device).
int zonal_stats_dev(obia_ctx *ctx, const float *raw, const int32_t *labels, int H, int W, int C,
                    const int32_t *bands_host, int n_bands, int n_labels, int start_label,
                    int64_t *count, double *mean, double *var, float *mn, float *mx);
int zonal_moments_dev(obia_ctx *ctx, const float *raw, const int32_t *labels, int H, int W, int C,
                      const int32_t *bands_host, int n_bands, int n_labels, int start_label, const double *mean,
                      double *skew, double *kurt);

}  // namespace obia
