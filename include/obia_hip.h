/*
 * obia_hip.h -- C ABI of libobia_hip.so: the MI355X (gfx950) tiled-SLIC + zonal-statistics path
 * that drops in behind obia.segmentation.segment(method="slic") and
 * obia.utils.tiling.create_tiled_segments.
 *
 * The reference (iosefa/obia) is pure Python and has no FFI of its own; the boundary is defined by
 * its call sites.  Each entry point below names the reference interface it replaces (file:line
 * under /root/reference).  Plain pointers and sizes only -- no torch / numpy types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative OBIA_E_* code on failure;
 *     obia_last_error() returns a thread-local message for the last failure.  No C++ exception
 *     crosses this ABI.
 *   - one obia_ctx per (device, stream).  A context is NOT thread-safe; independent contexts may
 *     run concurrently.  A context owns a growing device workspace that is reused across calls
 *     (no hipMalloc in the steady state).
 *   - `*_dev` functions take DEVICE pointers valid on the context's device and enqueue work on
 *     the context's stream; they synchronise the stream only where they return a host scalar.
 *     Functions without the suffix take HOST pointers and do H2D / D2H themselves.
 *   - images are (H, W, C) float32, band-interleaved, C-contiguous -- the layout of
 *     obia.handlers.geotif.Image.img_data (geotif.py:100, tiling.py:47).  Labels are int32.
 */
#ifndef OBIA_HIP_H
#define OBIA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OBIA_ABI_VERSION 2   /* 2 (round 3): obia_slic_params grew sigma_zyx, spacing_zyx */

#define OBIA_OK 0
#define OBIA_E_INVALID (-1)     /* bad argument (Python side raises ValueError)                     */
#define OBIA_E_HIP (-2)         /* HIP runtime error                                                */
#define OBIA_E_NOMEM (-3)
#define OBIA_E_UNSUPPORTED (-4) /* valid in the reference, not implemented here (NotImplementedError) */
#define OBIA_E_EMPTY (-5)       /* empty / fully masked input (the reference raises ValueError; the tiler
                                   catches it per tile, tiling.py:149-150)                          */
#define OBIA_E_NONFINITE (-6)   /* constant band (0/0 in normalize_band) or NaN/inf input           */

typedef struct obia_ctx obia_ctx;

int obia_abi_version(void);
const char *obia_last_error(void);

/* Create a context on `device_id` with its own HIP stream / on a caller-owned hipStream_t. */
obia_ctx *obia_create(int device_id);
obia_ctx *obia_create_on_stream(int device_id, void *hip_stream);
void obia_destroy(obia_ctx *ctx);
int obia_synchronize(obia_ctx *ctx);
/* bytes of device workspace currently held by the context */
int64_t obia_workspace_bytes(obia_ctx *ctx);

/* SLIC parameters: the keyword arguments obia forwards untouched to skimage.segmentation.slic
 * (segment.py:86 -> segment_boundaries.py:51; tiling.py:137-143). */
typedef struct obia_slic_params {
    double compactness;            /* doubles: Python floats, so 1/compactness and int(factor*size)   */
    double min_size_factor;        /* round exactly as in the reference                               */
    double max_size_factor;
    int32_t n_segments;
    int32_t max_num_iter;          /* scikit-image `max_num_iter` (`max_iter` before 0.19)            */
    int32_t convert2lab;           /* -1 auto (Lab iff C == 3), 0, 1                                  */
    int32_t enforce_connectivity;
    int32_t slic_zero;             /* 1 = SLIC-zero: colour term / max colour distance of the cluster  */
    int32_t start_label;           /* 0 or 1                                                          */
    int32_t normalize_bands;       /* 1: apply normalize_band (segment_boundaries.py:11-16,32-33) to
                                      every band before segmenting, as create_segments does           */
    int32_t exit_on_fixed_point;   /* 1: stop sweeping a raster / tile as soon as a sweep starts from centroid
                                      records that are bit-identical to those of the previous sweep: every
                                      later sweep would reproduce the same labels and the same centroids, so
                                      the result is bit-identical to running all max_num_iter sweeps (the
                                      reference's own `if change == 0: break` intends this but never fires).
                                      0: always run max_num_iter sweeps.  Default 0.                    */
    int32_t reserved;              /* (keeps the doubles below 8-byte aligned; 0)                          */
    double sigma_zyx[3];           /* scikit-image `sigma`: Gaussian pre-smoothing per axis (depth, row, column) of the
                                      (1, H, W, C) image slic() builds -- scipy.ndimage.gaussian_filter, mode 'reflect',
                                      truncate 4, applied after the Lab conversion and before `* 1/compactness`
                                      (slic_superpixels.py; a scalar `sigma` is the same value three times: the one-plane
                                      depth axis is filtered too).  0 = none, the default.  A caller that mirrors slic()
                                      divides a SCALAR sigma by the spacing first and passes a sequence as it is.       */
    double spacing_zyx[3];         /* scikit-image `spacing`: voxel size per axis (depth, row, column); the row / column
                                      differences are scaled by it before they are squared (_slic.pyx: dy = (sy * (cy - y))^2).
                                      (1, 1, 1) = the default.  Anything else takes the direct (unstaged) sweep path: exact,
                                      not tuned -- anisotropic pixels are rare in rasters.                              */
} obia_slic_params;

void obia_slic_default_params(obia_slic_params *p);

/* ---- B1: segmentation operator --------------------------------------------------------------------
 * Replaces  `segments = slic(img_to_segment, **kwargs)`  (segment_boundaries.py:48-51) together with
 * the per-band normalisation in front of it (segment_boundaries.py:32-33, when normalize_bands=1).
 *   img        (H,W,C) float32; not modified
 *   mask       (H,W) uint8, nullable; 0 = excluded (labelled start_label-1)
 *   labels_out (H,W) int32: consecutive labels from start_label (after connectivity enforcement)
 *   n_labels_out  host int: number of labels produced
 * With a mask the maskSLIC structure of the reference is kept (spatial-only pre-pass, then the
 * main pass) but seeding uses the deterministic masked-grid rule documented in DESIGN.md.            */
int obia_slic_f32(obia_ctx *ctx, const float *img_hwc, int H, int W, int C, const uint8_t *mask,
                  const obia_slic_params *params, int32_t *labels_out, int *n_labels_out);
int obia_slic_f32_dev(obia_ctx *ctx, const float *img_hwc, int H, int W, int C, const uint8_t *mask,
                      const obia_slic_params *params, int32_t *labels_out, int *n_labels_out);

/* Stage-level entry points (device pointers), used by the parity tests to compare each stage with
 * the oracle: labels before connectivity enforcement, and connectivity enforcement alone
 * (restates _enforce_label_connectivity_cython, slic_superpixels.py:320-328).                        */
int obia_slic_assign_only_f32_dev(obia_ctx *ctx, const float *img_hwc, int H, int W, int C,
                                  const uint8_t *mask, const obia_slic_params *params,
                                  int32_t *labels_pre_out, int *n_centroids_out);
int obia_enforce_connectivity_i32_dev(obia_ctx *ctx, const int32_t *labels_in, int H, int W,
                                      int min_size, int max_size, int start_label,
                                      int32_t *labels_out, int *n_labels_out);

/* B1 with caller-supplied initial centroids instead of the library's seeding rule: the output of scikit-image's
 * own `_get_mask_centroids(mask, n_segments)` (slic_superpixels.py:14-68: RandomState / kmeans2 / pdist -- RNG and
 * version dependent, so the library does not restate it) or of `_get_grid_centroids` (:71-104).  Everything after
 * the seeding is the reference's: `step = max(steps)` (:288), spatial-only pre-pass when a mask is given (:310-314),
 * main pass, connectivity with segment_size = mask.sum() / n_centroids (:321-326).  This is how the maskSLIC path
 * every tile of create_tiled_segments takes (tiling.py:121-143) is pinned on scikit-image output.
 *   seeds->yx        HOST pointer, n x (y, x) float64 centroid positions in pixel coordinates
 *   seeds->steps_zyx the `steps` array returned with them (depth axis first; 1.0 for a 2-D image)
 *   params->n_segments is not used for seeding (K = seeds->n)
 *   stage 0: final labels (n_out = number of labels); 1: labels before connectivity (n_out = K).               */
typedef struct obia_slic_seeds {
    const double *yx;
    double steps_zyx[3];
    int32_t n;
    int32_t reserved;
} obia_slic_seeds;
int obia_slic_seeded_f32_dev(obia_ctx *ctx, const float *img_hwc, int H, int W, int C, const uint8_t *mask,
                             const obia_slic_params *params, const obia_slic_seeds *seeds, int stage,
                             int32_t *labels_out, int *n_out);

/* ---- B2: zonal-statistics operator ----------------------------------------------------------------
 * Replaces the per-segment loop crop_image_to_bbox -> mask_image_with_polygon ->
 * calculate_spectral_stats (segment_statistics.py:475-491, :143-172; utils/utils.py:37-67), batched
 * over all segments: "pixels inside polygon p" == "pixels carrying label p" (SURVEY.md 3.3).
 *   raw     (H,W,C) float32 RAW (un-normalised) raster
 *   labels  (H,W) int32; labels outside [start_label, start_label+n_labels) are ignored
 *   bands   n_bands band indices (nullable = all C bands)
 *   outputs: count[n_labels] int64; mean/var [n_labels*n_bands] float64 (var: ddof 0);
 *            min/max [n_labels*n_bands] float32.  Empty label -> NaN (segment_statistics.py:150-162). */
int obia_zonal_stats_f32(obia_ctx *ctx, const float *raw_hwc, const int32_t *labels_hw, int H, int W, int C,
                         const int32_t *bands, int n_bands, int n_labels, int start_label,
                         int64_t *count_out, double *mean_out, double *var_out, float *min_out, float *max_out);
int obia_zonal_stats_f32_dev(obia_ctx *ctx, const float *raw_hwc, const int32_t *labels_hw, int H, int W, int C,
                             const int32_t *bands, int n_bands, int n_labels, int start_label,
                             int64_t *count_out, double *mean_out, double *var_out, float *min_out, float *max_out);

/* ---- B2 (next row f2): skewness and kurtosis per (label, band) ---------------------------------------------------
 * Completes calculate_spectral_stats (segment_statistics.py:173-175: scipy.stats.skew / kurtosis with their defaults,
 * bias=True, fisher=True; NaN for nearly constant data as scipy >= 1.9 does, with float32 eps).  Second pass over
 * (labels, raw) with the per-label means as pivots (central power sums in float64).
 *   _dev : mean_dev = the mean table of obia_zonal_stats_f32_dev [n_labels*n_bands]; outputs on the device
 *   host : runs both passes itself; outputs [n_labels*n_bands] float64 on the host                              */
int obia_zonal_moments_f32_dev(obia_ctx *ctx, const float *raw_hwc, const int32_t *labels_hw, int H, int W, int C,
                               const int32_t *bands, int n_bands, int n_labels, int start_label,
                               const double *mean_dev, double *skew_out, double *kurt_out);
int obia_zonal_moments_f32(obia_ctx *ctx, const float *raw_hwc, const int32_t *labels_hw, int H, int W, int C,
                           const int32_t *bands, int n_bands, int n_labels, int start_label,
                           double *skew_out, double *kurt_out);

/* ---- next row f1: label raster -> polygon rings ------------------------------------------------------------------
 * Replaces the vectorisation loop of create_segments (segment_boundaries.py:59-77: per segment id a full-raster
 * mask + rasterio.features.shapes / GDAL polygonize, 4-connected) with one pass over the label raster.
 *   labels      (H,W) int32 on the device; labels < start_label (masked pixels, -1 / 0) get no polygon
 *   rings       every closed chain of pixel edges around a label, the label on the RIGHT of the direction of travel
 *               (exterior rings clockwise on screen, holes counter-clockwise), in raster order of the ring's smallest
 *               corner; ring r owns vertices [ring_offset[r], ring_offset[r+1]) of xy
 *   xy          (x, y) int32 pixel-CORNER coordinates, (0,0) = top-left corner of the raster; vertices only where the
 *               direction changes; first vertex repeated at the end.  Map coordinates = affine * (x, y).
 *   ring_label  label of the ring; ring_is_hole 1 for an interior ring of that label.
 * Two calls: _count returns the sizes, _rings fills caller-allocated DEVICE buffers of at least that capacity
 * (ring_offset holds n_rings + 1 entries).  A label is one 4-connected component (the output of B1 / B3).          */
int obia_polygon_count_i32_dev(obia_ctx *ctx, const int32_t *labels_hw, int H, int W, int start_label,
                               int64_t *n_rings_out, int64_t *n_vertices_out);
int obia_polygon_rings_i32_dev(obia_ctx *ctx, const int32_t *labels_hw, int H, int W, int start_label,
                               int64_t cap_rings, int64_t cap_vertices, int32_t *ring_label, uint8_t *ring_is_hole,
                               int64_t *ring_offset, int32_t *xy, int64_t *n_rings_out, int64_t *n_vertices_out);

/* ---- next row f4: consumers of the label raster -----------------------------------------------------------------
 * obia_label_edges_u8_dev   : `slic_edge` (obia/utils/cost.py:44-48): edge[y][x] = 1 when the label differs from the
 *                             pixel below or from the pixel to the right; n_edge_out = number of edge pixels (the
 *                             percentile normalisation of cost.py:21-26 on a 0/1 image only needs that count).
 * obia_sample_labels_i32_dev: the point-in-segment join of `label_segments` (obia/utils/utils.py:12-34) on a label
 *                             raster.  inverse_affine6 = [a, b, d, e, xoff, yoff] (HOST pointer) maps map coordinates
 *                             to pixel-corner coordinates (col = a*X + b*Y + xoff, row = d*X + e*Y + yoff); the point
 *                             takes the label of pixel (floor(row), floor(col)), `outside_value` outside the raster.
 *                             points_xy [n][2] float64 and labels_out [n] are DEVICE pointers.                     */
int obia_label_edges_u8_dev(obia_ctx *ctx, const int32_t *labels_hw, int H, int W, uint8_t *edge_out_hw, int64_t *n_edge_out);
int obia_sample_labels_i32_dev(obia_ctx *ctx, const int32_t *labels_hw, int H, int W, const double *inverse_affine6,
                               const double *points_xy, int64_t n_points, int outside_value, int32_t *labels_out);

/* ---- next row f3: GLCM texture statistics per (label, band) ------------------------------------------------------
 * Restates calculate_textural_stats (segment_statistics.py:179-298) on the masked bounding-box crop of every segment,
 * for the band PLANE the code evidently means (the reference indexes a column, :214; see oracle/glcm.py): crop of the
 * band to the segment's bounding box, 0 outside the segment and at NaN pixels, uint8((v-min)/(max-min)*255) over that
 * crop, GLCM at distance 2 in four directions, 256 levels, symmetric, normed, scikit-image's greycoprops, mean over the
 * four angles.  out6 [6][n_labels][n_bands] float64 (device): contrast, dissimilarity, homogeneity, ASM, energy,
 * correlation; NaN for an empty label or a band without a valid pixel.                                              */
int obia_texture_stats_f32_dev(obia_ctx *ctx, const float *raw_hwc, const int32_t *labels_hw, int H, int W, int C,
                               const int32_t *bands, int n_bands, int n_labels, int start_label, double *out6);

/* ---- B1': quickshift (the alternate method of create_segments, segment_boundaries.py:48-49) ------------------
 * Replaces `segments = quickshift(img_to_segment, **kwargs)`: skimage _quickshift.py:59-74 + _quickshift_cy.pyx.
 * Arithmetic is float64, the dtype of the pinned scikit-image 0.18.3 kernel.  tie_noise_hw: the (H,W) float64
 * noise scikit-image adds to the densities, RandomState(random_seed).normal(scale=1e-5) -- generated by the host
 * (NumPy's legacy stream is stable); NULL = no noise.  labels_out: consecutive ids from 0 in ascending order of the
 * root pixel (np.unique(...)[1]).  Up to 16 bands, any kernel_size >= 1 (1 / 3 / 4 bands with kernel_size <= 5 -- the
 * reference's usual calls -- take the LDS-staged kernel, everything else the same arithmetic on global memory).
 * sigma (ABI 2): scikit-image's Gaussian pre-smoothing, `ndi.gaussian_filter(image, [sigma, sigma, 0])` on the float64 image after
 * the Lab conversion and before `* ratio`; 0 = none.                                                                        */
int obia_quickshift_f32(obia_ctx *ctx, const float *img_hwc, int H, int W, int C, double ratio, double kernel_size,
                        double max_dist, double sigma, int convert2lab, const double *tie_noise_hw, int normalize_bands,
                        int32_t *labels_out, int *n_labels_out);
int obia_quickshift_f32_dev(obia_ctx *ctx, const float *img_hwc, int H, int W, int C, double ratio, double kernel_size,
                            double max_dist, double sigma, int convert2lab, const double *tie_noise_hw, int normalize_bands,
                            int32_t *labels_out, int *n_labels_out);

/* ---- B3: tiled driver ------------------------------------------------------------------------------
 * Replaces the tile loops of create_tiled_segments (tiling.py:103-291) on label rasters: pass 1
 * "black" checkerboard tiles on exact windows, pass 2 "white" tiles on windows grown by `buffer`,
 * existing segments wholly inside a grown window (minus the two bottom corner squares) are erased
 * and re-segmented, straddling ones are kept and masked out (tiling.py:205-260); ids 1..N at the
 * end (tiling.py:289-290).  n_segments per tile: params->n_segments scaled by valid area if > 0,
 * else the reference's crown rule round(valid_px * pixel_area / (pi * crown_radius^2))
 * (tiling.py:126-135).
 *   row0/rows: this call processes tile rows whose first pixel row lies in [row0, row0+rows) of the
 *   full H-row raster (multi-GPU slabs); pass 0,H for the whole raster.                               */
typedef struct obia_tiling_params {
    double crown_radius;
    double pixel_width;    /* |geotransform[1]| */
    double pixel_height;   /* |geotransform[5]| */
    int32_t tile_size;
    int32_t buffer;
    int32_t white_order;   /* 0: white tiles in the reference's raster order (tile-row by tile-row);
                              1: two parity classes of tile rows (even rows, then odd rows) -- the order
                              the sharded driver needs so that neighbouring slabs never touch the same seam
                              at once; both orders give the same kind of result, they differ in who wins
                              the 2*buffer x 2*buffer corner overlaps of diagonal white neighbours          */
    int32_t reserved;
} obia_tiling_params;

int obia_tiled_slic_f32_dev(obia_ctx *ctx, const float *img_hwc, const uint8_t *mask, int H, int W, int C,
                            const obia_tiling_params *tiling, const obia_slic_params *params,
                            int32_t *labels_out, int64_t *n_segments_out);
int obia_tiled_slic_f32(obia_ctx *ctx, const float *img_hwc, const uint8_t *mask, int H, int W, int C,
                        const obia_tiling_params *tiling, const obia_slic_params *params,
                        int32_t *labels_out, int64_t *n_segments_out);

/* ---- B3, sharded: the same tile loops as a session, for slabs of a raster spread over several GPUs --------
 * The caller holds rows [row0, row0 + H_local) of a (H_global, W) raster on this GPU: its slab plus the halo
 * rows its white windows reach (`buffer` rows, +1 label row so that "segment continues beyond the halo" can be
 * seen).  labels_local (same rows) is the persistent label raster G of the session: provisional ids 1..next_id-1,
 * 0 = no segment.  Between passes the host exchanges halo rows of G with the neighbouring ranks (RCCL send/recv)
 * and registers the segments it imported with obia_tiler_set_segments (their pixel counts as seen locally;
 * 0xffffffff for a segment that continues beyond the halo and can therefore never be "within" a window).
 * tile rows are GLOBAL indices; row_parity -1 = all rows, 0/1 = rows of that parity (white_order 1).
 * The context must not be used for other calls while a session is open.                                         */
typedef struct obia_tiler obia_tiler;
obia_tiler *obia_tiler_create(obia_ctx *ctx, const float *img_local, const uint8_t *mask_local, int H_local, int W, int C,
                              int H_global, int row0, const obia_tiling_params *tiling, const obia_slic_params *params,
                              int32_t *labels_local, int extra_ids);
void obia_tiler_destroy(obia_tiler *t);
int obia_tiler_run(obia_tiler *t, int white, int tile_row_lo, int tile_row_hi, int row_parity);
int obia_tiler_next_id(obia_tiler *t);
int obia_tiler_set_segments(obia_tiler *t, int first_id, int count, const uint32_t *sizes_dev);
/* alive flags of the provisional ids [0, count): 1 = the segment exists, 0 = dropped (it was `within` a white
 * window) or never created.  A rank that dropped a neighbour's segment tells the owner, which clears the flag. */
int obia_tiler_get_alive(obia_tiler *t, uint8_t *alive_out_dev, int count);
int obia_tiler_set_alive(obia_tiler *t, const uint8_t *alive_in_dev, int count);
int obia_tiler_finalize(obia_tiler *t, int64_t *n_segments_out);
/* Import of a seam (round 4): the boundary label rows a neighbouring rank sent, as wire codes, become local ids in ONE call --
 * the step between `ncclRecv` and the next pass (semantic anchor obia/utils/tiling.py:289-290: one id space).
 *   codes_dev [n]        int32: (owner_rank + 1) << 24 | the owner's local id; 0 = no segment
 *   fmap_dev [fmap_cap]  int32, persistent per (session, owner): the owner's local id -> my local id, 0 = not imported yet
 *   code_of_dev [cap]    int32, persistent per session: my local id -> wire code of an imported segment (0: one of my own);
 *                        the caller keeps cap above obia_tiler_next_id() + n
 *   ids_out_dev [n]      int32: my local ids (codes of `my_rank` map to their id field, codes of other owners to 0)
 * Codes of `owner_rank` that are not in fmap yet get consecutive new local ids in ascending order of the owner's ids, starting
 * at *first_new_out = obia_tiler_next_id(); the range is registered like obia_tiler_set_segments(first, n_new, 0xffffffff...)
 * (sizes follow from the caller's view of its halo).  *max_owner_id_out = the largest owner id on the seam: when it is
 * >= fmap_cap the ids beyond the map were left out (ids_out 0) and the caller calls again with a larger map (entries kept; what
 * the first call imported -- *n_new_out ids from *first_new_out -- stays imported).                                          */
int obia_tiler_import_seam(obia_tiler *t, const int32_t *codes_dev, int n, int my_rank, int owner_rank,
                           int32_t *fmap_dev, int fmap_cap, int32_t *code_of_dev, int32_t *ids_out_dev,
                           int *first_new_out, int *n_new_out, int *max_owner_id_out);

/* ---- measurement hooks ------------------------------------------------------------------------------
 * Time of the most recent call's kernels by class, measured with HIP events on the context's
 * stream (bench.py's roofline leg).  `what`: 0 = SLIC colour sweeps (sum of launches, ms), 1 = number of
 * those launches, 2 = feature preparation, 3 = connectivity, 4 = zonal statistics, 5 = whole call,
 * 6 = maskSLIC spatial-only pre-pass sweeps (ms), 7 = pixels actually processed by the launches of 0
 * (sum; tiles skipped by exit_on_fixed_point are not counted), 8 = the same for the pre-pass launches,
 * 9 = pixels of the launches of 0 that also stored their labels (only the last sweep of a batch does),
 * 10 / 11 = time during which at least one colour / pre-pass sweep was running (equals 0 / 6 unless the batch's problems run
 * as groups on side streams, OBIA_SWEEP_GROUPS).  The events of a sweep are bound to its dispatch (hipExtLaunchKernelGGL): 0 and
 * 6 are sums of the kernels' own start-to-end times, as a rocprofv3 kernel trace reports them.
 * 12 = batches of the call whose sweeps ran a second time with every sweep storing its labels (a valid pixel that no window
 * reached keeps the label of the sweep before: DESIGN.md 3.2 item 5) -- counted whether profiling is on or not.
 * `enabled`: 0 off, 1 every class, 2 only the colour sweeps (an event pair costs ~2.5 us of stream time: with all classes on,
 * a step of the headline workload records ~420 pairs = 1.1 ms; bench.py times its steps in mode 2).                       */
int obia_set_profiling(obia_ctx *ctx, int enabled);
double obia_last_timing(obia_ctx *ctx, int what);

#ifdef __cplusplus
}
#endif
#endif /* OBIA_HIP_H */
