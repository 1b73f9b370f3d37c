// zonal.hip -- per-segment zonal statistics (count, mean, variance, min, max per band) on gfx950.
//
// Replaces the per-segment loop of obia create_objects (segment_statistics.py:475-491):
//   crop_image_to_bbox + mask_image_with_polygon (utils/utils.py:37-67)  -> "pixels of label p"
//   calculate_spectral_stats (segment_statistics.py:143-172)            -> np.mean / np.var / np.min / np.max
// One pass over (labels, raw raster); kernel shape described at zonal_kernel.  Sums are float64 (sum, sum of
// squares); variance = E[x^2] - E[x]^2 in float64 meets the 1e-5 relative tolerance for uint16-range rasters.
#include "slic.hpp"

namespace obia {

#ifndef ZW
#define ZW 6
#endif
#ifndef ZR
#define ZR 2
#endif
#ifndef ZTW
#define ZTW 128   /* columns of a tile (ZTH rows) */
#endif
#ifndef ZTH
#define ZTH 64
#endif
#ifndef ZSLOTS
#define ZSLOTS 64
#endif
constexpr int Z_MAXB = 16, Z_ROWS = ZR;

struct BandList { int n; int identity; int b[Z_MAXB]; };   // identity: b[i] == i for every i < n
// (the bands of a lane: 16 or 12 bytes at a 4-byte aligned address -- gfx950 loads a dwordx4 / dwordx3 from any dword address, the
// `aligned(4)` of the vector types only tells the compiler so)
typedef float z_v4f __attribute__((ext_vector_type(4), aligned(4)));
typedef float z_v3f __attribute__((ext_vector_type(3), aligned(4)));
// the BPL consecutive bands of a lane: ONE nontemporal load (the raster is read once) of 16 or 12 bytes at dword alignment
template <int BPL> __device__ __forceinline__ void ld_raster(const float *p, float *v);
template <> __device__ __forceinline__ void ld_raster<4>(const float *p, float *v) {
    const z_v4f t = __builtin_nontemporal_load(reinterpret_cast<const z_v4f *>(p));
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld_raster<3>(const float *p, float *v) {
    const z_v3f t = __builtin_nontemporal_load(reinterpret_cast<const z_v3f *>(p));
    v[0] = t.x; v[1] = t.y; v[2] = t.z;
}
// min / max of two numbers that are known not to be NaN: ONE instruction (fminf / fmaxf first quiet both operands)
__device__ __forceinline__ float zmin(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -INFINITY); }
__device__ __forceinline__ float zmax(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, INFINITY); }

__device__ __forceinline__ unsigned zkey(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float zunkey(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// One workgroup per ZTW x ZTH tile.  A lane owns FOUR bands of ONE pixel column: lane = (column, band quad), LPP = NBP / 4
// lanes per pixel, so the 64 lanes of a wave read 64 consecutive 16-byte chunks of a raster row (one 1-KB request) and a
// workgroup of ZTW * LPP / 64 waves covers the ZTW columns.  Every lane walks its column down the rows of the tile:
//   per lane : runs of equal label down the column are summed in registers (sum, sum of squares in double; min, max); the
//              loads of the next Z_ROWS rows are in flight while the current ones are folded
//   per run  : the partial goes to a ZSLOTS-slot LDS hash table keyed by label (native ds_add_f64 / ds_min_f32 / ds_max_f32)
//   per tile : one global atomic per (tile, label, band, statistic) at the end.  A tile that holds more than ZSLOTS labels
//              sends the overflow straight to global memory.
// The walk is bound by instruction issue as much as by memory (round 3, PMC: 95 vector + 44 scalar instructions per wave and
// row in the first version, the vector units busy half of the time at 4.5 waves per SIMD), so the row loop is kept lean:
//   * addresses are (uniform row base) + (a lane offset that never changes): no per-row address arithmetic on the vector
//     unit; columns / rows past the raster are clamped to the last one and their label forced to -1, so no branch guards a load
//   * the raster is read whatever the label says (a load that waits for the label is two round trips in a row)
//   * one test per row for "any NaN among my four values"; the clean path has no per-band conditions
// MODE 0: the band list is 0..C-1 in order and C == NBP -- every lane reads ONE 16-byte chunk per row;
// MODE 1: the band list is 0..C-1 in order, C not a multiple of four (the author's rasters hold nine bands): full quads are one
//         16-byte load at dword alignment (gfx950 loads a dwordx4 from any dword address), the lanes of the last quad load their
//         1..3 bands one by one;   MODE 2: the lanes gather their bands through the list (a subset, any order).
template <int LPP, int BPL, int MODE>
__global__ __launch_bounds__(ZTW * LPP) __attribute__((amdgpu_waves_per_eu(ZW, ZW))) void zonal_kernel(const float *__restrict__ raw, const int32_t *__restrict__ labels,
                                                         int H, int W, int C, BandList bl, int n_labels, int start_label,
                                                         unsigned *__restrict__ g_cnt, unsigned *__restrict__ g_nan,
                                                         double *__restrict__ g_sum, double *__restrict__ g_sq,
                                                         unsigned *__restrict__ g_mn, unsigned *__restrict__ g_mx) {
    constexpr int NBP = LPP * BPL, NT = ZTW * LPP;   // band slots of a pixel, lanes of the workgroup
    __shared__ int s_key[ZSLOTS];
    __shared__ unsigned s_cnt[ZSLOTS];
    __shared__ unsigned s_nan[ZSLOTS][NBP];
    __shared__ double s_sum[ZSLOTS][NBP], s_sq[ZSLOTS][NBP];
    __shared__ float s_mn[ZSLOTS][NBP], s_mx[ZSLOTS][NBP];   // (+inf, -inf) = nothing but NaNs so far
    const int tid = threadIdx.x;
    const int nb = bl.n;
    for (int i = tid; i < ZSLOTS; i += NT) { s_key[i] = -1; s_cnt[i] = 0; }
    for (int i = tid; i < ZSLOTS * NBP; i += NT) {
        (&s_sum[0][0])[i] = 0.0; (&s_sq[0][0])[i] = 0.0; (&s_nan[0][0])[i] = 0u;
        (&s_mn[0][0])[i] = INFINITY; (&s_mx[0][0])[i] = -INFINITY;
    }
    __syncthreads();
    const int tiles_x = (W + ZTW - 1) / ZTW;
    const int ty0 = (blockIdx.x / tiles_x) * ZTH, tx0 = (blockIdx.x % tiles_x) * ZTW;
    const int x = tx0 + tid / LPP, q = tid % LPP;       // column, band group (BPL bands: a quad, or a triple for 3 / 6 / 9 bands)
    const int nbq = min(BPL, max(0, nb - BPL * q));     // bands of this lane's group that exist (>= 1: NBP is the smallest that holds nb)
    const bool col_ok = x < W;
    const int xc = col_ok ? x : W - 1;                  // a column past the raster reads the last one (and is given label -1)
    // byte offsets inside a raster row / a label row: fixed for the whole walk (a row of the raster stays below 4 GB)
    const unsigned loff = (unsigned)xc * 4u;
    unsigned roff[BPL];
#pragma unroll
    for (int b = 0; b < BPL; ++b) {   // constant indices only: the band list stays in scalar registers
        const int band = MODE < 2 ? BPL * q + b : ((q == 0) ? bl.b[b] : (q == 1) ? bl.b[4 + b] : (q == 2) ? bl.b[8 + b] : bl.b[12 + b]);   // (MODE 2 comes with BPL == 4)
        roff[b] = ((unsigned)xc * (unsigned)C + (unsigned)(b < nbq ? band : 0)) * 4u;
    }
    const bool full = MODE == 0 || (MODE == 1 && nbq == BPL);

    int c_lab = -2, c_slot = -1;                        // one-entry cache of the last label -> slot lookup
    auto find_slot = [&](int l) -> int {
        if (l == c_lab) return c_slot;
        const unsigned h = ((unsigned)l * 2654435761u) >> 16;
        int found = -1;
#pragma unroll 1
        for (int probe = 0; probe < ZSLOTS; ++probe) {
            const int sidx = (h + probe) % ZSLOTS;
            const int old = atomicCAS(&s_key[sidx], -1, l);
            if (old == -1 || old == l) { found = sidx; break; }
        }
        c_lab = l; c_slot = found;
        return found;
    };

    int rl = -1;
    unsigned rn = 0;
    double rs[BPL], rq[BPL];
    float rmn[BPL], rmx[BPL];
#pragma unroll
    for (int b = 0; b < BPL; ++b) { rs[b] = 0.0; rq[b] = 0.0; rmn[b] = INFINITY; rmx[b] = -INFINITY; }
    auto close_run = [&]() {
        if (rl < 0) return;
        const int slot = find_slot(rl);
        if (slot >= 0) {
            // no conditions on this path (it is executed by the whole wave whenever ONE lane closes a run): a run of NaNs only
            // adds 0 and folds (+inf, -inf), a band slot past the last band collects zeros nobody reads
            if (q == 0) atomicAdd(&s_cnt[slot], rn);
#pragma unroll
            for (int b = 0; b < BPL; ++b) {
                atomicAdd(&s_sum[slot][BPL * q + b], rs[b]);
                atomicAdd(&s_sq[slot][BPL * q + b], rq[b]);
                __hip_atomic_fetch_min(&s_mn[slot][BPL * q + b], rmn[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_min_f32
                __hip_atomic_fetch_max(&s_mx[slot][BPL * q + b], rmx[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {   // table full (more than ZSLOTS labels in one tile): straight to global memory
            if (q == 0) atomicAdd(&g_cnt[rl], rn);
#pragma unroll
            for (int b = 0; b < BPL; ++b) {
                if (b >= nbq || !(rmn[b] <= rmx[b])) continue;   // band absent, or only NaNs in this run
                const size_t o = (size_t)rl * nb + BPL * q + b;
                unsafeAtomicAdd(&g_sum[o], rs[b]);
                unsafeAtomicAdd(&g_sq[o], rq[b]);
                atomicMin(&g_mn[o], zkey(rmn[b]));
                atomicMax(&g_mx[o], zkey(rmx[b]));
            }
        }
    };

    // rows in groups of Z_ROWS: the loads of group g+1 are issued before group g is folded
    int lab[2][Z_ROWS];
    float val[2][Z_ROWS][BPL];
    // the rows are fetched in order: two uniform row pointers (scalar registers) move down the raster, a lane adds its fixed offset
    const size_t row_bytes = (size_t)W * C * 4, lab_bytes = (size_t)W * 4;
    const char *lrow = reinterpret_cast<const char *>(labels) + (size_t)ty0 * lab_bytes;
    const char *rrow = reinterpret_cast<const char *>(raw) + (size_t)ty0 * row_bytes;
    int y_next = ty0;
    auto fetch = [&](int buf) {
#pragma unroll
        for (int j = 0; j < Z_ROWS; ++j) {
            int l = *reinterpret_cast<const int *>(lrow + loff) - start_label;
            if ((unsigned)l >= (unsigned)n_labels || !col_ok || y_next >= H) l = -1;
            float v[BPL];
#pragma unroll
            for (int b = 0; b < BPL; ++b) v[b] = 0.0f;
            if (full) ld_raster<BPL>(reinterpret_cast<const float *>(rrow + roff[0]), v);
            else {
#pragma unroll
                for (int b = 0; b < BPL; ++b) if (b < nbq) v[b] = *reinterpret_cast<const float *>(rrow + roff[b]);
            }
            lab[buf][j] = l;
#pragma unroll
            for (int b = 0; b < BPL; ++b) val[buf][j][b] = v[b];
            ++y_next;
            if (y_next < H) { lrow += lab_bytes; rrow += row_bytes; }   // (uniform) a row past the raster reads the last one again
        }
    };
    constexpr int NG = ZTH / Z_ROWS;
    const int y_end = min(ty0 + ZTH, H);
    fetch(0);
#pragma unroll 1
    for (int g = 0; g < NG; g += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int y0 = ty0 + (g + half) * Z_ROWS;
            if (y0 >= y_end) break;                      // workgroup-uniform
            if (g + half + 1 < NG) fetch(half ^ 1);   // rows past the raster come back as label -1
#pragma unroll
            for (int j = 0; j < Z_ROWS; ++j) {
                const int l = lab[half][j];
                if (l != rl) {
                    close_run();
                    rl = l; rn = 0;
#pragma unroll
                    for (int b = 0; b < BPL; ++b) { rs[b] = 0.0; rq[b] = 0.0; rmn[b] = INFINITY; rmx[b] = -INFINITY; }
                }
                if (l < 0) continue;
                rn += 1;
                const float *v = val[half][j];
                bool clean = true;
#pragma unroll
                for (int b = 0; b < BPL; ++b) clean &= (v[b] == v[b]);
                if (clean) {   // (a band slot past the last band holds 0)
#pragma unroll
                    for (int b = 0; b < BPL; ++b) {
                        const double dv = (double)v[b];
                        rs[b] += dv; rq[b] = fma(dv, dv, rq[b]);
                        rmn[b] = zmin(rmn[b], v[b]); rmx[b] = zmax(rmx[b], v[b]);
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < BPL; ++b) {
                        if (v[b] == v[b]) {
                            const double dv = (double)v[b];
                            rs[b] += dv; rq[b] = fma(dv, dv, rq[b]);
                            rmn[b] = zmin(rmn[b], v[b]); rmx[b] = zmax(rmx[b], v[b]);
                        } else {
                            // NaN pixels are dropped per band (`band[~isnan]`, segment_statistics.py:145-147): remember how
                            // many, the per-band count is (label count - NaN count).  Rare: direct atomics.
                            const int slot = find_slot(l);
                            if (slot >= 0) atomicAdd(&s_nan[slot][BPL * q + b], 1u);
                            else if (b < nbq) atomicAdd(&g_nan[(size_t)l * nb + BPL * q + b], 1u);
                        }
                    }
                }
            }
        }
    }
    close_run();
    __syncthreads();
    for (int i = tid; i < ZSLOTS * nb; i += NT) {
        const int slot = i / nb, b = i - slot * nb;
        const int l = s_key[slot];
        if (l < 0) continue;
        if (b == 0 && s_cnt[slot]) atomicAdd(&g_cnt[l], s_cnt[slot]);
        if (s_nan[slot][b]) atomicAdd(&g_nan[(size_t)l * nb + b], s_nan[slot][b]);
        if (!(s_mn[slot][b] <= s_mx[slot][b])) continue;   // only NaNs (or nothing) for this band
        unsafeAtomicAdd(&g_sum[(size_t)l * nb + b], s_sum[slot][b]);
        unsafeAtomicAdd(&g_sq[(size_t)l * nb + b], s_sq[slot][b]);
        atomicMin(&g_mn[(size_t)l * nb + b], zkey(s_mn[slot][b]));
        atomicMax(&g_mx[(size_t)l * nb + b], zkey(s_mx[slot][b]));
    }
}

__global__ void zonal_init_kernel(unsigned *g_cnt, unsigned *g_bcnt, double *g_sum, double *g_sq, unsigned *g_mn, unsigned *g_mx,
                                  long long n_labels, int nb) {
    const long long n = n_labels * nb;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        g_sum[i] = 0.0; g_sq[i] = 0.0; g_bcnt[i] = 0u; g_mn[i] = 0xffffffffu; g_mx[i] = 0u;   // g_bcnt holds NaN counts
        if (i < n_labels) g_cnt[i] = 0;
    }
}

// mean / variance divide by the number of non-NaN pixels of the band, as `band[~isnan]` does.
__global__ void zonal_finalize_kernel(const unsigned *__restrict__ g_cnt, const unsigned *__restrict__ g_bcnt,
                                      const double *__restrict__ g_sum,
                                      const double *__restrict__ g_sq, const unsigned *__restrict__ g_mn,
                                      const unsigned *__restrict__ g_mx, long long n_labels, int nb,
                                      int64_t *__restrict__ count, double *__restrict__ mean, double *__restrict__ var,
                                      float *__restrict__ mn, float *__restrict__ mx) {
    const long long n = n_labels * nb;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long l = i / nb;
        if (i % nb == 0) count[l] = (int64_t)g_cnt[l];
        const unsigned c = g_cnt[l] - g_bcnt[i];   // pixels of the label minus the NaN pixels of this band
        if (c == 0 || g_mn[i] == 0xffffffffu) {
            mean[i] = NAN; var[i] = NAN; mn[i] = NAN; mx[i] = NAN;
        } else {
            const double m = g_sum[i] / (double)c;
            double v = g_sq[i] / (double)c - m * m;
            mean[i] = m;
            var[i] = v < 0.0 ? 0.0 : v;
            mn[i] = zunkey(g_mn[i]);
            mx[i] = zunkey(g_mx[i]);
        }
    }
}

// ---- higher moments: skewness and kurtosis per (label, band) -------------------------------------------
// scipy.stats.skew / kurtosis with their defaults (bias=True, fisher=True; segment_statistics.py:173-175):
//   m_k = mean((x - mean)^k),  skewness = m3 / m2^1.5,  kurtosis = m4 / m2^2 - 3,
//   NaN where m2 <= (eps * mean)^2 (scipy >= 1.9: "nearly constant" data), eps = float32 epsilon (the raster dtype).
// Second pass of the same shape as zonal_kernel: the per-label means of the first pass are the pivots, so the power
// sums are CENTRAL (no cancellation); a lane loads the four means of its band quad once per run.
// (row loop as in zonal_kernel: uniform row pointers + fixed lane offsets, clamped columns / rows, raster read whatever the label,
// no per-band branches -- a NaN contributes a zero difference and no count.  MODE 0: all band quads full and in order; 1: in order
// with a partial last quad; 2: gathered subset.)
template <int NBP, int MODE>
__global__ __launch_bounds__(ZTW / 4 * NBP) __attribute__((amdgpu_waves_per_eu(ZW, ZW))) void zonal_moments_kernel(
    const float *__restrict__ raw, const int32_t *__restrict__ labels, int H, int W, int C, BandList bl, int n_labels,
    int start_label, const double *__restrict__ mean, unsigned *__restrict__ g_n, double *__restrict__ g_s2,
    double *__restrict__ g_s3, double *__restrict__ g_s4) {
    constexpr int LPP = NBP / 4, NT = ZTW * LPP;
    __shared__ int s_key[ZSLOTS];
    __shared__ unsigned s_n[ZSLOTS][NBP];
    __shared__ double s_s2[ZSLOTS][NBP], s_s3[ZSLOTS][NBP], s_s4[ZSLOTS][NBP];
    const int tid = threadIdx.x;
    const int nb = bl.n;
    for (int i = tid; i < ZSLOTS; i += NT) s_key[i] = -1;
    for (int i = tid; i < ZSLOTS * NBP; i += NT) {
        (&s_n[0][0])[i] = 0u; (&s_s2[0][0])[i] = 0.0; (&s_s3[0][0])[i] = 0.0; (&s_s4[0][0])[i] = 0.0;
    }
    __syncthreads();
    const int tiles_x = (W + ZTW - 1) / ZTW;
    const int ty0 = (blockIdx.x / tiles_x) * ZTH, tx0 = (blockIdx.x % tiles_x) * ZTW;
    const int x = tx0 + tid / LPP, q = tid % LPP;
    const int nbq = min(4, max(0, nb - 4 * q));
    const bool col_ok = x < W;
    const int xc = col_ok ? x : W - 1;
    const unsigned loff = (unsigned)xc * 4u;
    unsigned roff[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int band = MODE < 2 ? 4 * q + b : ((q == 0) ? bl.b[b] : (q == 1) ? bl.b[4 + b] : (q == 2) ? bl.b[8 + b] : bl.b[12 + b]);
        roff[b] = ((unsigned)xc * (unsigned)C + (unsigned)(b < nbq ? band : 0)) * 4u;
    }
    const bool full = MODE == 0 || (MODE == 1 && nbq == 4);

    int rl = -1;
    unsigned rn[4];
    double mu[4], r2[4], r3[4], r4[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) { rn[b] = 0; mu[b] = 0.0; r2[b] = 0.0; r3[b] = 0.0; r4[b] = 0.0; }
    auto close_run = [&]() {
        if (rl < 0) return;
        const unsigned h = ((unsigned)rl * 2654435761u) >> 16;
        int slot = -1;
#pragma unroll 1
        for (int probe = 0; probe < ZSLOTS; ++probe) {
            const int sidx = (h + probe) % ZSLOTS;
            const int old = atomicCAS(&s_key[sidx], -1, rl);
            if (old == -1 || old == rl) { slot = sidx; break; }
        }
        if (slot >= 0) {   // no conditions: an empty band adds zeros, a padded band slot collects zeros nobody reads
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                atomicAdd(&s_n[slot][4 * q + b], rn[b]);
                atomicAdd(&s_s2[slot][4 * q + b], r2[b]);
                atomicAdd(&s_s3[slot][4 * q + b], r3[b]);
                atomicAdd(&s_s4[slot][4 * q + b], r4[b]);
            }
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b >= nbq || rn[b] == 0) continue;
                const size_t o = (size_t)rl * nb + 4 * q + b;
                atomicAdd(&g_n[o], rn[b]);
                unsafeAtomicAdd(&g_s2[o], r2[b]);
                unsafeAtomicAdd(&g_s3[o], r3[b]);
                unsafeAtomicAdd(&g_s4[o], r4[b]);
            }
        }
    };
    int lab[2][Z_ROWS];
    float val[2][Z_ROWS][4];
    const size_t row_bytes = (size_t)W * C * 4, lab_bytes = (size_t)W * 4;
    const char *lrow = reinterpret_cast<const char *>(labels) + (size_t)ty0 * lab_bytes;
    const char *rrow = reinterpret_cast<const char *>(raw) + (size_t)ty0 * row_bytes;
    int y_next = ty0;
    auto fetch = [&](int buf) {
#pragma unroll
        for (int j = 0; j < Z_ROWS; ++j) {
            int l = *reinterpret_cast<const int *>(lrow + loff) - start_label;
            if ((unsigned)l >= (unsigned)n_labels || !col_ok || y_next >= H) l = -1;
            float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (full) ld_raster<4>(reinterpret_cast<const float *>(rrow + roff[0]), v);
            else {
#pragma unroll
                for (int b = 0; b < 4; ++b) if (b < nbq) v[b] = *reinterpret_cast<const float *>(rrow + roff[b]);
            }
            lab[buf][j] = l;
#pragma unroll
            for (int b = 0; b < 4; ++b) val[buf][j][b] = v[b];
            ++y_next;
            if (y_next < H) { lrow += lab_bytes; rrow += row_bytes; }
        }
    };
    constexpr int NG = ZTH / Z_ROWS;
    const int y_end = min(ty0 + ZTH, H);
    fetch(0);
#pragma unroll 1
    for (int g = 0; g < NG; g += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int y0 = ty0 + (g + half) * Z_ROWS;
            if (y0 >= y_end) break;
            if (g + half + 1 < NG) fetch(half ^ 1);
#pragma unroll
            for (int j = 0; j < Z_ROWS; ++j) {
                const int l = lab[half][j];
                if (l != rl) {
                    close_run();
                    rl = l;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        rn[b] = 0; r2[b] = 0.0; r3[b] = 0.0; r4[b] = 0.0;
                        mu[b] = (l >= 0 && b < nbq) ? mean[(size_t)l * nb + 4 * q + b] : 0.0;
                    }
                }
                if (l < 0) continue;
                const float *v = val[half][j];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const bool ok = v[b] == v[b];                 // NaN pixels are dropped per band: no count, a zero difference
                    const double d = ok ? (double)v[b] - mu[b] : 0.0;
                    const double d2 = d * d;
                    rn[b] += ok ? 1u : 0u; r2[b] += d2; r3[b] = fma(d2, d, r3[b]); r4[b] = fma(d2, d2, r4[b]);
                }
            }
        }
    }
    close_run();
    __syncthreads();
    for (int i = tid; i < ZSLOTS * nb; i += NT) {
        const int slot = i / nb, b = i - slot * nb;
        const int l = s_key[slot];
        if (l < 0 || s_n[slot][b] == 0) continue;
        const size_t o = (size_t)l * nb + b;
        atomicAdd(&g_n[o], s_n[slot][b]);
        unsafeAtomicAdd(&g_s2[o], s_s2[slot][b]);
        unsafeAtomicAdd(&g_s3[o], s_s3[slot][b]);
        unsafeAtomicAdd(&g_s4[o], s_s4[slot][b]);
    }
}

__global__ void zonal_moments_finalize_kernel(const unsigned *__restrict__ g_n, const double *__restrict__ g_s2,
                                              const double *__restrict__ g_s3, const double *__restrict__ g_s4,
                                              const double *__restrict__ mean, long long n, double *__restrict__ skew,
                                              double *__restrict__ kurt) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned c = g_n[i];
        double sk = NAN, ku = NAN;
        if (c > 0) {
            const double m2 = g_s2[i] / (double)c, m3 = g_s3[i] / (double)c, m4 = g_s4[i] / (double)c;
            const double t = 1.1920928955078125e-07 * mean[i];      // float32 eps * mean
            if (!(m2 <= t * t)) {
                sk = m3 / (m2 * sqrt(m2));
                ku = m4 / (m2 * m2) - 3.0;
            }
        }
        skew[i] = sk; kurt[i] = ku;
    }
}

int zonal_moments_dev(obia_ctx *ctx, const float *raw, const int32_t *labels, int H, int W, int C,
                      const int32_t *bands_host, int n_bands, int n_labels, int start_label, const double *mean,
                      double *skew, double *kurt) {
    ScopedSpan span(ctx, T_ZONAL);
    if (H <= 0 || W <= 0 || C <= 0 || n_labels < 0) { set_error("bad zonal_moments shape"); return OBIA_E_INVALID; }
    BandList bl;
    if (bands_host == nullptr) {
        if (C > Z_MAXB) { set_error("more than %d bands not supported", Z_MAXB); return OBIA_E_UNSUPPORTED; }
        bl.n = C;
        for (int i = 0; i < C; ++i) bl.b[i] = i;
    } else {
        if (n_bands < 1 || n_bands > Z_MAXB) { set_error("n_bands %d out of range (1..%d)", n_bands, Z_MAXB); return OBIA_E_UNSUPPORTED; }
        bl.n = n_bands;
        for (int i = 0; i < n_bands; ++i) {
            if (bands_host[i] < 0 || bands_host[i] >= C) { set_error("band index %d out of range (0..%d)", bands_host[i], C - 1); return OBIA_E_INVALID; }
            bl.b[i] = bands_host[i];
        }
    }
    for (int i = bl.n; i < Z_MAXB; ++i) bl.b[i] = 0;
    bl.identity = 1;
    for (int i = 0; i < bl.n; ++i) if (bl.b[i] != i) bl.identity = 0;
    if (n_labels == 0) return OBIA_OK;
    Arena &A = ctx->arena;
    const size_t nlb = (size_t)n_labels * bl.n;
    unsigned *g_n = A.get<unsigned>(nlb);
    double *g_s2 = A.get<double>(nlb), *g_s3 = A.get<double>(nlb), *g_s4 = A.get<double>(nlb);
    if (!g_n || !g_s2 || !g_s3 || !g_s4) return OBIA_E_NOMEM;
    OBIA_HIP_TRY(hipMemsetAsync(g_n, 0, sizeof(unsigned) * nlb, ctx->stream));
    OBIA_HIP_TRY(hipMemsetAsync(g_s2, 0, sizeof(double) * nlb, ctx->stream));
    OBIA_HIP_TRY(hipMemsetAsync(g_s3, 0, sizeof(double) * nlb, ctx->stream));
    OBIA_HIP_TRY(hipMemsetAsync(g_s4, 0, sizeof(double) * nlb, ctx->stream));
    if ((long long)W * C * 4 >= (1ll << 32)) { set_error("a raster row of %lld bytes is not supported (4 GB at most)", (long long)W * C * 4); return OBIA_E_UNSUPPORTED; }
    const int tiles = cdiv(W, ZTW) * cdiv(H, ZTH);
    const bool ident = bl.identity && bl.n == C;
#define LAUNCH_ZM_MODE(NBPV, MODEV)                                                                                  \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(zonal_moments_kernel<NBPV, MODEV>), dim3(tiles), dim3(ZTW / 4 * NBPV), 0, ctx->stream, raw, \
                       labels, H, W, C, bl, n_labels, start_label, mean, g_n, g_s2, g_s3, g_s4)
#define LAUNCH_ZM(NBPV)                                                                                              \
    do {                                                                                                             \
        if (ident && bl.n == NBPV) LAUNCH_ZM_MODE(NBPV, 0);                                                          \
        else if (ident) LAUNCH_ZM_MODE(NBPV, 1);                                                                     \
        else LAUNCH_ZM_MODE(NBPV, 2);                                                                                \
    } while (0)
    if (bl.n <= 4) LAUNCH_ZM(4);
    else if (bl.n <= 8) LAUNCH_ZM(8);
    else if (bl.n <= 12) LAUNCH_ZM(12);
    else LAUNCH_ZM(16);
#undef LAUNCH_ZM
#undef LAUNCH_ZM_MODE
    int ib = cdiv((long long)nlb, 256);
    if (ib > 4096) ib = 4096;
    hipLaunchKernelGGL(zonal_moments_finalize_kernel, dim3(ib), dim3(256), 0, ctx->stream, g_n, g_s2, g_s3, g_s4, mean,
                       (long long)nlb, skew, kurt);
    OBIA_HIP_TRY(hipGetLastError());
    return OBIA_OK;
}

int zonal_stats_dev(obia_ctx *ctx, const float *raw, const int32_t *labels, int H, int W, int C,
                    const int32_t *bands_host, int n_bands, int n_labels, int start_label, int64_t *count,
                    double *mean, double *var, float *mn, float *mx) {
    ScopedSpan span(ctx, T_ZONAL);
    if (H <= 0 || W <= 0 || C <= 0 || n_labels < 0) { set_error("bad zonal_stats shape"); return OBIA_E_INVALID; }
    BandList bl;
    if (bands_host == nullptr) {
        if (C > Z_MAXB) { set_error("more than %d bands not supported", Z_MAXB); return OBIA_E_UNSUPPORTED; }
        bl.n = C;
        for (int i = 0; i < C; ++i) bl.b[i] = i;
    } else {
        if (n_bands < 1 || n_bands > Z_MAXB) { set_error("n_bands %d out of range (1..%d)", n_bands, Z_MAXB); return OBIA_E_UNSUPPORTED; }
        bl.n = n_bands;
        for (int i = 0; i < n_bands; ++i) {
            if (bands_host[i] < 0 || bands_host[i] >= C) { set_error("band index %d out of range (0..%d)", bands_host[i], C - 1); return OBIA_E_INVALID; }
            bl.b[i] = bands_host[i];
        }
    }
    for (int i = bl.n; i < Z_MAXB; ++i) bl.b[i] = 0;
    bl.identity = 1;
    for (int i = 0; i < bl.n; ++i) if (bl.b[i] != i) bl.identity = 0;
    if ((long long)W * C * 4 >= (1ll << 32)) { set_error("a raster row of %lld bytes is not supported (4 GB at most)", (long long)W * C * 4); return OBIA_E_UNSUPPORTED; }
    if (n_labels == 0) return OBIA_OK;
    Arena &A = ctx->arena;
    const size_t nl = (size_t)n_labels, nlb = nl * bl.n;
    unsigned *g_cnt = A.get<unsigned>(nl), *g_bcnt = A.get<unsigned>(nlb);
    double *g_sum = A.get<double>(nlb), *g_sq = A.get<double>(nlb);
    unsigned *g_mn = A.get<unsigned>(nlb), *g_mx = A.get<unsigned>(nlb);
    if (!g_cnt || !g_bcnt || !g_sum || !g_sq || !g_mn || !g_mx) return OBIA_E_NOMEM;
    int ib = cdiv((long long)nlb, 256);
    if (ib > 4096) ib = 4096;
    hipLaunchKernelGGL(zonal_init_kernel, dim3(ib), dim3(256), 0, ctx->stream, g_cnt, g_bcnt, g_sum, g_sq, g_mn, g_mx, (long long)n_labels, bl.n);
    const int tiles = cdiv(W, ZTW) * cdiv(H, ZTH);
    const bool ident = bl.identity && bl.n == C;
#define LAUNCH_ZONAL_MODE(LPPV, BPLV, MODEV)                                                                        \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(zonal_kernel<LPPV, BPLV, MODEV>), dim3(tiles), dim3(ZTW * LPPV), 0, ctx->stream, raw, labels, H, W, C, \
                       bl, n_labels, start_label, g_cnt, g_bcnt, g_sum, g_sq, g_mn, g_mx)
#define LAUNCH_ZONAL(LPPV)                                                                                          \
    do {                                                                                                            \
        if (ident && bl.n == 4 * LPPV) LAUNCH_ZONAL_MODE(LPPV, 4, 0);                                               \
        else if (ident) LAUNCH_ZONAL_MODE(LPPV, 4, 1);                                                              \
        else LAUNCH_ZONAL_MODE(LPPV, 4, 2);                                                                         \
    } while (0)
    // 3 / 6 / 9 bands in order (an RGB raster; the author's nine bands): lanes own band TRIPLES -- one 12-byte load per lane and row,
    // no padded band slots, a quarter fewer lanes' worth of arithmetic per pixel than quads with a partial last one
    if (ident && bl.n == 3) LAUNCH_ZONAL_MODE(1, 3, 0);
    else if (ident && bl.n == 6) LAUNCH_ZONAL_MODE(2, 3, 0);
    else if (ident && bl.n == 9) LAUNCH_ZONAL_MODE(3, 3, 0);
    else if (bl.n <= 4) LAUNCH_ZONAL(1);
    else if (bl.n <= 8) LAUNCH_ZONAL(2);
    else if (bl.n <= 12) LAUNCH_ZONAL(3);
    else LAUNCH_ZONAL(4);
#undef LAUNCH_ZONAL
#undef LAUNCH_ZONAL_MODE
    hipLaunchKernelGGL(zonal_finalize_kernel, dim3(ib), dim3(256), 0, ctx->stream, g_cnt, g_bcnt, g_sum, g_sq, g_mn, g_mx,
                       (long long)n_labels, bl.n, count, mean, var, mn, mx);
    OBIA_HIP_TRY(hipGetLastError());
    return OBIA_OK;
}

}  // namespace obia
