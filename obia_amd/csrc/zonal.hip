// zonal.hip -- per-segment zonal statistics (count, mean, variance, min, max per band) on gfx950.
//
// Replaces the per-segment loop of obia create_objects (segment_statistics.py:475-491):
//   crop_image_to_bbox + mask_image_with_polygon (utils/utils.py:37-67)  -> "pixels of label p"
//   calculate_spectral_stats (segment_statistics.py:143-172)            -> np.mean / np.var / np.min / np.max
// One pass over (labels, raw raster): a workgroup owns a 32x32 tile, lanes own 1x4 vertical strips and
// merge runs of equal label in registers, partials go to a small LDS hash table keyed by label
// (64 slots; a 32x32 tile of S~18 superpixels touches ~9-16 labels), then one global atomic per
// (tile, label, band, statistic).  Sums are float64 (sum, sum of squares); variance = E[x^2] - E[x]^2
// in float64 meets the 1e-5 relative tolerance for uint16-range rasters.
#include "slic.hpp"

namespace obia {

constexpr int Z_TW = 32, Z_TH = 32, Z_NT = 256, Z_PPT = 4, Z_SLOTS = 64, Z_MAXB = 16;

struct BandList { int n; int b[Z_MAXB]; };

__device__ __forceinline__ unsigned zkey(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float zunkey(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

template <int NBP>   // NBP: band count rounded up (4, 8, 16) so the per-lane run state stays in registers
__global__ __launch_bounds__(Z_NT) void zonal_kernel(const float *__restrict__ raw, const int32_t *__restrict__ labels,
                                                     int H, int W, int C, BandList bl, int n_labels, int start_label,
                                                     unsigned *__restrict__ g_cnt, unsigned *__restrict__ g_bcnt,
                                                     double *__restrict__ g_sum, double *__restrict__ g_sq,
                                                     unsigned *__restrict__ g_mn, unsigned *__restrict__ g_mx) {
    __shared__ int s_key[Z_SLOTS];
    __shared__ unsigned s_cnt[Z_SLOTS];
    __shared__ unsigned s_bcnt[Z_SLOTS][NBP];
    __shared__ double s_sum[Z_SLOTS][NBP], s_sq[Z_SLOTS][NBP];
    __shared__ unsigned s_mn[Z_SLOTS][NBP], s_mx[Z_SLOTS][NBP];
    const int tid = threadIdx.x;
    const int nb = bl.n;
    for (int i = tid; i < Z_SLOTS; i += Z_NT) { s_key[i] = -1; s_cnt[i] = 0; }
    for (int i = tid; i < Z_SLOTS * NBP; i += Z_NT) {
        (&s_sum[0][0])[i] = 0.0; (&s_sq[0][0])[i] = 0.0; (&s_bcnt[0][0])[i] = 0u;
        (&s_mn[0][0])[i] = 0xffffffffu; (&s_mx[0][0])[i] = 0u;
    }
    __syncthreads();
    const int tiles_x = (W + Z_TW - 1) / Z_TW;
    const int ty0 = (blockIdx.x / tiles_x) * Z_TH, tx0 = (blockIdx.x % tiles_x) * Z_TW;
    const int x = tx0 + (tid & 31), yb = ty0 + (tid >> 5) * Z_PPT;

    // run state
    int rl = -1;
    unsigned rn = 0;
    unsigned rc[NBP];
    double rs[NBP], rq[NBP];
    float rmn[NBP], rmx[NBP];
    auto flush = [&]() {
        if (rl < 0) return;
        // find / insert the slot of label rl
        unsigned h = ((unsigned)rl * 2654435761u) >> 26;
        int slot = -1;
        for (int probe = 0; probe < Z_SLOTS; ++probe) {
            const int sidx = (h + probe) & (Z_SLOTS - 1);
            const int old = atomicCAS(&s_key[sidx], -1, rl);
            if (old == -1 || old == rl) { slot = sidx; break; }
        }
        if (slot >= 0) {
            atomicAdd(&s_cnt[slot], rn);
#pragma unroll
            for (int b = 0; b < NBP; ++b) {
                if (b >= nb || rc[b] == 0) continue;
                atomicAdd(&s_bcnt[slot][b], rc[b]);
                atomicAdd(&s_sum[slot][b], rs[b]);
                atomicAdd(&s_sq[slot][b], rq[b]);
                atomicMin(&s_mn[slot][b], zkey(rmn[b]));
                atomicMax(&s_mx[slot][b], zkey(rmx[b]));
            }
        } else {   // table full (more than 64 labels in one tile): straight to global memory
            atomicAdd(&g_cnt[rl], rn);
#pragma unroll
            for (int b = 0; b < NBP; ++b) {
                if (b >= nb || rc[b] == 0) continue;
                atomicAdd(&g_bcnt[(size_t)rl * nb + b], rc[b]);
                unsafeAtomicAdd(&g_sum[(size_t)rl * nb + b], rs[b]);
                unsafeAtomicAdd(&g_sq[(size_t)rl * nb + b], rq[b]);
                atomicMin(&g_mn[(size_t)rl * nb + b], zkey(rmn[b]));
                atomicMax(&g_mx[(size_t)rl * nb + b], zkey(rmx[b]));
            }
        }
    };
#pragma unroll
    for (int j = 0; j < Z_PPT; ++j) {
        const int y = yb + j;
        int l = -1;
        if (y < H && x < W) {
            l = labels[(long long)y * W + x] - start_label;
            if (l < 0 || l >= n_labels) l = -1;
        }
        if (l != rl) {
            flush();
            rl = l; rn = 0;
#pragma unroll
            for (int b = 0; b < NBP; ++b) { rc[b] = 0; rs[b] = 0.0; rq[b] = 0.0; rmn[b] = INFINITY; rmx[b] = -INFINITY; }
        }
        if (l >= 0) {
            const float *px = raw + ((long long)y * W + x) * C;
            rn += 1;
#pragma unroll
            for (int b = 0; b < NBP; ++b) {
                if (b >= nb) continue;
                const float v = px[bl.b[b]];
                if (v == v) {   // NaN pixels are dropped per band (`band[~isnan]`, segment_statistics.py:145-147)
                    const double dv = (double)v;
                    rc[b] += 1; rs[b] += dv; rq[b] += dv * dv;
                    rmn[b] = fminf(rmn[b], v); rmx[b] = fmaxf(rmx[b], v);
                }
            }
        }
    }
    flush();
    __syncthreads();
    for (int i = tid; i < Z_SLOTS * nb; i += Z_NT) {
        const int slot = i / nb, b = i - slot * nb;
        const int l = s_key[slot];
        if (l < 0) continue;
        if (b == 0) atomicAdd(&g_cnt[l], s_cnt[slot]);
        if (s_bcnt[slot][b] == 0) continue;
        atomicAdd(&g_bcnt[(size_t)l * nb + b], s_bcnt[slot][b]);
        unsafeAtomicAdd(&g_sum[(size_t)l * nb + b], s_sum[slot][b]);
        unsafeAtomicAdd(&g_sq[(size_t)l * nb + b], s_sq[slot][b]);
        if (s_mn[slot][b] != 0xffffffffu) atomicMin(&g_mn[(size_t)l * nb + b], s_mn[slot][b]);
        if (s_mx[slot][b] != 0u) atomicMax(&g_mx[(size_t)l * nb + b], s_mx[slot][b]);
    }
}

__global__ void zonal_init_kernel(unsigned *g_cnt, unsigned *g_bcnt, double *g_sum, double *g_sq, unsigned *g_mn, unsigned *g_mx,
                                  long long n_labels, int nb) {
    const long long n = n_labels * nb;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        g_sum[i] = 0.0; g_sq[i] = 0.0; g_bcnt[i] = 0u; g_mn[i] = 0xffffffffu; g_mx[i] = 0u;
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
        const unsigned c = g_bcnt[i];
        if (c == 0) {
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
    const int tiles = cdiv(W, Z_TW) * cdiv(H, Z_TH);
#define LAUNCH_ZONAL(NBPV)                                                                                          \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(zonal_kernel<NBPV>), dim3(tiles), dim3(Z_NT), 0, ctx->stream, raw, labels, H, W, C, \
                       bl, n_labels, start_label, g_cnt, g_bcnt, g_sum, g_sq, g_mn, g_mx)
    if (bl.n <= 4) LAUNCH_ZONAL(4);
    else if (bl.n <= 8) LAUNCH_ZONAL(8);
    else LAUNCH_ZONAL(16);
#undef LAUNCH_ZONAL
    hipLaunchKernelGGL(zonal_finalize_kernel, dim3(ib), dim3(256), 0, ctx->stream, g_cnt, g_bcnt, g_sum, g_sq, g_mn, g_mx,
                       (long long)n_labels, bl.n, count, mean, var, mn, mx);
    OBIA_HIP_TRY(hipGetLastError());
    return OBIA_OK;
}

}  // namespace obia
