// texture.hip -- per-segment GLCM texture statistics on gfx950.
//
// Restates calculate_textural_stats (obia/segmentation/segment_statistics.py:179-298) on the masked bounding-box crop
// create_objects builds per segment (:478-479; utils/utils.py:37-67), with the band PLANE that the code evidently means
// (the reference writes image[:, :, band] on a (bands, h, w) crop, :214 -- see oracle/glcm.py):
//   crop of the band to the segment's bounding box, 0 outside the segment and at NaN pixels; min / max over that crop
//   (zeros included); uint8((v - min) / (max - min) * 255) in float32; grey-level co-occurrence matrix at distance 2,
//   angles 0, pi/4, pi/2, 3pi/4 = offsets (0,2), (1,1), (2,0), (1,-1), 256 levels, symmetric, normed; contrast,
//   dissimilarity, homogeneity, ASM, energy, correlation per angle (scikit-image greycoprops) and their mean over the
//   four angles.
// The matrix is never materialised for the common case.  With P symmetric and normed, over the n pixel pairs (a, b):
//   contrast = mean((a-b)^2), dissimilarity = mean|a-b|, homogeneity = mean(1/(1+(a-b)^2)),
//   mu = sum(a+b)/(2n), var = sum(a^2+b^2)/(2n) - mu^2, correlation = (mean(ab) - mu^2)/var  (1 when var == 0),
//   ASM = (sum over unordered value pairs {lo,hi} with count c of (lo == hi ? 4 : 2) * c^2) / (2n)^2
// -- integer sums (exact) except the homogeneity terms.  Only ASM needs counts per matrix cell: an LDS hash table keyed
// by the unordered value pair.  One workgroup per segment; crops of up to 4096 pixels (every tiled-SLIC segment) keep
// the quantised crop and the table in LDS; larger crops use a dense 256x256 counter matrix in global scratch.
#include "slic.hpp"

namespace obia {

constexpr int TX_NT = 256, TX_MAXPIX = 4096, TX_TABLE = 8192, TX_MAXB = 16;

struct TexBands { int n; int b[TX_MAXB]; };

// ---- bounding boxes ---------------------------------------------------------------------------------------------------
constexpr int BB_SLOTS = 128;
__global__ __launch_bounds__(64) void bbox_kernel(const int32_t *__restrict__ lab, int H, int W, int n_labels, int start_label,
                                                  int *__restrict__ bbox /*[n_labels][4]: y0, y1, x0, x1 (inclusive)*/) {
    __shared__ int s_key[BB_SLOTS];
    __shared__ int s_box[BB_SLOTS][4];
    const int bw = (W + 63) / 64;
    const int by = blockIdx.x / bw, bx = blockIdx.x % bw;
    const int lane = threadIdx.x;
    for (int i = lane; i < BB_SLOTS; i += 64) { s_key[i] = -1; s_box[i][0] = INT32_MAX; s_box[i][1] = -1; s_box[i][2] = INT32_MAX; s_box[i][3] = -1; }
    __syncthreads();
    const int x = bx * 64 + lane;
    const int y_lo = by * 64, y_hi = min(y_lo + 64, H);
    int rl = -1, ry0 = 0;
    auto close_run = [&](int y_last) {
        if (rl < 0) return;
        const unsigned h = ((unsigned)rl * 2654435761u) >> 25;
        int slot = -1;
#pragma unroll 1
        for (int probe = 0; probe < BB_SLOTS; ++probe) {
            const int sidx = (h + probe) & (BB_SLOTS - 1);
            const int old = atomicCAS(&s_key[sidx], -1, rl);
            if (old == -1 || old == rl) { slot = sidx; break; }
        }
        int *box = slot >= 0 ? s_box[slot] : bbox + 4 * (size_t)rl;
        atomicMin(&box[0], ry0); atomicMax(&box[1], y_last); atomicMin(&box[2], x); atomicMax(&box[3], x);
    };
#pragma unroll 1
    for (int y0 = y_lo; y0 < y_hi; y0 += 8) {
        int l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int y = y0 + j;
            int v = -1;
            if (x < W && y < y_hi) { v = lab[(long long)y * W + x] - start_label; if (v < 0 || v >= n_labels) v = -1; }
            l[j] = v;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (l[j] != rl) { close_run(y0 + j - 1); rl = l[j]; ry0 = y0 + j; }
        }
    }
    close_run(y_hi - 1);
    __syncthreads();
    for (int i = lane; i < BB_SLOTS; i += 64) {
        const int l = s_key[i];
        if (l < 0) continue;
        int *box = bbox + 4 * (size_t)l;
        atomicMin(&box[0], s_box[i][0]); atomicMax(&box[1], s_box[i][1]); atomicMin(&box[2], s_box[i][2]); atomicMax(&box[3], s_box[i][3]);
    }
}

__global__ void bbox_init_kernel(int *bbox, long long n_labels) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_labels) { bbox[4 * i] = INT32_MAX; bbox[4 * i + 1] = -1; bbox[4 * i + 2] = INT32_MAX; bbox[4 * i + 3] = -1; }
}

// ---- the statistics of one angle from its sums -------------------------------------------------------------------------
struct PairSums {
    unsigned long long n, s1, s2, sab, sd2, sd1, asm2;   // pairs, sum(a+b), sum(a^2+b^2), sum(ab), sum((a-b)^2), sum|a-b|, ASM numerator
    double sh;                                           // sum 1/(1+(a-b)^2)
};

__device__ __forceinline__ void add_angle(const PairSums &s, double out[6]) {
    if (s.n == 0) { out[5] += 1.0; return; }             // empty matrix: every sum is 0, both deviations are 0 -> correlation 1
    const double n = (double)s.n, n2 = 2.0 * n;
    out[0] += (double)s.sd2 / n;
    out[1] += (double)s.sd1 / n;
    out[2] += s.sh / n;
    const double asmv = (double)s.asm2 / (n2 * n2);
    out[3] += asmv;
    out[4] += sqrt(asmv);
    // var = (2n*s2 - s1^2) / (2n)^2 and cov = (4n*sab - s1^2) / (2n)^2 in exact integers (values <= 255)
    const unsigned __int128 a = (unsigned __int128)(2 * s.n) * s.s2, b = (unsigned __int128)s.s1 * s.s1;
    const unsigned __int128 c = (unsigned __int128)(4 * s.n) * s.sab;
    const double var = (double)(a - b) / (n2 * n2);
    const double cov = (c >= b ? (double)(c - b) : -(double)(b - c)) / (n2 * n2);
    out[5] += (sqrt(var) < 1e-15) ? 1.0 : cov / var;
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T *s_red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    T t = s_red[0];
    for (int w = 1; w < TX_NT / 64; ++w) t += s_red[w];
    return t;
}

// One workgroup per segment.  BIG = false: crops of at most TX_MAXPIX pixels, everything in LDS.  BIG = true: any crop,
// quantised values recomputed from global memory, counts in a dense 256x256 matrix of the workgroup's global scratch.
template <bool BIG>
__global__ __launch_bounds__(TX_NT) void texture_kernel(const float *__restrict__ raw, const int32_t *__restrict__ lab, int H, int W,
                                                        int C, TexBands tb, int n_labels, int start_label,
                                                        const int *__restrict__ bbox, const int *__restrict__ big_list, int n_big,
                                                        unsigned *__restrict__ scratch /*[gridDim.x][65536], BIG only*/,
                                                        double *__restrict__ out /*[6][n_labels][n_bands]*/) {
    __shared__ uint8_t s_q[BIG ? 4 : TX_MAXPIX];
    __shared__ unsigned s_tab[BIG ? 4 : TX_TABLE];
    __shared__ float s_f[TX_NT / 64][2];
    __shared__ unsigned long long s_u[TX_NT / 64];
    __shared__ double s_d[TX_NT / 64];
    __shared__ int s_any;
    const int tid = threadIdx.x;
    const int OFF[4][2] = {{0, 2}, {1, 1}, {2, 0}, {1, -1}};
    for (int item = blockIdx.x; item < (BIG ? n_big : n_labels); item += gridDim.x) {
        const int L = BIG ? big_list[item] : item;
        const int y0 = bbox[4 * (size_t)L], y1 = bbox[4 * (size_t)L + 1], x0 = bbox[4 * (size_t)L + 2], x1 = bbox[4 * (size_t)L + 3];
        if (y1 < y0) continue;                               // empty label: outputs stay NaN
        const int h = y1 - y0 + 1, w = x1 - x0 + 1;
        const long long npx = (long long)h * w;
        if (!BIG && npx > TX_MAXPIX) continue;               // handled by the BIG launch
        for (int bi = 0; bi < tb.n; ++bi) {
            const int band = tb.b[bi];
            auto clean_at = [&](int r, int c, bool &valid) -> float {
                const long long pix = (long long)(y0 + r) * W + x0 + c;
                const float v = raw[pix * C + band];
                valid = (lab[pix] - start_label == L) && (v == v);
                return valid ? v : 0.0f;
            };
            // min / max of the zero-filled crop, and whether any pixel is valid
            float lo = INFINITY, hi = -INFINITY;
            int anyv = 0;
            for (long long i = tid; i < npx; i += TX_NT) {
                bool valid;
                const float v = clean_at((int)(i / w), (int)(i % w), valid);
                anyv |= valid;
                lo = fminf(lo, v); hi = fmaxf(hi, v);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off)); hi = fmaxf(hi, __shfl_xor(hi, off)); anyv |= __shfl_xor(anyv, off); }
            __syncthreads();
            if (tid == 0) s_any = 0;
            __syncthreads();
            if ((tid & 63) == 0) { s_f[tid >> 6][0] = lo; s_f[tid >> 6][1] = hi; if (anyv) atomicOr(&s_any, 1); }
            __syncthreads();
            lo = fminf(fminf(s_f[0][0], s_f[1][0]), fminf(s_f[2][0], s_f[3][0]));
            hi = fmaxf(fmaxf(s_f[0][1], s_f[1][1]), fmaxf(s_f[2][1], s_f[3][1]));
            if (!s_any) continue;                            // no valid pixel in this band: NaN (whole workgroup)
            const float den = hi - lo;
            auto quant = [&](int r, int c) -> int {
                if (hi == lo) return 0;
                bool valid;
                const float v = clean_at(r, c, valid);
                return (int)(uint8_t)(((v - lo) / den) * 255.0f);
            };
            if (!BIG) {
                for (int i = tid; i < (int)npx; i += TX_NT) s_q[i] = (uint8_t)quant(i / w, i % w);
                __syncthreads();
            }
            double res[6] = {0, 0, 0, 0, 0, 0};
            for (int ang = 0; ang < 4; ++ang) {
                const int dr = OFF[ang][0], dc = OFF[ang][1];
                const int c_lo = dc < 0 ? -dc : 0, c_hi = dc > 0 ? w - dc : w, r_hi = h - dr;
                const int pw = c_hi - c_lo;
                const long long npairs = (r_hi > 0 && pw > 0) ? (long long)r_hi * pw : 0;
                unsigned *tab = BIG ? scratch + (size_t)blockIdx.x * 65536 : s_tab;
                // the table only has to hold the distinct value pairs (<= pairs <= crop pixels): size it to the crop, so
                // that clearing it does not dominate the small segments of a tiled SLIC
                int tbits = 16;
                if (!BIG) { tbits = 8; while ((1 << tbits) < 2 * (int)npx && tbits < 13) ++tbits; }
                const int tabn = 1 << tbits;
                for (int i = tid; i < tabn; i += TX_NT) tab[i] = 0u;
                __syncthreads();
                PairSums ps{0, 0, 0, 0, 0, 0, 0, 0.0};
                for (long long i = tid; i < npairs; i += TX_NT) {
                    const int r = (int)(i / pw), c = c_lo + (int)(i % pw);
                    const int a = BIG ? quant(r, c) : s_q[r * w + c];
                    const int b = BIG ? quant(r + dr, c + dc) : s_q[(r + dr) * w + c + dc];
                    const int d = a > b ? a - b : b - a;
                    ps.n += 1; ps.s1 += a + b; ps.s2 += a * a + b * b; ps.sab += a * b; ps.sd2 += d * d; ps.sd1 += d;
                    ps.sh += 1.0 / (1.0 + (double)(d * d));
                    const unsigned lo8 = a < b ? a : b, hi8 = a < b ? b : a;
                    if (BIG) atomicAdd(&tab[lo8 * 256 + hi8], 1u);
                    else {
                        const unsigned key = ((lo8 << 8) | hi8) + 1u;                 // 1 .. 65536, 0 = empty entry
                        unsigned idx = (key * 2654435761u) >> (32 - tbits);
                        for (;;) {
                            const unsigned e = tab[idx];
                            if (e == 0u) {
                                const unsigned old = atomicCAS(&tab[idx], 0u, (key << 15) | 1u);   // key in bits 15.., count below
                                if (old == 0u) break;
                                if ((old >> 15) == key) { atomicAdd(&tab[idx], 1u); break; }
                            } else if ((e >> 15) == key) { atomicAdd(&tab[idx], 1u); break; }
                            idx = (idx + 1) & (tabn - 1);
                        }
                    }
                }
                __syncthreads();
                // ASM numerator from the cell counts
                unsigned long long a2 = 0;
                for (int i = tid; i < tabn; i += TX_NT) {
                    const unsigned e = tab[i];
                    if (!e) continue;
                    unsigned long long cnt;
                    bool diag;
                    if (BIG) { cnt = e; diag = (i >> 8) == (i & 255); }
                    else { cnt = e & 0x7fffu; const unsigned key = (e >> 15) - 1u; diag = (key >> 8) == (key & 255u); }
                    a2 += (diag ? 4ull : 2ull) * cnt * cnt;
                }
                PairSums tot;
                tot.n = block_sum(ps.n, s_u); tot.s1 = block_sum(ps.s1, s_u); tot.s2 = block_sum(ps.s2, s_u);
                tot.sab = block_sum(ps.sab, s_u); tot.sd2 = block_sum(ps.sd2, s_u); tot.sd1 = block_sum(ps.sd1, s_u);
                tot.asm2 = block_sum(a2, s_u);
                tot.sh = block_sum(ps.sh, s_d);
                add_angle(tot, res);
                __syncthreads();
            }
            if (tid < 6) out[((size_t)tid * n_labels + L) * tb.n + bi] = res[tid] * 0.25;
        }
    }
}

__global__ void tex_list_big_kernel(const int *__restrict__ bbox, int n_labels, int *__restrict__ big_list, int *__restrict__ n_big) {
    const int L = blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= n_labels) return;
    const int y0 = bbox[4 * L], y1 = bbox[4 * L + 1], x0 = bbox[4 * L + 2], x1 = bbox[4 * L + 3];
    if (y1 < y0) return;
    if ((long long)(y1 - y0 + 1) * (x1 - x0 + 1) > TX_MAXPIX) big_list[atomicAdd(n_big, 1)] = L;
}

__global__ void fill_nan_kernel(double *p, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = NAN;
}

int texture_stats_dev(obia_ctx *ctx, const float *raw, const int32_t *labels, int H, int W, int C, const int32_t *bands_host,
                      int n_bands, int n_labels, int start_label, double *out6) {
    if (H <= 0 || W <= 0 || C <= 0 || n_labels < 0) { set_error("bad texture_stats shape"); return OBIA_E_INVALID; }
    TexBands tb;
    if (bands_host == nullptr) {
        if (C > TX_MAXB) { set_error("more than %d bands not supported", TX_MAXB); return OBIA_E_UNSUPPORTED; }
        tb.n = C;
        for (int i = 0; i < C; ++i) tb.b[i] = i;
    } else {
        if (n_bands < 1 || n_bands > TX_MAXB) { set_error("n_bands %d out of range (1..%d)", n_bands, TX_MAXB); return OBIA_E_UNSUPPORTED; }
        tb.n = n_bands;
        for (int i = 0; i < n_bands; ++i) {
            if (bands_host[i] < 0 || bands_host[i] >= C) { set_error("band index %d out of range (0..%d)", bands_host[i], C - 1); return OBIA_E_INVALID; }
            tb.b[i] = bands_host[i];
        }
    }
    for (int i = tb.n; i < TX_MAXB; ++i) tb.b[i] = 0;
    if (n_labels == 0) return OBIA_OK;
    Arena &A = ctx->arena;
    int *bbox = A.get<int>(4 * (size_t)n_labels);
    int *big_list = A.get<int>((size_t)n_labels);
    int *d_nbig = A.get<int>(1);
    if (!bbox || !big_list || !d_nbig) return OBIA_E_NOMEM;
    const long long nout = 6LL * n_labels * tb.n;
    hipLaunchKernelGGL(fill_nan_kernel, dim3(cdiv(nout, 256)), dim3(256), 0, ctx->stream, out6, nout);
    hipLaunchKernelGGL(bbox_init_kernel, dim3(cdiv(n_labels, 256)), dim3(256), 0, ctx->stream, bbox, (long long)n_labels);
    OBIA_HIP_TRY(hipMemsetAsync(d_nbig, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(bbox_kernel, dim3(cdiv(W, 64) * cdiv(H, 64)), dim3(64), 0, ctx->stream, labels, H, W, n_labels, start_label, bbox);
    hipLaunchKernelGGL(tex_list_big_kernel, dim3(cdiv(n_labels, 256)), dim3(256), 0, ctx->stream, bbox, n_labels, big_list, d_nbig);
    int grid = n_labels < 65536 ? n_labels : 65536;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(texture_kernel<false>), dim3(grid), dim3(TX_NT), 0, ctx->stream, raw, labels, H, W, C, tb,
                       n_labels, start_label, bbox, (const int *)nullptr, 0, (unsigned *)nullptr, out6);
    int n_big = 0;
    OBIA_TRY(read_back(ctx, &n_big, d_nbig, sizeof(int)));
    if (n_big > 0) {
        const int gb = n_big < 256 ? n_big : 256;
        unsigned *scratch = A.get<unsigned>((size_t)gb * 65536);
        if (!scratch) return OBIA_E_NOMEM;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(texture_kernel<true>), dim3(gb), dim3(TX_NT), 0, ctx->stream, raw, labels, H, W, C, tb,
                           n_labels, start_label, bbox, big_list, n_big, scratch, out6);
    }
    OBIA_HIP_TRY(hipGetLastError());
    return OBIA_OK;
}

}  // namespace obia

using namespace obia;

extern "C" {

int obia_texture_stats_f32_dev(obia_ctx *ctx, const float *raw, const int32_t *labels, int H, int W, int C, const int32_t *bands,
                               int n_bands, int n_labels, int start_label, double *out6) {
    if (!ctx) { set_error("null context"); return OBIA_E_INVALID; }
    if (!raw || !labels || !out6) { set_error("null pointer argument"); return OBIA_E_INVALID; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return OBIA_E_HIP; }
    ctx->arena.reset();
    OBIA_TRY(texture_stats_dev(ctx, raw, labels, H, W, C, bands, n_bands, n_labels, start_label, out6));
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return OBIA_OK;
}

}  // extern "C"
