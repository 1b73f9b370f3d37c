// quickshift.hip -- quickshift segmentation (BASELINE config 5) on gfx950.
//
// Restates skimage.segmentation.quickshift as obia calls it (obia/segmentation/segment_boundaries.py:48-49):
//   driver  _quickshift.py:59-74   img_as_float, rgb2lab (convert2lab), image *= ratio
//   kernel  _quickshift_cy.pyx     P1 density, P2 nearest pixel of higher density, P3 cut at max_dist + flatten
// (oracle/obia_oracle.c: obia_oracle_quickshift_core, pinned bit-exact on scikit-image 0.18.3 goldens).
// Arithmetic is float64, the dtype of the pinned scikit-image 0.18.3 kernel (it accepts nothing else; newer
// versions also take float32 -- see DESIGN.md).  Not HBM-bound: (2*ceil(3*ks)+1)^2 = 961 neighbour evaluations
// with one exp() each per pixel at ks = 5; the neighbourhood is staged in LDS as channel planes (lanes read
// consecutive doubles: conflict-free), a 16x16 pixel tile per workgroup, 46x46x3 doubles = 50.8 KB.
// Any band count (1..16) and any kernel_size are accepted, as the reference forwards **kwargs untouched: when the
// neighbourhood of a tile does not fit the LDS (more than 4 bands, or a window wider than the staged one) the same
// arithmetic runs on the channel planes in global memory (qs_density_parent_global_kernel).
#include "slic.hpp"

#include <cmath>

namespace obia {

constexpr int QT = 16;          // tile side
constexpr int QKW_MAX = 15;     // largest half-window staged in LDS (kernel_size <= 5)
constexpr int QC_MAX = 16;      // bands accepted (the feature pass packs at most 16)

__device__ __forceinline__ void rgb2lab_f64(double r, double g, double b, double &L, double &A, double &B) {
    double a[3] = {r, g, b};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double v = a[c];
        a[c] = (v > 0.04045) ? pow((v + 0.055) / 1.055, 2.4) : v / 12.92;
    }
    const double m[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
    const double wr[3] = {0.95047, 1.0, 1.08883};
    double xyz[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double s = a[0] * m[i][0];
        s = s + a[1] * m[i][1];
        s = s + a[2] * m[i][2];
        s = s / wr[i];
        xyz[i] = (s > 0.008856) ? cbrt(s) : 7.787 * s + 16.0 / 116.0;
    }
    L = 116.0 * xyz[1] - 16.0;
    A = 500.0 * (xyz[0] - xyz[1]);
    B = 200.0 * (xyz[1] - xyz[2]);
}

// float32 features (already normalised by features_kernel with ratio 1) -> float64 image: Lab (optional) * ratio,
// stored as channel planes [C][H][W]
__global__ __launch_bounds__(256) void qs_prepare_kernel(const float *__restrict__ feat, int CP, int C, long long npix,
                                                         int to_lab, double ratio, double *__restrict__ img) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        double v[16];
        for (int c = 0; c < C; ++c) v[c] = (double)feat[i * CP + c];
        if (to_lab) {
            double L, A, B;
            rgb2lab_f64(v[0], v[1], v[2], L, A, B);
            v[0] = L; v[1] = A; v[2] = B;
        }
        for (int c = 0; c < C; ++c) img[(long long)c * npix + i] = v[c] * ratio;
    }
}

// Gaussian pre-smoothing (`ndi.gaussian_filter(image, [sigma, sigma, 0])` on the float64 image): one axis pass of scipy's correlate1d
// (symmetric weights, mode 'reflect', see gauss_axis_kernel in slic.hip) over the channel planes [C][H][W]; `scale` multiplies the
// output (the last pass applies `* ratio`: the filter's own result is a double, so the product rounds as `image * ratio` does).
__global__ __launch_bounds__(256) void qs_gauss_axis_kernel(const double *__restrict__ in, double *__restrict__ out, int H, int W,
                                                            long long n_el, int axis_y, const double *__restrict__ w, int r, double scale) {
    const long long plane = (long long)H * W;
    const int n = axis_y ? H : W;
    const long long stride = axis_y ? W : 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (long long)gridDim.x * blockDim.x) {
        const long long in_plane = i % plane;
        const int pos = axis_y ? (int)(in_plane / W) : (int)(in_plane % W);
        const double *line = in + (i - (long long)pos * stride);
        double tmp = in[i] * w[0];
        for (int j = r; j >= 1; --j) {
            const int per = 2 * n;
            int lo = (pos - j) % per; if (lo < 0) lo += per; if (lo >= n) lo = per - 1 - lo;
            int hi = (pos + j) % per; if (hi >= n) hi = per - 1 - hi;
            tmp += (line[(long long)lo * stride] + line[(long long)hi * stride]) * w[j];
        }
        out[i] = tmp * scale;
    }
}

// P1 (density) and P2 (parent) share the staged neighbourhood.
template <int C>
__global__ __launch_bounds__(QT * QT) void qs_density_parent_kernel(const double *__restrict__ img, const double *__restrict__ noise,
                                                                   int H, int W, int kw, double inv, int phase,
                                                                   double *__restrict__ dens, int *__restrict__ parent,
                                                                   double *__restrict__ dist_parent) {
    extern __shared__ double s_tile[];   // [C (+1 in phase 2: density)][side][side]
    const int side = QT + 2 * kw;
    const int planes = C + (phase == 2 ? 1 : 0);
    const int ty0 = blockIdx.y * QT, tx0 = blockIdx.x * QT;
    const long long npix = (long long)H * W;
    for (int i = threadIdx.x; i < planes * side * side; i += QT * QT) {
        const int pl = i / (side * side), rem = i - pl * side * side;
        const int yy = ty0 - kw + rem / side, xx = tx0 - kw + rem % side;
        double v = 0.0;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            v = (pl < C) ? img[(long long)pl * npix + (long long)yy * W + xx] : dens[(long long)yy * W + xx];
        s_tile[i] = v;
    }
    __syncthreads();
    const int ly = threadIdx.x / QT, lx = threadIdx.x % QT;
    const int r = ty0 + ly, c = tx0 + lx;
    if (r >= H || c >= W) return;
    double cur[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) cur[ch] = s_tile[(ch * side + ly + kw) * side + lx + kw];
    const int r0 = max(r - kw, 0), r1 = min(r + kw + 1, H), c0 = max(c - kw, 0), c1 = min(c + kw + 1, W);
    if (phase == 1) {
        double acc = 0.0;
        for (int r_ = r0; r_ < r1; ++r_) {
            const double tr = (double)(r - r_);
            for (int c_ = c0; c_ < c1; ++c_) {
                double dist = 0.0;
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    const double t = cur[ch] - s_tile[(ch * side + (r_ - ty0 + kw)) * side + (c_ - tx0 + kw)];
                    dist += t * t;
                }
                dist += tr * tr;
                const double tc = (double)(c - c_);
                dist += tc * tc;
                acc += exp(dist * inv);
            }
        }
        // "this will break ties that otherwise would give us headache": densities += normal(scale=1e-5)
        dens[(long long)r * W + c] = acc + (noise ? noise[(long long)r * W + c] : 0.0);
    } else {
        const double cd = s_tile[(C * side + ly + kw) * side + lx + kw];
        double closest = INFINITY;
        int par = r * W + c;
        for (int r_ = r0; r_ < r1; ++r_) {
            const double tr = (double)(r - r_);
            for (int c_ = c0; c_ < c1; ++c_) {
                if (s_tile[(C * side + (r_ - ty0 + kw)) * side + (c_ - tx0 + kw)] > cd) {
                    double dist = 0.0;
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                        const double t = cur[ch] - s_tile[(ch * side + (r_ - ty0 + kw)) * side + (c_ - tx0 + kw)];
                        dist += t * t;
                    }
                    dist += tr * tr;
                    const double tc = (double)(c - c_);
                    dist += tc * tc;
                    if (dist < closest) { closest = dist; par = r_ * W + c_; }
                }
            }
        }
        parent[(long long)r * W + c] = par;
        dist_parent[(long long)r * W + c] = sqrt(closest);
    }
}

// The same two phases without LDS staging: any band count, any window.  Neighbours are read from the channel planes in
// global memory (a tile's lanes read 16 consecutive doubles of a row: coalesced, served by L1 / L2 after the first touch).
__global__ __launch_bounds__(QT * QT) void qs_density_parent_global_kernel(const double *__restrict__ img, const double *__restrict__ noise,
                                                                          int H, int W, int C, int kw, double inv, int phase,
                                                                          double *__restrict__ dens, int *__restrict__ parent,
                                                                          double *__restrict__ dist_parent) {
    const int ly = threadIdx.x / QT, lx = threadIdx.x % QT;
    const int r = blockIdx.y * QT + ly, c = blockIdx.x * QT + lx;
    if (r >= H || c >= W) return;
    const long long npix = (long long)H * W;
    const long long me = (long long)r * W + c;
    double cur[QC_MAX];
#pragma unroll
    for (int ch = 0; ch < QC_MAX; ++ch) cur[ch] = ch < C ? img[(long long)ch * npix + me] : 0.0;
    const int r0 = max(r - kw, 0), r1 = min(r + kw + 1, H), c0 = max(c - kw, 0), c1 = min(c + kw + 1, W);
    auto dist_to = [&](int r_, int c_) {
        const long long q = (long long)r_ * W + c_;
        double dist = 0.0;
#pragma unroll
        for (int ch = 0; ch < QC_MAX; ++ch)
            if (ch < C) { const double t = cur[ch] - img[(long long)ch * npix + q]; dist += t * t; }
        const double tr = (double)(r - r_), tc = (double)(c - c_);
        dist += tr * tr;
        dist += tc * tc;
        return dist;
    };
    if (phase == 1) {
        double acc = 0.0;
        for (int r_ = r0; r_ < r1; ++r_)
            for (int c_ = c0; c_ < c1; ++c_) acc += exp(dist_to(r_, c_) * inv);
        dens[me] = acc + (noise ? noise[me] : 0.0);
    } else {
        const double cd = dens[me];
        double closest = INFINITY;
        int par = (int)me;
        for (int r_ = r0; r_ < r1; ++r_)
            for (int c_ = c0; c_ < c1; ++c_)
                if (dens[(long long)r_ * W + c_] > cd) {
                    const double dist = dist_to(r_, c_);
                    if (dist < closest) { closest = dist; par = r_ * W + c_; }
                }
        parent[me] = par;
        dist_parent[me] = sqrt(closest);
    }
}

// P3: remove links longer than max_dist, then pointer jumping until every pixel points at its root
__global__ void qs_cut_kernel(int *__restrict__ parent, const double *__restrict__ dist_parent, long long n, double max_dist) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        if (dist_parent[i] > max_dist) parent[i] = (int)i;
}

__global__ void qs_jump_kernel(const int *__restrict__ pin, int *__restrict__ pout, long long n, int *__restrict__ changed) {
    bool ch = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int p = pin[i], pp = pin[p];
        pout[i] = pp;
        ch |= pp != p;
    }
    if (__ballot(ch) && (threadIdx.x & 63) == 0) atomicOr(changed, 1);
}

// labels = np.unique(roots, return_inverse=True)[1]: rank of each root among the roots in ascending pixel order
constexpr int QS_CHUNK = 4096;
__global__ __launch_bounds__(256) void qs_rootcount_kernel(const int *__restrict__ parent, long long n, int *__restrict__ block_sums) {
    __shared__ int s_w[4];
    const long long base = (long long)blockIdx.x * QS_CHUNK;
    int c = 0;
    for (int j = 0; j < QS_CHUNK / 256; ++j) {
        const long long i = base + (long long)j * 256 + threadIdx.x;
        c += (i < n && parent[i] == (int)i);
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(1024) void qs_scan_kernel(int *__restrict__ block_sums, int nb, int *__restrict__ total) {
    __shared__ int s_part[1024];
    const int tid = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, nb);
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
    if (tid == 1023) *total = s_part[1023];
}
__global__ __launch_bounds__(256) void qs_rootrank_kernel(const int *__restrict__ parent, long long n, const int *__restrict__ block_sums,
                                                          int *__restrict__ rank) {
    __shared__ int s_w[4];
    __shared__ int s_run;
    const long long base = (long long)blockIdx.x * QS_CHUNK;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_run = block_sums[blockIdx.x];
    __syncthreads();
    for (int j = 0; j < QS_CHUNK / 256; ++j) {
        const long long i = base + (long long)j * 256 + threadIdx.x;
        const bool flag = i < n && parent[i] == (int)i;
        const unsigned long long bal = __ballot(flag);
        if (lane == 0) s_w[wv] = __popcll(bal);
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wv; ++w) before += s_w[w];
        if (flag) rank[i] = s_run + before + __popcll(bal & ((1ull << lane) - 1ull));
        const int total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
        if (threadIdx.x == 0) s_run += total;
        __syncthreads();
    }
}
__global__ void qs_labels_kernel(const int *__restrict__ parent, const int *__restrict__ rank, long long n, int32_t *__restrict__ labels) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        labels[i] = rank[parent[i]];
}

static int quickshift_dev(obia_ctx *ctx, const float *img, int H, int W, int C, double ratio, double kernel_size, double max_dist,
                          double sigma, int convert2lab, const double *noise, int normalize_bands, int32_t *labels_out, int *n_labels_out) {
    if (!img || !labels_out || H <= 0 || W <= 0 || C <= 0) { set_error("bad arguments"); return OBIA_E_INVALID; }
    if ((long long)H * W > 0x7fffffffLL) { set_error("raster above 2^31 pixels"); return OBIA_E_INVALID; }
    if (!(kernel_size >= 1.0)) { set_error("`kernel_size` should be >= 1."); return OBIA_E_INVALID; }
    if (convert2lab && C != 3) { set_error("Only RGB images can be converted to Lab space."); return OBIA_E_INVALID; }
    if (C > QC_MAX) { set_error("quickshift: more than %d bands not supported (got %d)", QC_MAX, C); return OBIA_E_UNSUPPORTED; }
    if (!(kernel_size < 1.0e4)) { set_error("`kernel_size` too large"); return OBIA_E_INVALID; }
    if (!(sigma >= 0.0) || !(sigma < 1.0e6)) { set_error("sigma must be >= 0"); return OBIA_E_INVALID; }
    const int kw = (int)std::ceil(3.0 * kernel_size);
    // the LDS-staged kernel covers the reference's usual calls (1, 3 or 4 bands, kernel_size <= 5); everything else runs the
    // same arithmetic on global memory
    const bool staged = (C == 1 || C == 3 || C == 4) && kw <= QKW_MAX;
    Arena &A = ctx->arena;
    const long long n = (long long)H * W;
    // 1. obia's per-band normalisation (float32, segment_boundaries.py:32-33) through the SLIC feature kernels
    SlicBatch b;
    b.nprob = 1; b.C = C; b.CP = (C + 3) & ~3; b.total_pix = n;
    SlicProblem P{}; P.H = H; P.W = W;
    b.probs.assign(1, P);
    b.feat_planes = false;   // pixel-major features: qs_prepare_kernel reads them per pixel
    b.windows.assign(1, SrcWindow{0, 0, H, W, 0, 0, 0});
    b.d_windows = A.get<SrcWindow>(1);
    b.d_feat = A.get<float>((size_t)n * b.CP);
    double *d_img = A.get<double>((size_t)n * C);
    double *d_dens = A.get<double>((size_t)n), *d_dp = A.get<double>((size_t)n);
    int *d_par = A.get<int>((size_t)n), *d_par2 = A.get<int>((size_t)n), *d_rank = A.get<int>((size_t)n);
    const int nb = cdiv(n, QS_CHUNK);
    int *d_bs = A.get<int>(nb), *d_flags = A.get<int>(4);
    if (!b.d_windows || !b.d_feat || !d_img || !d_dens || !d_dp || !d_par || !d_par2 || !d_rank || !d_bs || !d_flags) return OBIA_E_NOMEM;
    OBIA_HIP_TRY(hipMemcpyAsync(b.d_windows, b.windows.data(), sizeof(SrcWindow), hipMemcpyHostToDevice, ctx->stream));
    OBIA_TRY(slic_prepare_features(ctx, b, img, H, W, normalize_bands, 0, 1.0f));
    int gs = cdiv(n, 256 * 4);
    if (gs > 65535) gs = 65535;
    const bool smoothing = sigma > 1e-15;
    hipLaunchKernelGGL(qs_prepare_kernel, dim3(gs), dim3(256), 0, ctx->stream, b.d_feat, b.CP, C, n, convert2lab ? 1 : 0, smoothing ? 1.0 : ratio, d_img);
    if (smoothing) {   // rows, then columns (gaussian_filter walks the axes in order; the band axis has sigma 0), then `* ratio`
        std::vector<double> w;
        const int r = gaussian_weights_host(sigma, false, w);
        double *d_w = A.get<double>(w.size()), *d_tmp = A.get<double>((size_t)n * C);
        if (!d_w || !d_tmp) return OBIA_E_NOMEM;
        OBIA_TRY(upload_async(ctx, d_w, w.data(), sizeof(double) * w.size()));
        int gg = cdiv(n * C, 256);
        if (gg > 65535) gg = 65535;
        hipLaunchKernelGGL(qs_gauss_axis_kernel, dim3(gg), dim3(256), 0, ctx->stream, d_img, d_tmp, H, W, n * C, 1, d_w, r, 1.0);
        hipLaunchKernelGGL(qs_gauss_axis_kernel, dim3(gg), dim3(256), 0, ctx->stream, d_tmp, d_img, H, W, n * C, 0, d_w, r, ratio);
    }
    // 2. density, 3. parent
    const double inv = -0.5 / (kernel_size * kernel_size);
    const int side = QT + 2 * kw;
    dim3 grid(cdiv(W, QT), cdiv(H, QT));
#define QS_LAUNCH(CV, PH) do {                                                                                         \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qs_density_parent_kernel<CV>),                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize,                                             \
                              (int)(sizeof(double) * (size_t)(CV + 1) * side * side));                                \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(qs_density_parent_kernel<CV>), grid, dim3(QT * QT),                            \
                       sizeof(double) * (size_t)(CV + (PH == 2 ? 1 : 0)) * side * side, ctx->stream, d_img, noise, H, W, kw, inv, \
                       PH, d_dens, d_par, d_dp); } while (0)
    for (int ph = 1; ph <= 2; ++ph) {
        if (!staged)
            hipLaunchKernelGGL(qs_density_parent_global_kernel, grid, dim3(QT * QT), 0, ctx->stream, d_img, noise, H, W, C, kw, inv, ph,
                               d_dens, d_par, d_dp);
        else if (C == 1) { if (ph == 1) QS_LAUNCH(1, 1); else QS_LAUNCH(1, 2); }
        else if (C == 3) { if (ph == 1) QS_LAUNCH(3, 1); else QS_LAUNCH(3, 2); }
        else { if (ph == 1) QS_LAUNCH(4, 1); else QS_LAUNCH(4, 2); }
    }
#undef QS_LAUNCH
    OBIA_HIP_TRY(hipGetLastError());
    // 4. cut, flatten
    hipLaunchKernelGGL(qs_cut_kernel, dim3(gs), dim3(256), 0, ctx->stream, d_par, d_dp, n, max_dist);
    int *pa = d_par, *pb = d_par2;
    for (int it = 0; it < 64; ++it) {   // pointer jumping halves every chain: 31 rounds reach the root of any chain below 2^31
        OBIA_HIP_TRY(hipMemsetAsync(d_flags, 0, sizeof(int), ctx->stream));
        hipLaunchKernelGGL(qs_jump_kernel, dim3(gs), dim3(256), 0, ctx->stream, pa, pb, n, d_flags);
        std::swap(pa, pb);
        int changed = 0;
        OBIA_TRY(read_back(ctx, &changed, d_flags, sizeof(int)));
        if (!changed) break;
    }
    // 5. consecutive labels by ascending root index
    hipLaunchKernelGGL(qs_rootcount_kernel, dim3(nb), dim3(256), 0, ctx->stream, pa, n, d_bs);
    hipLaunchKernelGGL(qs_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_bs, nb, d_flags + 1);
    hipLaunchKernelGGL(qs_rootrank_kernel, dim3(nb), dim3(256), 0, ctx->stream, pa, n, d_bs, d_rank);
    hipLaunchKernelGGL(qs_labels_kernel, dim3(gs), dim3(256), 0, ctx->stream, pa, d_rank, n, labels_out);
    OBIA_HIP_TRY(hipGetLastError());
    int total = 0;
    OBIA_TRY(read_back(ctx, &total, d_flags + 1, sizeof(int)));
    if (n_labels_out) *n_labels_out = total;
    return OBIA_OK;
}

}  // namespace obia

using namespace obia;

extern "C" {

int obia_quickshift_f32_dev(obia_ctx *ctx, const float *img, int H, int W, int C, double ratio, double kernel_size,
                            double max_dist, double sigma, int convert2lab, const double *tie_noise_hw, int normalize_bands,
                            int32_t *labels_out, int *n_labels_out) {
    if (!ctx) { set_error("null context"); return OBIA_E_INVALID; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return OBIA_E_HIP; }
    ctx->arena.reset();
    begin_timing(ctx);
    int rc;
    {
        ScopedSpan total(ctx, T_TOTAL);
        rc = quickshift_dev(ctx, img, H, W, C, ratio, kernel_size, max_dist, sigma, convert2lab, tie_noise_hw, normalize_bands,
                            labels_out, n_labels_out);
    }
    if (rc != OBIA_OK) return rc;
    OBIA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    resolve_timing(ctx);
    return OBIA_OK;
}

int obia_quickshift_f32(obia_ctx *ctx, const float *img, int H, int W, int C, double ratio, double kernel_size,
                        double max_dist, double sigma, int convert2lab, const double *tie_noise_hw, int normalize_bands,
                        int32_t *labels_out, int *n_labels_out) {
    if (!ctx) { set_error("null context"); return OBIA_E_INVALID; }
    if (!img || !labels_out || H <= 0 || W <= 0 || C <= 0) { set_error("bad arguments"); return OBIA_E_INVALID; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return OBIA_E_HIP; }
    const size_t npix = (size_t)H * W;
    float *d_img = nullptr; double *d_noise = nullptr; int32_t *d_lab = nullptr;
    int rc = OBIA_OK;
    if (hipMalloc(&d_img, npix * C * sizeof(float)) != hipSuccess || hipMalloc(&d_lab, npix * sizeof(int32_t)) != hipSuccess ||
        (tie_noise_hw && hipMalloc(&d_noise, npix * sizeof(double)) != hipSuccess)) {
        set_error("device allocation for host-pointer call failed");
        rc = OBIA_E_NOMEM;
    }
    if (rc == OBIA_OK && hipMemcpyAsync(d_img, img, npix * C * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = OBIA_E_HIP;
    if (rc == OBIA_OK && tie_noise_hw && hipMemcpyAsync(d_noise, tie_noise_hw, npix * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = OBIA_E_HIP;
    if (rc == OBIA_E_HIP) set_error("host->device copy failed in obia_quickshift_f32");
    if (rc == OBIA_OK) rc = obia_quickshift_f32_dev(ctx, d_img, H, W, C, ratio, kernel_size, max_dist, sigma, convert2lab, d_noise, normalize_bands, d_lab, n_labels_out);
    if (rc == OBIA_OK && hipMemcpyAsync(labels_out, d_lab, npix * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
        set_error("device->host copy failed in obia_quickshift_f32");
        rc = OBIA_E_HIP;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_img); (void)hipFree(d_lab); (void)hipFree(d_noise);
    return rc;
}

}  // extern "C"
