// Microbenchmark: achievable HBM read / write / copy bandwidth on gfx950 with float4 accesses (grid-stride, unroll U).
// hipcc --offload-arch=gfx950 -O3 -o hbm_rate hbm_rate.hip && ./hbm_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int U>
__global__ __launch_bounds__(256) void rd(const float4 *__restrict__ a, float *out, size_t n) {
    float s = 0;
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (; i + 256 * (U - 1) < n; i += stride) {
        float4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = a[i + 256 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) s += t[u].x + t[u].y + t[u].z + t[u].w;
    }
    if (s == 12345.678f) out[0] = s;
}
template <int U>
__global__ __launch_bounds__(256) void wr(float4 *__restrict__ a, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (; i + 256 * (U - 1) < n; i += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) a[i + 256 * u] = make_float4(1.f, 2.f, 3.f, (float)i);
    }
}
template <int U>
__global__ __launch_bounds__(256) void cp(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (; i + 256 * (U - 1) < n; i += stride) {
        float4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = a[i + 256 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) b[i + 256 * u] = t[u];
    }
}
// windowed copy: `nw` windows of wh x ww pixels (32 B each) out of a raster Ws pixels wide -> dense destination.
// MODE 0: one workgroup per window row (the row piece is contiguous, rows are Ws*32 B apart)
// MODE 1: one workgroup per 8-row x 256-pixel block of the window
template <int MODE>
__global__ __launch_bounds__(256) void wincopy(const float4 *__restrict__ src, float4 *__restrict__ dst, int Ws, int ww, int wh) {
    const int w = blockIdx.y;
    const size_t sbase = ((size_t)(w / 4) * 2048 * Ws + (size_t)(w % 4) * 2 * 2048) * 2;   // window origin, in float4
    const size_t dbase = (size_t)w * ww * wh * 2;
    if (MODE == 0) {
        for (int y = blockIdx.x; y < wh; y += gridDim.x) {
            const float4 *r = src + sbase + (size_t)y * Ws * 2;
            float4 *d = dst + dbase + (size_t)y * ww * 2;
            for (int base = threadIdx.x; base < ww * 2; base += 1024) {
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) if (base + 256 * u < ww * 2) t[u] = r[base + 256 * u];
#pragma unroll
                for (int u = 0; u < 4; ++u) if (base + 256 * u < ww * 2) d[base + 256 * u] = t[u];
            }
        }
    } else {
        const int bx = blockIdx.x % ((ww + 255) / 256), by = blockIdx.x / ((ww + 255) / 256);
        float4 t[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int y = by * 8 + j;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = bx * 512 + 256 * u + threadIdx.x;
                if (y < wh && c < ww * 2) t[2 * j + u] = src[sbase + (size_t)y * Ws * 2 + c];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int y = by * 8 + j;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = bx * 512 + 256 * u + threadIdx.x;
                if (y < wh && c < ww * 2) dst[dbase + (size_t)y * ww * 2 + c] = t[2 * j + u];
            }
        }
    }
}
// replica of the feature-preparation pass (one pixel of 8 bands per thread and iteration) with its parts switchable
struct Win { int y0, x0, h, w; long long pix_off; };
template <int DIV, int KEYS, int ATOM, int UNR>
__global__ __launch_bounds__(256) void featlike(const float *__restrict__ src, int Ws, const Win *__restrict__ wins,
                                                const unsigned *__restrict__ keys, float ratio, float *__restrict__ feat,
                                                unsigned *__restrict__ maxabs) {
    const int p = blockIdx.y;
    const Win wdw = wins[p];
    float bmn[8], bden[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        bmn[c] = 1.0f; bden[c] = 3.0f;
        if (KEYS) { bmn[c] = __uint_as_float(keys[(p * 8 + c) * 2]); bden[c] = __uint_as_float(keys[(p * 8 + c) * 2 + 1]); }
    }
    float local_max = 0.0f;
    for (int y = blockIdx.x; y < wdw.h; y += gridDim.x)
    for (int x0 = threadIdx.x; x0 < wdw.w; x0 += 256 * UNR) {
        float4 t[UNR][2];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int x = x0 + 256 * u;
            if (x < wdw.w) {
                const float4 *px = reinterpret_cast<const float4 *>(src + ((long long)(wdw.y0 + y) * Ws + wdw.x0 + x) * 8);
                t[u][0] = px[0]; t[u][1] = px[1];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int x = x0 + 256 * u;
            if (x >= wdw.w) continue;
            float v[8] = {t[u][0].x, t[u][0].y, t[u][0].z, t[u][0].w, t[u][1].x, t[u][1].y, t[u][1].z, t[u][1].w};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float r = DIV ? (v[c] - bmn[c]) / bden[c] : (v[c] - bmn[c]) * bden[c];
                if (!(fabsf(r) <= 3.0e38f)) r = 0.0f;
                v[c] = r * ratio;
                local_max = fmaxf(local_max, fabsf(v[c]));
            }
            float4 *dst = reinterpret_cast<float4 *>(feat + (wdw.pix_off + (long long)y * wdw.w + x) * 8);
            dst[0] = make_float4(v[0], v[1], v[2], v[3]); dst[1] = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
    if (ATOM) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) local_max = fmaxf(local_max, __shfl_xor(local_max, off));
        if ((threadIdx.x & 63) == 0) atomicMax(maxabs, __float_as_uint(local_max));
    } else if (local_max == 12345.0f) maxabs[1] = 1;
}
template <class F>
static void timeit(const char *name, int blocks, double traffic, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s blocks %7d: %7.1f GB/s\n", name, blocks, traffic * 3 / (ms * 1e6));
}
int main() {
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    float4 *a, *b; float *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 1, bytes);
    for (int blocks : {2048, 8192, 65536, 524288}) {
        timeit("read U4", blocks, (double)bytes, [&] { rd<4><<<blocks, 256>>>(a, o, n); });
        timeit("read U8", blocks, (double)bytes, [&] { rd<8><<<blocks, 256>>>(a, o, n); });
        timeit("write U4", blocks, (double)bytes, [&] { wr<4><<<blocks, 256>>>(a, n); });
        timeit("copy U4", blocks, 2.0 * bytes, [&] { cp<4><<<blocks, 256>>>(a, b, n); });
        timeit("copy U8", blocks, 2.0 * bytes, [&] { cp<8><<<blocks, 256>>>(a, b, n); });
    }
    {
        const int Ws = 16384, ww = 2176, wh = 2176;     // needs (3*2048+2176) rows of the raster: 8.6 GB raster -> use rows < 8192+...
        float4 *big; hipMalloc(&big, (size_t)16384 * 16384 * 32); hipMemset(big, 1, (size_t)16384 * 16384 * 32);
        {
            Win hw[4]; for (int i = 0; i < 4; ++i) hw[i] = Win{1984, 1984 + 4096 * i, 2176, 2176, (long long)i * 2176 * 2176};
            Win *dw; hipMalloc(&dw, sizeof(hw)); hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice);
            unsigned *dk; hipMalloc(&dk, 4096); hipMemset(dk, 0x3f, 4096);
            const double traffic = 2.0 * 4 * 2176.0 * 2176.0 * 32.0;
            const float *sf = reinterpret_cast<const float *>(big); float *df = reinterpret_cast<float *>(b);
            timeit("feat full", 4, traffic, [&] { featlike<1, 1, 1, 1><<<dim3(2176, 4), 256>>>(sf, Ws, dw, dk, 0.1f, df, dk + 512); });
            timeit("feat nodiv", 4, traffic, [&] { featlike<0, 1, 1, 1><<<dim3(2176, 4), 256>>>(sf, Ws, dw, dk, 0.1f, df, dk + 512); });
            timeit("feat nokeys", 4, traffic, [&] { featlike<1, 0, 1, 1><<<dim3(2176, 4), 256>>>(sf, Ws, dw, dk, 0.1f, df, dk + 512); });
            timeit("feat noatom", 4, traffic, [&] { featlike<1, 1, 0, 1><<<dim3(2176, 4), 256>>>(sf, Ws, dw, dk, 0.1f, df, dk + 512); });
            timeit("feat bare", 4, traffic, [&] { featlike<0, 0, 0, 1><<<dim3(2176, 4), 256>>>(sf, Ws, dw, dk, 0.1f, df, dk + 512); });
            timeit("feat unr3", 4, traffic, [&] { featlike<1, 1, 1, 3><<<dim3(2176, 4), 256>>>(sf, Ws, dw, dk, 0.1f, df, dk + 512); });
            timeit("feat u3 half", 4, traffic, [&] { featlike<1, 1, 1, 3><<<dim3(1088, 4), 256>>>(sf, Ws, dw, dk, 0.1f, df, dk + 512); });
            timeit("feat u3 bare", 4, traffic, [&] { featlike<0, 0, 0, 3><<<dim3(2176, 4), 256>>>(sf, Ws, dw, dk, 0.1f, df, dk + 512); });
        }
        for (int nw : {4, 16}) {
            const double traffic = 2.0 * nw * ww * wh * 32.0;
            timeit("win rows", nw, traffic, [&] { wincopy<0><<<dim3(wh, nw), 256>>>(big, b, Ws, ww, wh); });
            timeit("win 8x256", nw, traffic, [&] { wincopy<1><<<dim3(((ww + 255) / 256) * ((wh + 7) / 8), nw), 256>>>(big, b, Ws, ww, wh); });
        }
    }
    return 0;
}
