// Microbenchmark: what the ACCESS SHAPE of the zonal-statistics pass costs on gfx950, without its arithmetic.
// Raster [H][W][8] float32 + labels [H][W] int32; a lane owns a 16-byte chunk (four bands of one pixel) and walks down rows.
//   TW  columns per workgroup (lanes = 2 * TW), TH rows per workgroup, R rows in flight per lane.
// hipcc --offload-arch=gfx950 -O3 -o zonal_pattern zonal_pattern.hip && ./zonal_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int TW, int TH, int R, bool NT, int WV = 8>
__global__ __launch_bounds__(2 * TW) __attribute__((amdgpu_waves_per_eu(WV, WV))) void walk(const float *__restrict__ raw, const int *__restrict__ lab, int H, int W, float *out) {
    const int tiles_x = W / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const int x = tx0 + threadIdx.x / 2, q = threadIdx.x & 1;
    float s = 0.f; int ls = 0;
    v4f v[2][R]; int l[2][R];
    auto fetch = [&](int b, int y0) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const long long pix = (long long)(y0 + j) * W + x;
            l[b][j] = lab[pix];
            const v4f *p = reinterpret_cast<const v4f *>(raw + pix * 8 + 4 * q);
            v[b][j] = NT ? __builtin_nontemporal_load(p) : *p;
        }
    };
    fetch(0, ty0);
#pragma unroll 1
    for (int g = 0; g < TH / R; g += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int y0 = ty0 + (g + h) * R;
            if (y0 + R < ty0 + TH) fetch(h ^ 1, y0 + R);
#pragma unroll
            for (int j = 0; j < R; ++j) { s += v[h][j].x + v[h][j].y + v[h][j].z + v[h][j].w; ls += l[h][j]; }
        }
    }
    if (s == 12345.678f || ls == -77) out[0] = s;
}
template <int TW, int TH, int R, bool NT, int WV = 8>
static void run(const char *name, const float *raw, const int *lab, int H, int W, float *out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = (W / TW) * (H / TH);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((walk<TW, TH, R, NT, WV>), dim3(blocks), dim3(2 * TW), 0, 0, raw, lab, H, W, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms;
    }
    printf("%-34s %7.3f ms  %6.0f GB/s\n", name, best, (double)H * W * 36 / best / 1e6);
}
int main() {
    const int H = 16384, W = 16384;
    float *raw, *out; int *lab;
    hipMalloc(&raw, (size_t)H * W * 32); hipMalloc(&lab, (size_t)H * W * 4); hipMalloc(&out, 64);
    hipMemset(raw, 0, (size_t)H * W * 32); hipMemset(lab, 0, (size_t)H * W * 4);
    run<64, 64, 2, false>("64x64 R2 (zonal_kernel today)", raw, lab, H, W, out);
    run<64, 64, 2, true>("64x64 R2 nt", raw, lab, H, W, out);
    run<64, 64, 4, false>("64x64 R4", raw, lab, H, W, out);
    run<64, 64, 8, false>("64x64 R8", raw, lab, H, W, out);
    run<128, 64, 2, false>("128x64 R2", raw, lab, H, W, out);
    run<128, 64, 4, false>("128x64 R4", raw, lab, H, W, out);
    run<128, 32, 4, false>("128x32 R4", raw, lab, H, W, out);
    run<256, 32, 4, false>("256x32 R4", raw, lab, H, W, out);
    run<256, 16, 4, false>("256x16 R4", raw, lab, H, W, out);
    run<512, 16, 4, false>("512x16 R4", raw, lab, H, W, out);
    run<512, 16, 8, false>("512x16 R8", raw, lab, H, W, out);
    run<64, 256, 4, false>("64x256 R4", raw, lab, H, W, out);
    run<32, 64, 4, false>("32x64 R4", raw, lab, H, W, out);
    run<128, 128, 4, true>("128x128 R4 nt", raw, lab, H, W, out);
    run<128, 64, 2, true, 8>("128x64 R2 nt 8 waves", raw, lab, H, W, out);
    run<128, 64, 2, true, 6>("128x64 R2 nt 6 waves", raw, lab, H, W, out);
    run<128, 64, 2, true, 5>("128x64 R2 nt 5 waves", raw, lab, H, W, out);
    run<128, 64, 2, true, 4>("128x64 R2 nt 4 waves", raw, lab, H, W, out);
    run<128, 64, 4, true, 4>("128x64 R4 nt 4 waves", raw, lab, H, W, out);
    run<128, 64, 4, true, 3>("128x64 R4 nt 3 waves", raw, lab, H, W, out);
    run<128, 64, 8, true, 2>("128x64 R8 nt 2 waves", raw, lab, H, W, out);
    run<128, 64, 1, true, 8>("128x64 R1 nt 8 waves", raw, lab, H, W, out);
    run<64, 64, 2, true, 6>("64x64 R2 nt 6 waves", raw, lab, H, W, out);
    return 0;
}
