// Microbenchmark: wave64 VALU issue rate on gfx950 as a function of waves per SIMD and instruction mix.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.000001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 8 independent FMA chains
            a0 = a0 * b + c; a1 = a1 * b + c; a2 = a2 * b + c; a3 = a3 * b + c;
            a4 = a4 * b + c; a5 = a5 * b + c; a6 = a6 * b + c; a7 = a7 * b + c;
        } else if (MODE == 1) {   // separate mul + add (contraction off), like the distance arithmetic
            a0 = a0 * b; a0 = a0 + c; a1 = a1 * b; a1 = a1 + c; a2 = a2 * b; a2 = a2 + c; a3 = a3 * b; a3 = a3 + c;
        } else {   // compare + select mix
            a0 = (a0 < a1) ? a2 : a0; a1 = (a1 < a2) ? a3 : a1; a2 = (a2 < a3) ? a4 : a2; a3 = (a3 < a4) ? a5 : a3;
            a4 = a4 * b + c; a5 = a5 * b + c; a6 = a6 * b + c; a7 = a7 * b + c;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE>
void run(const char *name, int instr_per_iter) {
    float *d; hipMalloc(&d, 256 * 1024 * 8 * sizeof(float));
    const int iters = 20000;
    for (int wpb : {256, 512, 1024}) {           // 256 threads = 1 wave/SIMD, 512 = 2, 1024 = 4 (one block per CU)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(wpb), 0, 0, d, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(wpb), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double waves_per_simd = wpb / 256.0;
        const double instr = (double)iters * instr_per_iter * waves_per_simd;       // per SIMD
        printf("%-28s waves/SIMD %.0f: %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
               ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
    }
    hipFree(d);
}
int main() {
    run<0>("8 independent v_fma_f32", 8);
    run<1>("mul+add pairs (no fma)", 8);
    run<2>("cmp/cndmask + fma mix", 12);
    return 0;
}
