// Microbenchmark: issue cost of single VALU instructions on gfx950 (wave64), written as inline asm so the optimizer cannot
// merge or pack anything.  Every kernel runs 8 independent dependency chains of one instruction; 1 / 2 / 4 / 8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, DECL, BODY)                                                                  \
    __global__ __launch_bounds__(1024) void NAME(float *out, int iters) {                          \
        DECL                                                                                       \
        for (int i = 0; i < iters; ++i) { BODY BODY BODY BODY }                                    \
        float s = 0; for (int q = 0; q < 8; ++q) s += a[q].x + a[q].y;                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                            \
    }
typedef float v2f __attribute__((ext_vector_type(2)));
#define DECLA v2f a[8]; for (int q = 0; q < 8; ++q) a[q] = (v2f){(float)threadIdx.x + q, 1.0f + q}; const v2f b = {1.000001f, 0.999f};
#define ADD(q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[q].x) : "v"(b.x));
#define MUL(q) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[q].x) : "v"(b.x));
#define FMA(q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[q].x) : "v"(b.x));
#define PKADD(q) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[q]) : "v"(b));
#define PKMUL(q) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[q]) : "v"(b));
#define PKFMA(q) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[q]) : "v"(b));
#define CNDM(q) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[q].x) : "v"(b.x) : );
#define CMP64(q) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(a[q]), "v"(b) : "vcc");
#define CMP32(q) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[q].x), "v"(b.x) : "vcc");
#define DPP(q) asm volatile("v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[q].x));
#define RFL(q) { int t; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(t) : "v"(a[q].x)); asm volatile("" :: "s"(t)); }
#define CVT(q) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[q].x));
#define MOV(q) asm volatile("v_mov_b32 %0, %1" : "+v"(a[q].x) : "v"(b.x));
KERNEL(k_add, DECLA, REP8(ADD))
KERNEL(k_mul, DECLA, REP8(MUL))
KERNEL(k_fma, DECLA, REP8(FMA))
KERNEL(k_pkadd, DECLA, REP8(PKADD))
KERNEL(k_pkmul, DECLA, REP8(PKMUL))
KERNEL(k_pkfma, DECLA, REP8(PKFMA))
KERNEL(k_cndmask, DECLA, REP8(CNDM))
KERNEL(k_cmp64, DECLA, REP8(CMP64))
KERNEL(k_cmp32, DECLA, REP8(CMP32))
KERNEL(k_dpp, DECLA, REP8(DPP))
KERNEL(k_rfl, DECLA, REP8(RFL))
KERNEL(k_cvt, DECLA, REP8(CVT))
KERNEL(k_mov, DECLA, REP8(MOV))
template <typename K> void run(const char *name, K kern) {
    float *d; hipMalloc(&d, 256 * 2048 * sizeof(float));
    const int iters = 4000;
    printf("%-14s", name);
    for (int wpb : {256, 512, 1024}) {           // one block per CU: 1, 2, 4 waves per SIMD; then 2 blocks of 1024 per CU = 8
        for (int nb : {256}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kern, dim3(nb), dim3(wpb), 0, 0, d, 10);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nb), dim3(wpb), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)iters * 32 * (wpb / 256.0);       // wave-instructions per SIMD
            printf("  %d w/SIMD: %.2f cyc", wpb / 256, ms * 1e6 / instr * 2.4);
        }
    }
    printf("   (cycles per wave-instruction per SIMD at 2.4 GHz)\n");
    hipFree(d);
}
int main() {
    run("v_add_f32", k_add); run("v_mul_f32", k_mul); run("v_fma_f32", k_fma);
    run("v_pk_add_f32", k_pkadd); run("v_pk_mul_f32", k_pkmul); run("v_pk_fma_f32", k_pkfma);
    run("v_cndmask", k_cndmask); run("v_cmp_lt_u64", k_cmp64); run("v_cmp_lt_f32", k_cmp32);
    run("v_min_u32_dpp", k_dpp); run("v_readfirstl.", k_rfl); run("v_cvt_i32_f32", k_cvt); run("v_mov_b32", k_mov);
    return 0;
}
