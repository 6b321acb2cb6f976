// Issue cost of single VALU instructions on gfx950: one wave, 16 independent chains, s_memtime around 2048 x 16 issues.
// hipcc --offload-arch=gfx950 -O3 tools/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int OP>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
    float r[16];
    v2f p[16];
    for (int i = 0; i < 16; ++i) { r[i] = seed + i * 0.01f + threadIdx.x * 1e-3f; p[i] = v2f{r[i], r[i] * 0.5f}; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 2048; ++it) {
#define ONE(i)                                                                                                        \
    if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));                                         \
    if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));                                                         \
    if (OP == 2) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));                                                        \
    if (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));                                                         \
    if (OP == 4) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(r[i]) : "v"(seed)); \
    if (OP == 5) asm volatile("v_add_f32_dpp %0, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(r[i]) : "v"(seed));  \
    if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));                     \
    if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));                         \
    if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(seed));                                    \
    if (OP == 9) asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(r[(i + 1) & 15])); \
    if (OP == 10) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
        REP16(ONE)
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += r[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[threadIdx.x >> 6] = t1; cyc[16 + (threadIdx.x >> 6)] = t0; }
}
template <int OP>
void run(const char* name, int waves) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64 * 8);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
    hipDeviceSynchronize();
    unsigned long long c[32]; hipMemcpy(c, cyc, 32 * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < waves; ++i) { if (c[16 + i] < tmin) tmin = c[16 + i]; if (c[i] > tmax) tmax = c[i]; }
    printf("%-28s %2d wave(s)/CU: %6.2f ticks per instruction per SIMD (first start to last end, %d waves per SIMD)\n", name, waves,
           (double)(tmax - tmin) / (2048.0 * 16 * ((waves + 3) / 4)), (waves + 3) / 4);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 4, 8, 12, 16}) {
        if (w == 1) { run<0>("v_add_f32", 1); run<8>("v_fma_f32", 1); run<7>("v_pk_add_f32", 1); run<6>("v_pk_fma_f32", 1); run<1>("v_exp_f32", 1); run<2>("v_sqrt_f32", 1); run<3>("v_rcp_f32", 1); run<4>("v_add_f32_dpp wave_shl:1", 1); run<5>("v_add_f32_dpp row_shl:1", 1); run<9>("v_mov_b32_dpp wave_shl:1", 1); run<10>("v_cvt_pk_bf16_f32", 1); }
        if (w == 4) { run<0>("v_add_f32", 4); run<1>("v_exp_f32", 4); run<4>("v_add_f32_dpp wave_shl:1", 4); run<6>("v_pk_fma_f32", 4); }
        if (w == 8) { run<0>("v_add_f32", 8); run<1>("v_exp_f32", 8); run<2>("v_sqrt_f32", 8); run<4>("v_add_f32_dpp wave_shl:1", 8); run<5>("v_add_f32_dpp row_shl:1", 8); run<6>("v_pk_fma_f32", 8); run<7>("v_pk_add_f32", 8); }
        if (w == 12) { run<0>("v_add_f32", 12); run<1>("v_exp_f32", 12); run<4>("v_add_f32_dpp wave_shl:1", 12); run<6>("v_pk_fma_f32", 12); }
        if (w == 16) { run<0>("v_add_f32", 16); run<8>("v_fma_f32", 16); run<1>("v_exp_f32", 16); run<4>("v_add_f32_dpp wave_shl:1", 16); run<6>("v_pk_fma_f32", 16); run<7>("v_pk_add_f32", 16); run<10>("v_cvt_pk_bf16_f32", 16); }
    }
    return 0;
}
