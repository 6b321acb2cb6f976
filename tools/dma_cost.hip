// Issue cost of an LDS-DMA instruction among MFMAs on gfx950: global_load_lds_dwordx4 (64-bit vaddr) vs
// buffer_load_dwordx4 ... offen lds (SGPR resource + 32-bit voffset). One wave per SIMD, 4 MFMAs + 1 DMA per trip.
// hipcc --offload-arch=gfx950 -O3 tools/dma_cost.hip -o /tmp/dma_cost && /tmp/dma_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const unsigned char* src, float* out, unsigned long long* cyc, unsigned nbytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned off = ((blockIdx.x * WAVES + wave) * 64 + lane) * 16;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 4096; ++it) {
        unsigned char* dst = lds + wave * 16384 + (it & 15) * 1024;
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            if (i == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 1) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off), (lds_ptr_t)dst, 16, 0, 0);
                if (MODE == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)dst, 16, off, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        off += 256 * WAVES * 64 * 16;
        if (off >= nbytes) off -= nbytes;
        if (MODE) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE, int WAVES>
void run(const char* name, const unsigned char* src, unsigned nbytes) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    auto kern = k<MODE, WAVES>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WAVES), 131072, 0, src, out, cyc, nbytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WAVES), 131072, 0, src, out, cyc, nbytes);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long c[256 * 8];
    hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < WAVES; ++w) { sum += (double)c[b * 8 + w]; ++n; }
    const double ticks = sum / n;
    printf("%-44s %d waves/CU: %7.1f ticks per trip; kernel %.1f us wall -> %.2f GHz tick rate if the loop is the kernel; %.0f TFLOP/s\n", name, WAVES,
           ticks / 4096.0, ms * 100.0, ticks / (ms * 100.0) / 1e3, 256.0 * WAVES * 4096 * 4 * 32768.0 / (ms / 10 * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}
int main() {
    const unsigned nbytes = 1u << 24;       // 16 MB: L2-resident after the first sweep (1 << 30 measures HBM: 13 B/clk/CU)
    unsigned char* src; hipMalloc(&src, nbytes); hipMemset(src, 1, nbytes);
    run<0, 4>("MFMAs only", src, nbytes);
    run<1, 4>("+ global_load_lds_dwordx4 (64-bit vaddr)", src, nbytes);
    run<2, 4>("+ buffer_load_dwordx4 offen lds", src, nbytes);
    run<0, 8>("MFMAs only", src, nbytes);
    run<1, 8>("+ global_load_lds_dwordx4 (64-bit vaddr)", src, nbytes);
    run<2, 8>("+ buffer_load_dwordx4 offen lds", src, nbytes);
    return 0;
}
