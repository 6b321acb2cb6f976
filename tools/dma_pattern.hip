// L2 -> LDS DMA rate of one CU against the ADDRESS PATTERN of a wave-instruction (global_load_lds_dwordx4: 64 lanes x 16 B
// = 1 KB): fully contiguous, or 8 rows of 128 B at a row stride (how the conv kernels fetch activation rows — stride =
// pixel pitch — and weight rows — stride = K * 2 bytes). All 256 CUs, 8 waves per CU, <= 8 instructions in flight per
// wave, a 1 MB region every CU re-reads (L2-resident after the first sweep).
// hipcc --offload-arch=gfx950 -O3 tools/dma_pattern.hip -o /tmp/dma_pattern && /tmp/dma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int DEPTH>
__global__ __launch_bounds__(512) void k(const unsigned char* src, unsigned long long* cyc, int stride, int iters, unsigned region, unsigned per_cu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // stride == 0: contiguous 1 KB per instruction; else 8 rows x 128 B, rows `stride` bytes apart
    const unsigned lane_off = stride == 0 ? lane * 16u : (unsigned)(lane >> 3) * (unsigned)stride + (lane & 7) * 16u;
    const unsigned step = stride == 0 ? 1024u : 8u * (unsigned)stride;      // next instruction: the next 8 rows
    // per_cu != 0: every CU streams its OWN region (distinct lines per CU, like activation rows); 0: all CUs the same one (weights)
    src += (size_t)blockIdx.x * per_cu;
    unsigned off = (unsigned)wave * 8u * step;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + (off + lane_off) % region), (lds_ptr_t)(lds + wave * 16384 + (it & 15) * 1024), 16, 0, 0);
        off += 64u * step;                                                   // 8 waves x 8 instructions apart
        if (off >= region) off -= region;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int DEPTH>
static void sweep(const unsigned char* src, unsigned long long* cyc, unsigned region, unsigned per_cu, const char* what) {
    const int iters = 4096;
    for (int stride : {0, 2304}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<DEPTH>, dim3(256), dim3(512), 131072, 0, src, cyc, stride, iters, region, per_cu);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<DEPTH>, dim3(256), dim3(512), 131072, 0, src, cyc, stride, iters, region, per_cu);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long c[256 * 8];
        hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        double sum = 0; for (int i = 0; i < 256 * 8; ++i) sum += (double)c[i];
        const double ticks = sum / (256 * 8);
        const double bytes_per_cu = 8.0 * iters * 1024.0;
        printf("%-28s <= %d in flight per wave, row stride %5d B: %6.1f B/clk/CU  (%.2f TB/s chip-wide, %.0f cycles per wave-instruction)\n", what, DEPTH + 1,
               stride, bytes_per_cu / ticks, 256.0 * bytes_per_cu / (ms * 1e-3) / 1e12, ticks / iters);
    }
}

int main() {
    const unsigned region = 1u << 20;                   // per stream: 1 MB, L2-resident after the first sweep
    unsigned char* src; hipMalloc(&src, 257u * region); hipMemset(src, 1, 257u * region);
    unsigned long long* cyc; hipMalloc(&cyc, 256 * 8 * 8);
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    // all CUs re-read ONE region (what weights look like) ...
    sweep<7>(src, cyc, region, 0, "one region for all CUs,");
    sweep<3>(src, cyc, region, 0, "one region for all CUs,");
    sweep<1>(src, cyc, region, 0, "one region for all CUs,");
    sweep<0>(src, cyc, region, 0, "one region for all CUs,");
    // ... or every CU its own 1 MB (what activation rows look like: 256 MB in all = the Infinity Cache, not the 4 MB L2s)
    sweep<7>(src, cyc, region, region, "a region per CU,");
    sweep<3>(src, cyc, region, region, "a region per CU,");
    // ... or every CU its own 64 KB (32 CUs x 64 KB = 2 MB per XCD: L2-resident AND distinct per CU)
    sweep<7>(src, cyc, 1u << 16, 1u << 16, "64 KB per CU (L2-resident),");
    sweep<3>(src, cyc, 1u << 16, 1u << 16, "64 KB per CU (L2-resident),");
    sweep<1>(src, cyc, 1u << 16, 1u << 16, "64 KB per CU (L2-resident),");
    return 0;
}
