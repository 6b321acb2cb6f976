// Shader-clock probe (measurement tool, not part of the product libraries): ONE wave on a side stream samples
// (s_memrealtime = constant 100 MHz, s_memtime = shader cycles) every few microseconds while the main stream runs the
// workload; a one-lane stamp kernel on the MAIN stream marks the workload's window in the same 100 MHz time base.
// clock(t) = d(s_memtime) / d(s_memrealtime) x 100 MHz. Each sample also times a fixed chain of dependent v_add_f32 in
// shader cycles: constant unless the probe wave loses issue slots to co-resident waves (a contention check, not a clock).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/clock_probe.hip -o tools/libclockprobe.so
#include <hip/hip_runtime.h>

__global__ void k_clock_probe(unsigned long long* __restrict__ log, int n, int sleeps) {
    if (threadIdx.x != 0) return;
    float v = 1.0f;
    for (int i = 0; i < n; ++i) {
        const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
        const unsigned long long c0 = __builtin_readcyclecounter();
#pragma unroll
        for (int k = 0; k < 256; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
        const unsigned long long c1 = __builtin_readcyclecounter();
        __builtin_nontemporal_store(rt, log + 3 * i);
        __builtin_nontemporal_store(c0, log + 3 * i + 1);
        __builtin_nontemporal_store(c1 - c0, log + 3 * i + 2);
        for (int k = 0; k < sleeps; ++k) __builtin_amdgcn_s_sleep(127);
    }
    if (v == 123.0f) log[0] = 0;
}

__global__ void k_stamp(unsigned long long* __restrict__ dst) {
    if (threadIdx.x == 0) {
        dst[0] = __builtin_amdgcn_s_memrealtime();
        dst[1] = __builtin_readcyclecounter();
    }
}

extern "C" int clockprobe_launch(unsigned long long* log, int n, int sleeps, void* stream) {
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), log, n, sleeps);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int clockprobe_stamp(unsigned long long* dst, void* stream) {
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), dst);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
