// Persistent chain kernel (see yolo_chain.h): tiles of several consecutive conv layers — 256 x 256 tiles of yolo_conv_pp.hip
// (alone or with the next block's 1x1 fused) and 256 x 128 tiles of yolo_conv_pp128.hip — as work items of ONE launch.
// The two tile bodies are compiled into this translation unit under their own namespaces (the launch-per-layer kernels of
// the two files are untouched).
#define ADAYOLO_TILE_ONLY
#define pp ppc
#include "yolo_conv_pp.hip"
#undef pp
#define pp128 pp128c
#include "yolo_conv_pp128.hip"
#undef pp128
#undef ADAYOLO_TILE_ONLY

namespace adayolo {
namespace chain {

using ppc::barrier;
using ppc::wait_vm;
constexpr int kSmemChain = kChainSmem;
// the scheduler words sit behind BOTH tile bodies' LDS (ring / epilogue overlay + bias): a change to a tile's pitch, BN or ring
// must move kChainSchedOff with it
static_assert(ppc::kSmem <= kChainSchedOff && pp128c::kSmem <= kChainSchedOff, "the chain's scheduler words overlap a tile body's LDS");
typedef chain_cint_p cint_p;
#ifdef ADAYOLO_CHAIN_STAMPS
__device__ unsigned long long g_chain_acc[16];
#endif

// Persistent form: one workgroup per CU draws tiles of SEVERAL consecutive layers from one work counter (layer-major order)
// until none is left. What a launch per layer costs and this does not: the ramp and tail of every launch (~4 us x layers), the
// 80 - 90 % full last round of every layer (460 tiles on 256 CUs), and every CU being in the same phase at the same time — the
// workgroups drift apart, so one CU's prologue / residual / store bursts meet other CUs' k-loops instead of 255 other bursts.
// Dependencies: a tile waits for the m-tiles of the producing layer its input window and its residual rows lie in (arrival
// counters, bumped when a tile's written-through stores are complete). It only ever waits for items that come before it in
// the hand-out order, and those are held by workgroups that are running: no deadlock whatever number of workgroups is resident.
__global__ __launch_bounds__(512) void k_conv_chain(const ChainArgs c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int* sched = reinterpret_cast<int*>(smem + kChainSchedOff);
    if (c.stagger > 0) {
        // de-phase the CUs once: left alone they stay in lock-step (equal tiles), every prologue / residual / store burst of the
        // chip at the same moment. Eight start groups per XCD, `stagger` cycles apart; the delay is paid once per launch.
        const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) & 7) * (unsigned)c.stagger;
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    if (tid == 0) {
        sched[0] = __hip_atomic_fetch_add(chain_head(c), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sched[1] = 0;
    }
    __syncthreads();
    int item = __builtin_amdgcn_readfirstlane(sched[0]), ready = 0;
    ChainCtx cx{&c, -1};
    // {layer, tile, arrival counter} of the item: read here for a workgroup's first item, handed over through LDS by the
    // look-ahead afterwards; the layer's arguments stay in their SGPRs while consecutive items belong to the same layer
    int hd[4] = {0, 0, 0, 0};
    if (item < c.total) chain_load4(c, c.off_heads, item, hd);
    ConvArgs a;
    int cur_layer = -1;
#ifdef ADAYOLO_CHAIN_STAMPS
    constexpr bool CHAIN = true; constexpr int ABL = 0;
    if (tid == 0) {
        unsigned long long* acc_ = reinterpret_cast<unsigned long long*>(smem + kChainSchedOff + 32);
        for (int i = 0; i < 15; ++i) acc_[i] = 0;
        acc_[15] = __builtin_readcyclecounter();
    }
#endif
    while (item < c.total) {
#ifdef ADAYOLO_CHAIN_STAMPS
        if (tid == 0) reinterpret_cast<unsigned long long*>(smem + kChainSchedOff + 32)[14] += 1;      // tiles
#endif
        if (!ready) {
            // slow path (a workgroup's first item, or the look-ahead found a counter short): publish what this workgroup still
            // holds back — a waiting workgroup must not sit on a finished tile others may need — then poll, bounded
            if (cx.pending >= 0) {
                wait_vm<0>();
                barrier();
                if (tid == 0) __hip_atomic_fetch_add(chain_done(c) + cx.pending, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cx.pending = -1;
            }
            if (wave == 0) {
                int d[4];
                chain_load4(c, c.off_deps, item, d);
                // bounded in WALL time: a wait that gives up records its item; once ANY wait of the launch has given up no other one
                // spins (the launch then finishes within one limit, wrong — and says so: the code reaches the sticky word and the
                // pinned host word, YoloEngine raises at its next forward / sync point: adayolo_conv_chain_poll / _status)
                const unsigned long long t_wait0 = __builtin_amdgcn_s_memrealtime();
                while (!chain_arrived(d, chain_counter(c, d, lane), lane)) {
                    const int e = __hip_atomic_load(chain_err(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (e != 0) break;
                    if (__builtin_amdgcn_s_memrealtime() - t_wait0 > kChainWaitTicks) {
                        if (lane == 0) __hip_atomic_store(chain_err(c), item + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(32);
                }
                // (an item whose input and residual were complete before the launch reads nothing another workgroup of THIS launch
                // wrote: the launch boundary has made it visible, no acquire — the first item of most workgroups)
                if (((d[1] | d[3]) >> 16) != 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
#ifdef ADAYOLO_CHAIN_STAMPS
            PP_STAMP(12);                                    // slow path (cycles), [13] = how often
            if (tid == 0) reinterpret_cast<unsigned long long*>(smem + kChainSchedOff + 32)[13] += 1;
#endif
        }
        // the layer's arguments through the scalar path (uniform index, constant table): SGPRs, as kernel arguments would be
        if (hd[0] != cur_layer) {
            cur_layer = hd[0];
            cint_p q = (cint_p)((unsigned long long)c.ws + (unsigned)c.off_layers + (unsigned long long)(unsigned)hd[0] * sizeof(ConvArgs));
            int w[sizeof(ConvArgs) / 4];
#pragma unroll
            for (int i = 0; i < (int)(sizeof(ConvArgs) / 4); ++i) w[i] = q[i];
            __builtin_memcpy(&a, w, sizeof(ConvArgs));
        }
#ifdef ADAYOLO_CHAIN_STAMPS
        PP_STAMP(10);                                        // loop top: item record, slow path, layer arguments
#endif
        if (a.chain_tile == 1) pp128c::conv_tile<0, false, true>(a, hd[1], smem, cx);
        else if (a.w2) ppc::conv_tile<0, true, true>(a, hd[1], smem, cx);
        else ppc::conv_tile<0, false, true>(a, hd[1], smem, cx);
        cx.pending = hd[2];
        item = __builtin_amdgcn_readfirstlane(sched[0]);
        ready = __builtin_amdgcn_readfirstlane(sched[1]);
        hd[0] = __builtin_amdgcn_readfirstlane(sched[2]);
        hd[1] = __builtin_amdgcn_readfirstlane(sched[3]);
        hd[2] = __builtin_amdgcn_readfirstlane(sched[4]);
    }
    if (cx.pending >= 0) {
        wait_vm<0>();
        barrier();
        if (tid == 0) __hip_atomic_fetch_add(chain_done(c) + cx.pending, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The counters are left ZERO for the next launch by the workgroup that leaves last (adayolo_conv_chain_prepare zeroes them
    // once). Not a memset ahead of the launch: captured into a hipGraph, a memset node came out WITHOUT its dependency on the
    // preceding nodes (round 5: the replayed chain ran beside the kernels in front of it). A workgroup's own arrivals are
    // complete (vmcnt) before it signs off, so nothing can land on a counter after the last one has zeroed it.
#ifdef ADAYOLO_CHAIN_STAMPS
    if (tid == 0) {
        const unsigned long long* acc_ = reinterpret_cast<const unsigned long long*>(smem + kChainSchedOff + 32);
        for (int i = 0; i < 15; ++i) atomicAdd(&g_chain_acc[i], acc_[i]);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) sched[5] = __hip_atomic_fetch_add(chain_exit(c), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (sched[5] == (int)gridDim.x - 1) {
        int* done = chain_done(c);
        for (int i = tid; i < c.ndone; i += 512) __hip_atomic_store(done + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(chain_head(c), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(chain_exit(c), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // a wait that gave up stops the waits of ITS launch only: the code moves to the sticky word adayolo_conv_chain_status
            // reads, and the next launch starts with a clean one
            const int e = __hip_atomic_load(chain_err(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (e != 0) {
                __hip_atomic_store(chain_err(c) + 2, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(chain_err(c), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // ... and to the pinned host word the production path polls without a device round trip
                if (c.host_err) __hip_atomic_store(c.host_err, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

static hipError_t launch_chain(const ChainArgs& c, int grid, hipStream_t s) {
    static_assert(kSmemChain <= 160 * 1024, "LDS budget");
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_chain), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kSmemChain);
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL(k_conv_chain, dim3(grid), dim3(512), kSmemChain, s, c);
    return hipGetLastError();
}


}  // namespace chain

// Persistent chain of 256 x 256-tile layers (k_conv_chain): ONE kernel node; the counters at the head of the workspace are zero
// when it starts (adayolo_conv_chain_prepare, then every launch's last workgroup).
hipError_t launch_conv_chain(const ChainArgs& c, int grid, hipStream_t s) {
    if (!c.ws || c.total <= 0 || grid <= 0 || c.ndone < 0) return hipErrorInvalidValue;
    return chain::launch_chain(c, grid < c.total ? grid : c.total, s);
}


#ifdef ADAYOLO_CHAIN_STAMPS
// measurement helper (not part of the ABI): reads and clears the chain kernels' phase accumulators
extern "C" int adayolo_debug_chain_stamps(unsigned long long* dst) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(chain::g_chain_acc), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(chain::g_chain_acc), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace adayolo
