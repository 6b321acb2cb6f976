// Persistent chain of conv layers (yolo_conv_chain.hip: k_conv_chain): what the tile bodies of yolo_conv_pp.hip (256 x 256) and
// yolo_conv_pp128.hip (256 x 128) share when they run as work items of ONE launch instead of one launch per layer.
//
// One work item = one tile of one layer, handed out from a work counter in layer-major order. A tile's outputs leave as
// written-through (sc1) stores; when the storing waves' vmcnt has reached 0 the tile's ARRIVAL COUNTER (one per layer and
// m-tile, counting n-tiles) is bumped. A tile waits for the counters of the producer m-tiles its input window and its residual
// rows lie in — it only ever waits for items BEFORE it in the hand-out order, which are held by running workgroups: no deadlock
// for any number of resident workgroups. Tables (host-built, yolo_api.hip) are read through the SCALAR path (constant during a
// launch); counters through agent-scope accesses.
#pragma once
#include "yolo_internal.h"
#ifndef CHAIN_INV_WAIT
#define CHAIN_INV_WAIT 1
#endif

namespace adayolo {

// LDS: both tile bodies end at 147456 + 1 KB of bias; the scheduler words follow (yolo_conv_chain.hip asserts that neither
// tile body's kSmem reaches them)
constexpr int kChainSchedOff = 147456 + 1024;      // int[8]: {next item, its inputs are ready, layer, tile, arrival counter, exit ticket, -, -}
#ifdef ADAYOLO_CHAIN_STAMPS
constexpr int kChainSmem = kChainSchedOff + 32 + 128;
#else
constexpr int kChainSmem = kChainSchedOff + 32;
#endif
// A dependency wait gives up after kChainWaitTicks of the constant 100 MHz s_memrealtime clock (wall time, whatever the
// shader clock and the contention: the old bound counted polling iterations)
constexpr unsigned long long kChainWaitTicks = 100000000ull;     // 1 s

struct ChainCtx {
    const ChainArgs* c;
    int pending;                 // done[] index of this workgroup's previous tile whose arrival is not yet published, or -1
};
typedef const __attribute__((address_space(4))) int* chain_cint_p;     // constant address space: uniform loads are scalar loads
__device__ __forceinline__ int* chain_head(const ChainArgs& c) { return reinterpret_cast<int*>(c.ws); }
__device__ __forceinline__ int* chain_err(const ChainArgs& c) { return reinterpret_cast<int*>(c.ws) + 1; }
__device__ __forceinline__ int* chain_exit(const ChainArgs& c) { return reinterpret_cast<int*>(c.ws) + 2; }
__device__ __forceinline__ int* chain_done(const ChainArgs& c) { return reinterpret_cast<int*>(c.ws + 64); }
__device__ __forceinline__ void chain_load4(const ChainArgs& c, int off, int item, int (&r)[4]) {
    chain_cint_p q = (chain_cint_p)((unsigned long long)c.ws + (unsigned)off + (unsigned long long)(unsigned)item * 16u);
    r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = q[3];
}
// lane l < 32: done[in_lo + l], lane 32 + l: done[res_lo + l] -> this lane's counter (lanes without one: INT_MAX)
__device__ __forceinline__ int chain_counter(const ChainArgs& c, const int (&d)[4], int lane) {
    const int l = lane & 31;
    const bool act = lane < 32 ? l < (d[1] >> 16) : l < (d[3] >> 16);
    int v = 0x7fffffff;
    if (act) v = __hip_atomic_load(chain_done(c) + (lane < 32 ? d[0] : d[2]) + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
__device__ __forceinline__ bool chain_arrived(const int (&d)[4], int v, int lane) {
    return __all(v >= ((lane < 32 ? d[1] : d[3]) & 0xFFFF));
}
// behind a barrier that follows every wave's vmcnt(0): the previous tile's stores are complete everywhere
__device__ __forceinline__ void chain_publish(ChainCtx& cx, int tid) {
    if (cx.pending >= 0 && tid == 0)
        __hip_atomic_fetch_add(chain_done(*cx.c) + cx.pending, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cx.pending = -1;
}

// Wave 0 looks ahead while the tile runs — four stages, each consuming what the previous one requested microseconds earlier
// (three dependent round trips: work counter -> item records -> arrival counters), so that the next tile starts without
// waiting for any of them:
//   0 (behind the prologue)  lane 0 draws the next item from the work counter (in flight across the k-loop)
//   1 (epilogue)             the item's records are requested (scalar loads)
//   2 (epilogue, later)      the arrival counters its input window / residual rows wait for are requested
//   3 (last stores issued)   all arrived -> ONE buffer_inv sc1 (this CU's L1 may hold lines of those tensors from an earlier
//                            forward; the counters have been observed, so the invalidate is the acquire) and {item, ready,
//                            layer, tile, counter} go to the workgroup through LDS. Not arrived (the producers are ~200 items
//                            ahead, so seldom): the slow path at the top of the next tile polls.
struct ChainLook {
    int item = 0x7fffffff, val = 0x7fffffff;
    int deps[4] = {0, 0, 0, 0}, head[4] = {0, 0, 0, 0};
    __device__ __forceinline__ void stage(int s, const ChainArgs& c, unsigned char* sched, int lane) {
        if (s == 0) {
            int t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(chain_head(c), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            item = t;
        } else if (s == 1) {
            item = __builtin_amdgcn_readfirstlane(item);
            if (item < c.total) {
                chain_load4(c, c.off_deps, item, deps);
                chain_load4(c, c.off_heads, item, head);
            }
        } else if (s == 2) {
            if (item < c.total) val = chain_counter(c, deps, lane);
        } else {
            int ready = 0;
            if (item < c.total && chain_arrived(deps, val, lane)) {
                // the acquire. Two hardware facts it leans on: (1) a CU's L1 takes the invalidate and the loads
                // the other seven waves issue behind the tile's end barrier in program order of ARRIVAL, so an invalidate that
                // was issued before the barrier is ahead of them; (2) an sc1 store whose vmcnt has retired is visible on
                // every XCD. CHAIN_INV_WAIT=1 additionally waits for the invalidate's acknowledgement before
                // `ready` is written — it also waits for this wave's own output stores: measured, see DESIGN 9
                asm volatile("buffer_inv sc1" ::: "memory");
#if CHAIN_INV_WAIT
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                ready = 1;
            }
            if (lane == 0) {
                *reinterpret_cast<int4*>(sched) = int4{item, ready, head[0], head[1]};
                *reinterpret_cast<int*>(sched + 16) = head[2];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
};

}  // namespace adayolo
