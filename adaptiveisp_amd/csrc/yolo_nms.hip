// Greedy IoU non-maximum suppression for the eval harness (reference call site: yolov3/utils/general.py:949,
// `torchvision.ops.nms(boxes, scores, iou_thres)` on boxes the caller has already sorted by descending score and
// offset by class). torchvision is not vendored by the reference; the semantics restated here are its documented ones:
// walk the boxes in score order, keep a box unless an earlier KEPT box overlaps it with IoU > threshold,
// IoU = inter / (area_a + area_b - inter), areas and intersections from (x2-x1)*(y2-y1) with the intersection
// extents clamped at 0.
//
// Two launches, no host round trip:
//   k_nms_mask  one wave per (row block, column block) pair of 64 boxes: lane r holds row box r, the 64 column boxes
//               sit in LDS, and the lane produces the 64-bit word "which column boxes does my box suppress" — a
//               wave64 ballot-shaped word per lane, written as mask[row][column block];
//   k_nms_scan  one workgroup walks the boxes in order, 64 at a time, with the running `removed` bit set in LDS. A block of 64
//               whose boxes are all removed already costs one LDS read. Otherwise its 64 diagonal mask words are fetched (one
//               memory round trip), the block is resolved sequentially in registers — the greedy order exactly — and the mask
//               rows of the boxes it KEEPS are OR-ed into the later blocks' words by all threads at once (one more round trip,
//               every load independent). Round 6: the form before walked box by box, one dependent round trip per kept box —
//               0.63 ms for 300 keeps among 30 000 candidates (config 3's loop at conf 0.001); the scan stops after max_det keeps.
#include "yolo_internal.h"

namespace adayolo {

__device__ __forceinline__ float box_iou_tv(const float4 a, const float4 b) {
    const float area_a = (a.z - a.x) * (a.w - a.y), area_b = (b.z - b.x) * (b.w - b.y);
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.0f);
    const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.0f);
    const float inter = w * h;
    return inter / (area_a + area_b - inter);
}

__global__ __launch_bounds__(64) void k_nms_mask(const float4* __restrict__ boxes, int n, float thr, int nb,
                                                 unsigned long long* __restrict__ mask) {
    const int rb = blockIdx.y, cb = blockIdx.x, lane = threadIdx.x;
    const int row = rb * 64 + lane;
    if (cb < rb) {                                   // earlier boxes are decided before this row is consulted
        if (row < n) mask[(long)row * nb + cb] = 0ull;
        return;
    }
    __shared__ float4 col[64];
    const int c = cb * 64 + lane;
    col[lane] = c < n ? boxes[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (row >= n) return;
    const float4 me = boxes[row];
    unsigned long long bits = 0ull;
    const int ncol = min(64, n - cb * 64);
    // (most pairs do not intersect at all — boxes of different classes are 7680 px apart by construction: their IoU is 0 / union,
    // never above a threshold >= 0, and the division is skipped: the same bits as evaluating box_iou_tv for every pair)
    const float area_me = (me.z - me.x) * (me.w - me.y);
    for (int j = 0; j < ncol; ++j) {
        const int cj = cb * 64 + j;
        const float4 o = col[j];
        const float w = fmaxf(fminf(me.z, o.z) - fmaxf(me.x, o.x), 0.0f);
        const float h = fmaxf(fminf(me.w, o.w) - fmaxf(me.y, o.y), 0.0f);
        const float inter = w * h;
        if (cj > row && inter > 0.0f) {
            const float area_o = (o.z - o.x) * (o.w - o.y);
            if (inter / (area_me + area_o - inter) > thr) bits |= 1ull << j;
        }
    }
    mask[(long)row * nb + cb] = bits;
}

constexpr int kScanThreads = 512;
__global__ __launch_bounds__(kScanThreads) void k_nms_scan(const unsigned long long* __restrict__ mask, int n, int nb,
                                                            int max_det, int* __restrict__ keep, int* __restrict__ num_keep) {
    extern __shared__ unsigned long long removed[];  // [nb] running "suppressed" bits, then [64] diagonal words
    unsigned long long* const diag = removed + nb;
    const int tid = threadIdx.x;
    for (int w = tid; w < nb; w += kScanThreads) removed[w] = 0ull;
    __syncthreads();
    int count = 0;
    for (int b = 0; b < nb && count < max_det; ++b) {
        const int nvalid = min(64, n - b * 64);
        const unsigned long long valid = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        const unsigned long long word = removed[b];                // broadcast read (uniform)
        if ((~word & valid) == 0ull) continue;                     // every box of the block is suppressed already: no memory access
        if (tid < 64) diag[tid] = tid < nvalid ? mask[(long)(b * 64 + tid) * nb + b] : 0ull;
        __syncthreads();
        // the greedy walk inside the block, identically in every thread (64 steps on LDS broadcasts)
        unsigned long long cur = word, kept = 0ull;
        int c = count;
        for (int r = 0; r < nvalid && c < max_det; ++r) {
            if (!((cur >> r) & 1ull)) {
                kept |= 1ull << r;
                cur |= diag[r];
                if (tid == 0) keep[c] = b * 64 + r;
                ++c;
            }
        }
        count = c;
        // OR the kept boxes' mask rows into the words of the later blocks: thread -> word, loop over the kept rows (independent loads)
        if (count < max_det) {
            for (int w = b + 1 + tid; w < nb; w += kScanThreads) {
                unsigned long long acc = 0ull, k = kept;
                while (k) {
                    const int r = __builtin_ctzll(k);
                    k &= k - 1ull;
                    acc |= mask[(long)(b * 64 + r) * nb + w];
                }
                removed[w] |= acc;
            }
        }
        __syncthreads();
    }
    if (tid == 0) *num_keep = count;
    for (int k = count + tid; k < max_det; k += kScanThreads) keep[k] = -1;
}

hipError_t launch_nms(const float* boxes, int n, float thr, int max_det, unsigned long long* mask_ws, int* keep,
                      int* num_keep, hipStream_t s) {
    const int nb = (n + 63) / 64;
    if (n > 0) hipLaunchKernelGGL(k_nms_mask, dim3(nb, nb), dim3(64), 0, s, reinterpret_cast<const float4*>(boxes), n, thr, nb, mask_ws);
    hipLaunchKernelGGL(k_nms_scan, dim3(1), dim3(kScanThreads), (size_t)((nb > 0 ? nb : 1) + 64) * 8, s, mask_ws, n, nb, max_det, keep, num_keep);
    return hipGetLastError();
}

}  // namespace adayolo
