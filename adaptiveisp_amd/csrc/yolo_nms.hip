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
//   k_nms_scan  one wave walks the boxes in order with the running `removed` bit set in LDS; only KEPT boxes (at most
//               max_det of them) touch their mask row, so the scan stops after max_det keeps.
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
    for (int j = 0; j < ncol; ++j) {
        const int cj = cb * 64 + j;
        if (cj > row && box_iou_tv(me, col[j]) > thr) bits |= 1ull << j;
    }
    mask[(long)row * nb + cb] = bits;
}

__global__ __launch_bounds__(64) void k_nms_scan(const unsigned long long* __restrict__ mask, int n, int nb,
                                                 int max_det, int* __restrict__ keep, int* __restrict__ num_keep) {
    extern __shared__ unsigned long long removed[];  // [nb]
    const int lane = threadIdx.x;
    for (int w = lane; w < nb; w += 64) removed[w] = 0ull;
    __syncthreads();
    int count = 0;
    for (int i = 0; i < n && count < max_det; ++i) {
        const unsigned long long word = removed[i >> 6];          // broadcast read
        if ((word >> (i & 63)) & 1ull) continue;                   // wave-uniform
        if (lane == 0) keep[count] = i;
        ++count;
        for (int w = (i >> 6) + lane; w < nb; w += 64) removed[w] |= mask[(long)i * nb + w];
        __syncthreads();
    }
    if (lane == 0) *num_keep = count;
    for (int k = count + lane; k < max_det; k += 64) keep[k] = -1;
}

hipError_t launch_nms(const float* boxes, int n, float thr, int max_det, unsigned long long* mask_ws, int* keep,
                      int* num_keep, hipStream_t s) {
    const int nb = (n + 63) / 64;
    if (n > 0) hipLaunchKernelGGL(k_nms_mask, dim3(nb, nb), dim3(64), 0, s, reinterpret_cast<const float4*>(boxes), n, thr, nb, mask_ws);
    hipLaunchKernelGGL(k_nms_scan, dim3(1), dim3(64), (size_t)(nb > 0 ? nb : 1) * 8, s, mask_ws, n, nb, max_det, keep, num_keep);
    return hipGetLastError();
}

}  // namespace adayolo
