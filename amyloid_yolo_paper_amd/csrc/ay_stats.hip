// Evaluation statistics on the device: the greedy true-positive matching of
// get_batch_statistics (utils/utils.py:154-190).  One wavefront per image walks that image's detections in order (the
// greedy choice is sequential by definition); for each detection the 64 lanes scan the image's targets, a wave reduction
// gives the first maximum of the +1-pixel IoU (torch.max semantics), and the detection is a true positive iff that IoU
// reaches the threshold and that target has not been claimed yet.  The reference's two early-outs are kept: stop once
// every target is claimed; skip detections whose class is not among the image's target classes.
#include "ay_common.h"

namespace ay {

__device__ __forceinline__ float iou_p1_s(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2, float by2) {
    const float ix1 = fmaxf(ax1, bx1), iy1 = fmaxf(ay1, by1);
    const float ix2 = fminf(ax2, bx2), iy2 = fminf(ay2, by2);
    const float inter = fmaxf(ix2 - ix1 + 1.0f, 0.0f) * fmaxf(iy2 - iy1 + 1.0f, 0.0f);
    const float a1 = (ax2 - ax1 + 1.0f) * (ay2 - ay1 + 1.0f);
    const float a2 = (bx2 - bx1 + 1.0f) * (by2 - by1 + 1.0f);
    return inter / (a1 + a2 - inter + 1e-16f);
}

constexpr int MAX_T = 2048;  // targets per image held in LDS

// rows [B][max_det][7] (x1,y1,x2,y2,conf,cls_conf,cls_pred), count[B]; targets [nT][6] (sample, class, x1,y1,x2,y2)
__global__ void __launch_bounds__(64) match_detections_kernel(const float* __restrict__ rows, const int* __restrict__ count, int max_det,
                                                             const float* __restrict__ targets, int nT, float iou_thres,
                                                             float* __restrict__ tp, int* __restrict__ overflow) {
    __shared__ float tb[MAX_T][5];           // class, x1, y1, x2, y2
    __shared__ unsigned char claimed[MAX_T];
    __shared__ int n_s;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (lane == 0) n_s = 0;
    __syncthreads();
    // this image's targets, in their original order (the reference's boolean mask keeps the order): one lane compacts
    if (lane == 0) {
        int n = 0;
        for (int t = 0; t < nT; ++t)
            if (targets[(size_t)t * 6] == (float)b) {
                if (n < MAX_T) {
                    for (int k = 0; k < 5; ++k) tb[n][k] = targets[(size_t)t * 6 + 1 + k];
                    claimed[n] = 0;
                }
                ++n;
            }
        if (n > MAX_T) {
            atomicExch(overflow, 1);
            n = MAX_T;
        }
        n_s = n;
    }
    __syncthreads();
    const int n = n_s;
    int nd = count[b];
    if (nd > max_det) nd = max_det;
    const float* rb = rows + (size_t)b * max_det * 7;
    float* tpb = tp + (size_t)b * max_det;
    for (int i = lane; i < max_det; i += 64) tpb[i] = 0.f;
    if (n == 0) return;
    int n_claimed = 0;
    for (int i = 0; i < nd; ++i) {
        if (n_claimed == n) break;
        const float x1 = rb[i * 7], y1 = rb[i * 7 + 1], x2 = rb[i * 7 + 2], y2 = rb[i * 7 + 3], label = rb[i * 7 + 6];
        float best = -1.f;
        int arg = 0x7fffffff;
        bool has_label = false;
        for (int t = lane; t < n; t += 64) {
            has_label = has_label || (tb[t][0] == label);
            const float v = iou_p1_s(x1, y1, x2, y2, tb[t][1], tb[t][2], tb[t][3], tb[t][4]);
            if (v > best) {  // first maximum within the lane's strided subsequence
                best = v;
                arg = t;
            }
        }
        if (!__any(has_label)) continue;  // `pred_label not in target_labels`
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {  // first maximum overall: larger IoU wins, ties go to the lower index
            const float ob = __shfl_xor(best, off);
            const int oa = __shfl_xor(arg, off);
            if (ob > best || (ob == best && oa < arg)) {
                best = ob;
                arg = oa;
            }
        }
        if (best >= iou_thres && !claimed[arg]) {  // wave-uniform
            if (lane == 0) {
                claimed[arg] = 1;
                tpb[i] = 1.f;
            }
            ++n_claimed;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace ay

extern "C" int ay_match_detections(const float* rows, const int32_t* count, int batch, int max_det, const float* targets, int n_targets,
                                   float iou_thres, float* tp, int32_t* overflow, ay_stream_t stream) {
    using namespace ay;
    AY_CHECK_ARG(rows && count && tp && overflow && batch > 0 && max_det > 0, "ay_match_detections: bad args");
    AY_CHECK_ARG(n_targets == 0 || targets, "ay_match_detections: targets null");
    hipStream_t st = S(stream);
    if (hipMemsetAsync(overflow, 0, sizeof(int32_t), st) != hipSuccess) {
        set_error("ay_match_detections: memset failed");
        return AY_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(match_detections_kernel, dim3(batch), dim3(64), 0, st, rows, count, max_det, targets, n_targets, iou_thres, tp,
                       overflow);
    AY_CHECK_LAUNCH("match_detections_kernel");
    return AY_OK;
}
