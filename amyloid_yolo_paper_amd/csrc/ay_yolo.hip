// YOLO head decode (models.py:127-172) and box math (utils/utils.py:53-59,193-232) in fp32.
// Arithmetic follows the reference's operation order; the library is built with -ffp-contract=off so
// no multiply-add is fused behind the reference's back (threshold comparisons must land on the same side).
#include "ay_common.h"

namespace ay {

struct Anchors {
    float w[16], h[16];
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// one thread = one (b, a, gy, gx) cell; row = a*G*G + gy*G + gx (models.py:163-170)
__global__ void yolo_decode_kernel(const float* __restrict__ head, int layout, float* __restrict__ out, int B, int A, int C, int G,
                                   float stride, Anchors an, int n_total, int row_offset) {
    const int cells = A * G * G;
    const size_t total = (size_t)B * cells;
    const int K = 5 + C;
    const int cpl = ((A * K + 31) / 32) * 2;  // blocked heads are padded to 32 channels (conv cout_pad)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cell = (int)(i % cells);
        const int b = (int)(i / cells);
        const int gx = cell % G, gy = (cell / G) % G, a = cell / (G * G);
        float* o = out + ((size_t)b * n_total + row_offset + cell) * K;
        auto ld = [&](int k) -> float {
            const int ch = a * K + k;
            if (layout == 1) return head[((((size_t)b * cpl + (ch >> 4)) * G + gy) * G + gx) * 16 + (ch & 15)];
            return head[(((size_t)b * A * K + ch) * G + gy) * G + gx];
        };
        // (sigmoid + grid) and (exp * anchor/stride) in grid units, then * stride, as the reference does
        const float aw = an.w[a] / stride, ah = an.h[a] / stride;
        o[0] = (sigmoidf_(ld(0)) + (float)gx) * stride;
        o[1] = (sigmoidf_(ld(1)) + (float)gy) * stride;
        o[2] = (expf(ld(2)) * aw) * stride;
        o[3] = (expf(ld(3)) * ah) * stride;
        for (int k = 4; k < K; ++k) o[k] = sigmoidf_(ld(k));
    }
}

__global__ void xywh2xyxy_kernel(float* boxes, int64_t n, int stride_) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float* p = boxes + i * stride_;
        const float cx = p[0], cy = p[1], hw = p[2] / 2.0f, hh = p[3] / 2.0f;
        p[0] = cx - hw;
        p[1] = cy - hh;
        p[2] = cx + hw;
        p[3] = cy + hh;
    }
}

__device__ __forceinline__ float iou_plus1(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2,
                                           float by2) {
    const float ix1 = fmaxf(ax1, bx1), iy1 = fmaxf(ay1, by1);
    const float ix2 = fminf(ax2, bx2), iy2 = fminf(ay2, by2);
    const float inter = fmaxf(ix2 - ix1 + 1.0f, 0.0f) * fmaxf(iy2 - iy1 + 1.0f, 0.0f);
    const float a1 = (ax2 - ax1 + 1.0f) * (ay2 - ay1 + 1.0f);
    const float a2 = (bx2 - bx1 + 1.0f) * (by2 - by1 + 1.0f);
    return inter / (a1 + a2 - inter + 1e-16f);
}

// GIoU on corner boxes, no +1 rule.  New feature: the reference has none -> parity unpinned.
__device__ __forceinline__ float giou(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2, float by2) {
    const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.0f);
    const float ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.0f);
    const float inter = iw * ih;
    const float a1 = (ax2 - ax1) * (ay2 - ay1), a2 = (bx2 - bx1) * (by2 - by1);
    const float uni = a1 + a2 - inter + 1e-16f;
    const float cw = fmaxf(ax2, bx2) - fminf(ax1, bx1), ch = fmaxf(ay2, by2) - fminf(ay1, by1);
    const float hull = cw * ch + 1e-16f;
    return inter / uni - (hull - uni) / hull;
}

__device__ __forceinline__ void load_box(const float* p, int xyxy, float& x1, float& y1, float& x2, float& y2) {
    if (xyxy) {
        x1 = p[0];
        y1 = p[1];
        x2 = p[2];
        y2 = p[3];
    } else {  // utils/utils.py:206-211
        x1 = p[0] - p[2] / 2.0f;
        x2 = p[0] + p[2] / 2.0f;
        y1 = p[1] - p[3] / 2.0f;
        y2 = p[1] + p[3] / 2.0f;
    }
}

__global__ void box_iou_kernel(const float* b1, int n1, const float* b2, int n2, int xyxy, int mode, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    float ax1, ay1, ax2, ay2, bx1, by1, bx2, by2;
    load_box(b1 + (n1 == 1 ? 0 : (size_t)i * 4), xyxy, ax1, ay1, ax2, ay2);
    load_box(b2 + (size_t)i * 4, xyxy, bx1, by1, bx2, by2);
    out[i] = mode == 0 ? iou_plus1(ax1, ay1, ax2, ay2, bx1, by1, bx2, by2) : giou(ax1, ay1, ax2, ay2, bx1, by1, bx2, by2);
}

__global__ void box_iou_pairwise_kernel(const float* b1, int n1, const float* b2, int n2, int mode, float* out) {
    const size_t total = (size_t)n1 * n2;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(t % n2);
        const int i = (int)(t / n2);
        const float* p = b1 + (size_t)i * 4;
        const float* q = b2 + (size_t)j * 4;
        out[t] = mode == 0 ? iou_plus1(p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3])
                           : giou(p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3]);
    }
}

}  // namespace ay

using namespace ay;

extern "C" int ay_yolo_decode(const float* head, int layout, float* out, int batch, int num_anchors, int num_classes, int grid,
                              int img_dim, const float* anchors_wh, int n_total, int row_offset, ay_stream_t stream) {
    AY_CHECK_ARG(head && out && anchors_wh, "ay_yolo_decode: null");
    AY_CHECK_ARG(num_anchors > 0 && num_anchors <= 16 && grid > 0 && (layout == 0 || layout == 1), "ay_yolo_decode: bad shape");
    AY_CHECK_ARG(row_offset >= 0 && row_offset + num_anchors * grid * grid <= n_total, "ay_yolo_decode: rows out of range");
    Anchors an;
    for (int a = 0; a < num_anchors; ++a) {
        an.w[a] = anchors_wh[2 * a];
        an.h[a] = anchors_wh[2 * a + 1];
    }
    const size_t total = (size_t)batch * num_anchors * grid * grid;
    unsigned g = (unsigned)((total + 255) / 256);
    if (g > 16384) g = 16384;
    // stride = img_dim / G in Python float division, then cast (models.py:119)
    const float stride = (float)((double)img_dim / (double)grid);
    hipLaunchKernelGGL(yolo_decode_kernel, dim3(g), dim3(256), 0, S(stream), head, layout, out, batch, num_anchors, num_classes,
                       grid, stride, an, n_total, row_offset);
    AY_CHECK_LAUNCH("yolo_decode_kernel");
    return AY_OK;
}

extern "C" int ay_xywh2xyxy(float* boxes, int64_t n_rows, int row_stride, ay_stream_t stream) {
    AY_CHECK_ARG(boxes && row_stride >= 4, "ay_xywh2xyxy: bad args");
    if (n_rows <= 0) return AY_OK;
    unsigned g = (unsigned)((n_rows + 255) / 256);
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(xywh2xyxy_kernel, dim3(g), dim3(256), 0, S(stream), boxes, n_rows, row_stride);
    AY_CHECK_LAUNCH("xywh2xyxy_kernel");
    return AY_OK;
}

extern "C" int ay_box_iou(const float* box1, int n1, const float* box2, int n2, int xyxy, int mode, float* out,
                          ay_stream_t stream) {
    AY_CHECK_ARG(box1 && box2 && out && (n1 == n2 || n1 == 1) && (mode == 0 || mode == 1), "ay_box_iou: bad args");
    if (n2 <= 0) return AY_OK;
    hipLaunchKernelGGL(box_iou_kernel, dim3((n2 + 255) / 256), dim3(256), 0, S(stream), box1, n1, box2, n2, xyxy, mode, out);
    AY_CHECK_LAUNCH("box_iou_kernel");
    return AY_OK;
}

extern "C" int ay_box_iou_pairwise(const float* box1, int n1, const float* box2, int n2, int mode, float* out,
                                   ay_stream_t stream) {
    AY_CHECK_ARG(box1 && box2 && out && (mode == 0 || mode == 1), "ay_box_iou_pairwise: bad args");
    if (n1 <= 0 || n2 <= 0) return AY_OK;
    const size_t total = (size_t)n1 * n2;
    unsigned g = (unsigned)((total + 255) / 256);
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(box_iou_pairwise_kernel, dim3(g), dim3(256), 0, S(stream), box1, n1, box2, n2, mode, out);
    AY_CHECK_LAUNCH("box_iou_pairwise_kernel");
    return AY_OK;
}
