// Tile ingest on the device: uint8 HWC RGB tiles -> the float32 CHW tensor the network takes, with the
// reference's preprocessing fused into one pass:
//   ToTensor            x / 255                                      (utils/transforms.py:96, torchvision ToTensor)
//   pad_to_square       centre zero padding, short side             (utils/datasets.py:22-32; PadSquare transforms.py:83-90)
//   resize              F.interpolate(mode="nearest"): src = min(floor(dst * (float)in / out), in - 1)   (utils/datasets.py:35-37)
// so the host uploads 3 bytes per pixel instead of 12 and never touches the pixels again.
#include "ay_common.h"

namespace ay {

__global__ void __launch_bounds__(256) ingest_u8_kernel(const uint8_t* __restrict__ img, int B, int H, int W, int S, float pad_value,
                                                         float* __restrict__ out) {
    const int D = H > W ? H : W;               // side of the padded square
    const int top = H <= W ? (W - H) / 2 : 0;  // pad1 = diff // 2 goes first (top / left)
    const int left = H > W ? (H - W) / 2 : 0;
    const float scale = (float)D / (float)S;   // ATen's nearest scale for size= (no scale_factor): in / out in fp32
    const size_t plane = (size_t)S * S;
    const size_t total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % S), y = (int)((i / S) % S);
        const size_t b = i / plane;
        const int sy = min((int)floorf(y * scale), D - 1) - top;
        const int sx = min((int)floorf(x * scale), D - 1) - left;
        float r = pad_value, g = pad_value, bl = pad_value;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
            const uint8_t* p = img + ((b * H + sy) * (size_t)W + sx) * 3;
            r = (float)p[0] / 255.0f;
            g = (float)p[1] / 255.0f;
            bl = (float)p[2] / 255.0f;
        }
        float* o = out + b * 3 * plane + (size_t)y * S + x;
        o[0] = r;
        o[plane] = g;
        o[2 * plane] = bl;
    }
}

// WSI -> tile streaming (crop.py:13-25,44-47): the tiles dzsave(layout='google', tile_size=1536) cuts out of a
// slide -- edge tiles padded to the full size with the background 255 -- taken straight out of a resident uint8 HWC region
// (a full-width strip of the slide: one contiguous upload), optionally after the 40x -> 20x halving, then the N1 chain
// (/255, nearest resize to the network size).  The halving is a 2x2 mean with round-half-up in uint8: pyvips' resize(0.5) is a
// lanczos3 reduce and the reference then goes through JPEG Q=90, neither is restated (parity unpinned).
__global__ void __launch_bounds__(256) region_tiles_u8_kernel(const uint8_t* __restrict__ reg, int RH, int RW, size_t stride, int shrink,
                                                               int tile, int tiles_y, int tiles_x, int S, float* __restrict__ out) {
    const int H = RH / shrink, W = RW / shrink;  // the (halved) image the tile grid lies on
    const float scale = (float)tile / (float)S;
    const size_t plane = (size_t)S * S;
    const size_t total = (size_t)tiles_y * tiles_x * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % S), y = (int)((i / S) % S);
        const size_t t = i / plane;
        const int ty = (int)(t / tiles_x), tx = (int)(t % tiles_x);
        const int Y = ty * tile + min((int)floorf(y * scale), tile - 1);
        const int X = tx * tile + min((int)floorf(x * scale), tile - 1);
        float v[3] = {1.0f, 1.0f, 1.0f};  // background 255
        if (Y < H && X < W) {
            if (shrink == 1) {
                const uint8_t* p = reg + (size_t)Y * stride + (size_t)X * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = (float)p[c] / 255.0f;
            } else {
                const uint8_t* p0 = reg + (size_t)(2 * Y) * stride + (size_t)(2 * X) * 3;
                const uint8_t* p1 = p0 + stride;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = (float)((p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2) / 255.0f;
            }
        }
        float* o = out + t * 3 * plane + (size_t)y * S + x;
        o[0] = v[0];
        o[plane] = v[1];
        o[2 * plane] = v[2];
    }
}

}  // namespace ay

extern "C" int ay_ingest_region_tiles_u8(const void* region_hwc_u8, int region_h, int region_w, size_t row_stride_bytes, int shrink,
                                         int tile, int tiles_y, int tiles_x, int out_size, float* out_nchw, ay_stream_t stream) {
    using namespace ay;
    AY_CHECK_ARG(region_hwc_u8 && out_nchw, "ay_ingest_region_tiles_u8: null");
    AY_CHECK_ARG(region_h > 0 && region_w > 0 && row_stride_bytes >= (size_t)region_w * 3 && (shrink == 1 || shrink == 2),
                 "ay_ingest_region_tiles_u8: region %dx%d stride %zu shrink %d", region_h, region_w, row_stride_bytes, shrink);
    AY_CHECK_ARG(tile > 0 && tiles_y > 0 && tiles_x > 0 && out_size > 0, "ay_ingest_region_tiles_u8: tile grid %dx%d of %d -> %d",
                 tiles_y, tiles_x, tile, out_size);
    const size_t total = (size_t)tiles_y * tiles_x * out_size * out_size;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(region_tiles_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, S(stream), (const uint8_t*)region_hwc_u8, region_h,
                       region_w, row_stride_bytes, shrink, tile, tiles_y, tiles_x, out_size, out_nchw);
    AY_CHECK_LAUNCH("region_tiles_u8_kernel");
    return AY_OK;
}

extern "C" int ay_ingest_tiles_u8(const void* img_hwc_u8, int batch, int h, int w, int out_size, float pad_value, float* out_nchw,
                                  ay_stream_t stream) {
    using namespace ay;
    AY_CHECK_ARG(img_hwc_u8 && out_nchw, "ay_ingest_tiles_u8: null");
    AY_CHECK_ARG(batch > 0 && h > 0 && w > 0 && out_size > 0, "ay_ingest_tiles_u8: bad shape %dx%dx%d -> %d", batch, h, w, out_size);
    const size_t total = (size_t)batch * out_size * out_size;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(ingest_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, S(stream), (const uint8_t*)img_hwc_u8, batch, h, w, out_size,
                       pad_value, out_nchw);
    AY_CHECK_LAUNCH("ingest_u8_kernel");
    return AY_OK;
}
