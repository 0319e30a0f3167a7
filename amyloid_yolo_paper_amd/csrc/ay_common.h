// Shared helpers for the gfx950 kernels of libamyloid_yolo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/amyloid_yolo.h"

namespace ay {

void set_error(const char* fmt, ...);

#define AY_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            ay::set_error(__VA_ARGS__);  \
            return AY_ERR_ARG;           \
        }                                \
    } while (0)

#define AY_CHECK_LAUNCH(what)                                                      \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            ay::set_error("%s: %s", what, hipGetErrorString(e_));                  \
            return AY_ERR_LAUNCH;                                                  \
        }                                                                          \
    } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// round-to-nearest-even f32 -> bf16 bits (plain cast: keeps NaN a NaN, v_cvt_pk_bf16_f32 at -O3)
__device__ __forceinline__ uint16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf2f(uint16_t u) {
    return __builtin_bit_cast(float, (uint32_t)u << 16);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// two floats -> one dword of bf16 (round to nearest even) in a single v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack2bf2(f32x2 v) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// one dword of bf16 -> two floats
__device__ __forceinline__ f32x2 bf2f2(uint32_t u) {
    return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}
// LeakyReLU on two values: max(x, slope * x) equals x > 0 ? x : slope * x for 0 < slope <= 1 (slope 1: identity)
__device__ __forceinline__ f32x2 leaky2(f32x2 x, float slope) {
    const f32x2 y = x * slope;
    return f32x2{__builtin_fmaxf(x[0], y[0]), __builtin_fmaxf(x[1], y[1])};
}

// ---- the two 16-bit storage types of the MFMA inference path -------------------------------------------------------
// The convolution kernels are templates over DT: operand / activation element type, its two matrix instructions and the
// conversions of the fused epilogue.  Same layouts, same bytes, same MFMA rate; Bf16 keeps fp32's exponent range (training
// and default), F16 (IEEE binary16, inference only) has an 11-bit significand: an 8x smaller rounding step per stored
// activation, finite up to 65504 -- stored activations are post-BatchNorm, the fp32 epilogue applies scale / shift before
// the single rounding.  v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32 both round to nearest even.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
struct Bf16 {
    typedef bf16x8 vec8;
    static constexpr int id = AY_DT_BF16;
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint32_t pack2(f32x2 v) { return pack2bf2(v); }
    static __device__ __forceinline__ f32x2 unpack2(uint32_t u) { return bf2f2(u); }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return f2bf(f); }
    static __device__ __forceinline__ float to_f32(uint16_t u) { return bf2f(u); }
    static __device__ __forceinline__ void enter() {}   // (F16::enter sets a conversion mode bit; bfloat16 shares fp32's range)
};
struct F16 {
    typedef f16x8 vec8;
    static constexpr int id = AY_DT_F16;
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint32_t pack2(f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t)); }
    static __device__ __forceinline__ f32x2 unpack2(uint32_t u) { return __builtin_convertvector(__builtin_bit_cast(f16x2_t, u), f32x2); }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
    // Called first thing by every kernel that stores halves: MODE.FP16_OVFL = 1 (hwreg 1 = MODE, bit 23): a conversion that
    // overflows the half range SATURATES at +-65504 instead of producing +-inf; true infinities and NaNs pass through.  A
    // post-BatchNorm or post-shortcut value beyond 65504 (the residual stream of Darknet-53 adds without bound across blocks; the
    // synthetic weights stay below 2 300) would otherwise become inf and turn into NaN one layer later without a diagnostic.
    // Free: a mode bit, no instruction per value.
    static __device__ __forceinline__ void enter() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }
    static __device__ __forceinline__ float to_f32(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
};
template <typename DT>
__device__ __forceinline__ uint32_t pack2_scalar(float lo, float hi) {
    return (uint32_t)DT::from_f32(lo) | ((uint32_t)DT::from_f32(hi) << 16);
}

// 16-byte LDS-DMA (global_load_lds_dwordx4): lane l copies 16 B from its own `gsrc` to LDS byte address `lds_addr + 16*l`
// (`lds_addr` wave-uniform).  Issued through inline asm so that hipcc neither counts it nor guards LDS reads against it:
// with the builtin the compiler inserted `s_waitcnt vmcnt(0)` between a DMA and the next ds_read / MFMA whenever it could
// not prove the buffers distinct (seen on the 3-deep 1x1 ring: every DMA was waited for on the spot, 6x slower).  The
// caller owns the ordering: a counted `s_waitcnt vmcnt(N)` + barrier before anyone reads the data.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_addr))
                 : "memory");
}
// The same through a buffer descriptor (buffer_load_dwordx4 ... lds): lane l copies 16 B from descriptor base + `soff` (scalar)
// + `voff` (its own 32-bit offset) to LDS byte address `lds_addr + 16*l`.  A lane whose `voff` is >= the descriptor's
// num_records (e.g. 0x80000000) is out of range and delivers zeros: border / padding lanes need no zero page and no select,
// and no lane carries a 64-bit address.  The scalar offset takes no part in the range check.
__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(soff), "s"(__builtin_amdgcn_readfirstlane(lds_addr))
                 : "memory");
}
// 4-byte variant (global_load_lds_dword): lane l copies 4 B from its own `gsrc` to LDS byte address `lds_addr + 4*l`;
// for sources that are only 4-byte aligned (fp32 image rows at arbitrary column offsets).
__device__ __forceinline__ void dma4(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_addr))
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

static inline hipStream_t S(ay_stream_t s) { return (hipStream_t)s; }

}  // namespace ay
