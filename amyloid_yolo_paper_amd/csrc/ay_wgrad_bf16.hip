// Weight gradient of a convolution on the MFMA path:
//     dW[co][ci][kh][kw] = sum_{b,oy,ox} dz[b][co][oy][ox] * x[b][ci][oy*s - pad + kh][ox*s - pad + kw]
// (autograd of nn.Conv2d, models.py:33-40).  The contraction runs over PIXELS, while both tensors are stored
// [plane][pixel][16 channels]; v_mfma_f32_16x16x32_bf16 wants, per lane, 8 consecutive k (= pixels) of one row (= channel).
// ds_read_b64_tr_b16 does exactly that transpose on the way out of LDS (4 pixel rows x 16 channels per 16 lanes), so the
// tiles are staged by plain lane-linear LDS-DMA and never re-laid-out (scripts/micro/tr_mfma_test.hip pins the mapping).
//
// Workgroup (8 waves): 8 co planes (128 channels) x 4 ci planes (64 channels) x all taps; wave (cw, iw) owns 4 co planes x
// 1 ci plane x taps = 36 accumulator tiles of 16x16 (3x3).  K step = one 32-pixel segment of one output row: dz tile
// 8 x 32 px, x tile 4 planes x 3 rows x 34 px (65 for stride 2).  Segments are dealt round-robin over a split-K grid
// dimension and the partial filters are added to dW with fp32 atomics (dW is zeroed first).
// LDS slots are XOR-swizzled (slot = px ^ ((px>>3 & 1) << 2)), through the DMA source address, so that the two 4-row blocks
// a 32-lane half reads land on different banks.
#include <stdlib.h>

#include "ay_common.h"

namespace ay {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
    const uint8_t* x;
    const uint8_t* dz;
    float* dw;
    int B, cin, cout, CIP, COP, hin, win, ho, wo;
    int nseg_x, total_segs;
    float* slab;       // split-K partial filters [gridDim.z][dw_elems] (plain stores, summed in fixed order by wgrad_reduce_kernel);
    size_t dw_elems;   // nullptr: the partial sums are added to dw with fp32 atomics
};

__device__ __forceinline__ int swz(int px) { return px ^ (((px >> 3) & 1) << 2); }

// SEGS: row segments per K step (and per barrier).  1x1 layers have 4 MFMAs per wave and segment: one segment per step left the
// kernel barrier-bound (173 us per launch on average at B=32 / 1024^2 against ~70 us of HBM time); they take 4 segments per step
// from a ring of 3 stages.
template <int KS, int STRIDE, int SEGS = 1, int NBUF = 4>
__global__ void __launch_bounds__(512, 2) wgrad_bf16_kernel(WgradArgs a) {
    constexpr int PAD = (KS - 1) / 2;
    constexpr int KK2 = KS * KS;
    constexpr int CO_PL = 8, CI_PL = 4, COW = 4;       // planes per workgroup; co planes per wave
    constexpr int XW = 31 * STRIDE + KS;                // input pixels per row of a segment
    constexpr int DZ_UNITS = CO_PL * 32 * 2;            // 16-byte units
    constexpr int X_UNITS = CI_PL * KS * XW * 2;
    constexpr int DZ_PIECES = DZ_UNITS / 64;
    constexpr int X_PIECES = (X_UNITS + 63) / 64;
    constexpr int NPIECE = DZ_PIECES + X_PIECES;
    constexpr int PW = (NPIECE + 7) / 8;
    constexpr int X_BASE = DZ_PIECES * 1024;
    constexpr int BUF_BYTES = NPIECE * 1024;            // one segment
    constexpr int DUMMY = NBUF * SEGS * BUF_BYTES;
    constexpr int PWS = PW * SEGS;                      // DMA pieces per wave and stage
    static_assert(NBUF * SEGS * BUF_BYTES + 1024 <= 160 * 1024, "LDS");
    static_assert(8 * 16 * 16 * KK2 * 4 <= NBUF * SEGS * BUF_BYTES, "epilogue staging reuses the stage buffers");
    static_assert(NBUF == 3 || NBUF == 4, "ring depth");

    __shared__ __attribute__((aligned(16))) uint8_t lds[NBUF * SEGS * BUF_BYTES + 1024];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave / CI_PL, iw = wave % CI_PL;
    const int cob = blockIdx.x, cib = blockIdx.y;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const unsigned lds_base = lds_addr_of(lds);

    // segment groups (SEGS consecutive segments) of this workgroup: blockIdx.z, + gridDim.z, ...
    const int seg = blockIdx.z;
    const int total_groups = (a.total_segs + SEGS - 1) / SEGS;
    const int nstep = (total_groups - seg + (int)gridDim.z - 1) / (int)gridDim.z;
    if (nstep <= 0) return;

    // ---- loader.  Which unit of a tile a lane carries in piece i (plane, row / tap row, pixel slot, channel half) does not depend
    // on the segment: the per-lane part of every address is computed ONCE, a K step adds scalars.  Pieces go through buffer
    // descriptors (dma16_buf): descriptor base = the segment's origin in image b (the top-left corner of its halo: 64-bit scalar
    // arithmetic, may lie before the tensor for the first row -- those lanes are masked), vector offset = the lane's constant
    // (never negative), or 0x80000000 (out of range: reads as zero) for lanes outside the image / beyond the planes.
    // Before, every K step redid the unit decomposition (divisions by the tile width, 64-bit addresses, a branch) per lane and
    // piece: 61 us of a 243-us launch (128->256 3x3 at 128^2, B=16).
    constexpr unsigned OOB = 0x80000000u;
    unsigned lane_off[PW];         // byte offset of the lane's unit from the segment's origin
    int lane_dy[PW], lane_dx[PW];  // x pieces: input row / column of the unit relative to the segment's output origin (may be negative)
    const unsigned x_plane = (unsigned)a.hin * a.win * 32u, dz_plane = (unsigned)a.ho * a.wo * 32u;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int qn = i * 8 + wave;
        lane_off[i] = OOB;
        lane_dy[i] = lane_dx[i] = 0;
        if (qn < DZ_PIECES) {
            const int u = qn * 64 + lane;
            const int half = u & 1, slot = (u >> 1) & 31, pl = u >> 6;
            const int cpl = cob * CO_PL + pl;
            lane_dx[i] = swz(slot);
            if (cpl < a.COP) lane_off[i] = (unsigned)cpl * dz_plane + (unsigned)swz(slot) * 32u + half * 16u;
        } else if (qn < NPIECE) {
            const int u = (qn - DZ_PIECES) * 64 + lane;
            if (u < X_UNITS) {
                const int half = u & 1;
                const int t = u >> 1;
                const int slot = t % XW, sgi = t / XW;  // sgi = ipl * KS + kh
                const int kh = sgi % KS, ipl = sgi / KS;
                const int cpl = cib * CI_PL + ipl;
                lane_dy[i] = kh - PAD;
                lane_dx[i] = swz(slot) - PAD;
                if (cpl < a.CIP) lane_off[i] = (unsigned)cpl * x_plane + (unsigned)((kh * a.win + swz(slot)) * 32) + half * 16u;
            }
        }
    }
    auto issue = [&](int sg, int buf) __attribute__((always_inline)) {
        const bool seg_ok = sg < a.total_segs;  // the tail of the last group: every lane out of range, the buffer reads as zeros
        if (!seg_ok) sg = 0;
        const int xs = sg % a.nseg_x;
        const int oy = (sg / a.nseg_x) % a.ho;
        const int b = sg / (a.nseg_x * a.ho);
        const int ox0 = xs * 32;
        const long long org_dz = ((long long)oy * a.wo + ox0) * 32;
        const long long org_x = ((long long)(oy * STRIDE - PAD) * a.win + (ox0 * STRIDE - PAD)) * 32;   // negative in the first row
        const __amdgpu_buffer_rsrc_t rdz = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.dz) + (long long)b * a.COP * dz_plane + org_dz, 0,
                                                                          0x7ffffffc, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.x) + (long long)b * a.CIP * x_plane + org_x, 0,
                                                                         0x7ffffffc, 0x00020000);
        const unsigned so_dz = 0u, so_x = 0u;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int qn = i * 8 + wave;  // wave-uniform piece id
            if (qn < DZ_PIECES) {
                const unsigned vo = (seg_ok && ox0 + lane_dx[i] < a.wo) ? lane_off[i] : OOB;
                dma16_buf(rdz, vo, so_dz, lds_base + buf * BUF_BYTES + qn * 1024);
            } else if (qn < NPIECE) {
                const int iy = oy * STRIDE + lane_dy[i], ix = ox0 * STRIDE + lane_dx[i];
                const unsigned vo = (seg_ok && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win) ? lane_off[i] : OOB;
                dma16_buf(rx, vo, so_x, lds_base + buf * BUF_BYTES + qn * 1024);
            } else {
                dma16_buf(rx, OOB, 0u, lds_base + DUMMY);   // keeps the per-wave piece count constant (counted vmcnt waits)
            }
        }
    };

    // fragment addresses (buffer-relative)
    int aoff[2], boff[KS][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r = 8 * g + 4 * s + q;
        aoff[s] = ((cw * COW) * 32 + swz(r)) * 32 + p * 8;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) boff[kw][s] = X_BASE + ((iw * KS) * XW + swz(r * STRIDE + kw)) * 32 + p * 8;
    }

    f32x4 acc[COW][KK2];
#pragma unroll
    for (int j = 0; j < COW; ++j)
#pragma unroll
        for (int t = 0; t < KK2; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue_stage = [&](int step, int slot) __attribute__((always_inline)) {
        const int group = seg + step * (int)gridDim.z;
#pragma unroll
        for (int j = 0; j < SEGS; ++j) issue(group * SEGS + j, slot * SEGS + j);
    };
    auto wait_landed = [&](int ahead) __attribute__((always_inline)) {  // `ahead` stages issued beyond the one that must have landed
        if (NBUF == 4 && ahead >= 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PWS) : "memory");
        else if (ahead >= 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PWS) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // prologue: up to NBUF-1 stages in flight
    int issued = 0;
    for (; issued < NBUF - 1 && issued < nstep; ++issued) issue_stage(issued, issued);
    wait_landed(issued - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int cur = 0, nxt = (NBUF - 1) % NBUF;
    for (int k = 0; k < nstep; ++k) {
        if (issued < nstep) {
            issue_stage(issued, nxt);
            ++issued;
        }
        if (++nxt == NBUF) nxt = 0;
#pragma unroll
        for (int sj = 0; sj < SEGS; ++sj) {
            const uint8_t* L = lds + (cur * SEGS + sj) * BUF_BYTES;
            bf16x8 af[COW];
#pragma unroll
            for (int j = 0; j < COW; ++j) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(L + aoff[0] + j * 1024));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(L + aoff[1] + j * 1024));
                const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                __builtin_memcpy(&af[j], v, 16);
            }
#pragma unroll
            for (int t = 0; t < KK2; ++t) {
                const int kh = t / KS, kw = t % KS;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(L + boff[kw][0] + kh * XW * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(L + boff[kw][1] + kh * XW * 32));
                const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bf16x8 bfr;
                __builtin_memcpy(&bfr, v, 16);
#pragma unroll
                for (int j = 0; j < COW; ++j) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], bfr, acc[j][t], 0, 0, 0);
            }
        }
        if (k + 1 < nstep) {
            wait_landed(issued - (k + 1) - 1);  // stages issued beyond k+1 may stay in flight
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (++cur == NBUF) cur = 0;
    }

    // ---- epilogue: the workgroup's partial filters go to its split-K slab (plain contiguous stores) or, without a workspace, to
    // dW by fp32 atomics; either way coalesced.  D of one 16x16 tile: row (co) = 4*(lane>>4) + reg, col (ci) = lane & 15.
    // Adding straight from the accumulators would scatter every wave-instruction over 64 cache lines (measured: 17x
    // slower than contiguous atomics, and it was 90 % of this kernel).  Instead each wave transposes one co plane at a
    // time through its own 9 KiB of LDS into dW order [co][ci 16][tap] (144 contiguous floats per co row of this ci
    // plane) and adds 64 consecutive floats per instruction.
    __builtin_amdgcn_s_barrier();  // every wave is done with the stage buffers
    asm volatile("" ::: "memory");
    float* stage = reinterpret_cast<float*>(lds) + wave * (16 * 16 * KK2);
    const int ci0 = (cib * CI_PL + iw) * 16;
    constexpr int ROW = 16 * KK2;  // floats per co row of one ci plane
#pragma unroll
    for (int j = 0; j < COW; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < KK2; ++t) stage[((4 * g + r) * 16 + (lane & 15)) * KK2 + t] = acc[j][t][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int co_base = (cob * CO_PL + cw * COW + j) * 16;
        for (int idx = lane; idx < 16 * ROW; idx += 64) {
            const int col = idx / ROW, rem = idx % ROW;  // rem = ci_local * KK2 + tap
            const int co = co_base + col, ci = ci0 + rem / KK2;
            if (co < a.cout && ci < a.cin) {
                const size_t off = ((size_t)co * a.cin + ci0) * KK2 + rem;
                if (a.slab)
                    a.slab[(size_t)blockIdx.z * a.dw_elems + off] = stage[idx];
                else
                    atomicAdd(a.dw + off, stage[idx]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// dw[i] = (accumulate ? dw[i] : 0) + sum_s slab[s][i], s in ascending order: the same bits on every run.  Replaces ks x |dW| fp32
// atomics (~1.3 TB/s chip-wide, 58 us of a 349-us launch at 128->256 3x3, B=32, 128^2) by plain stores + one streaming pass.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, size_t n, int ks, int accumulate) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        float4 acc = accumulate ? *reinterpret_cast<const float4*>(dw + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < ks; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)s * n + i);
            acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
        }
        *reinterpret_cast<float4*>(dw + i) = acc;
    } else {
        for (size_t j = i; j < n; ++j) {
            float acc = accumulate ? dw[j] : 0.f;
            for (int s = 0; s < ks; ++s) acc += slab[(size_t)s * n + j];
            dw[j] = acc;
        }
    }
}

constexpr int WGRAD_SEGS_1X1 = 4;
#ifndef AY_WGRAD_SEGS_3X3
#define AY_WGRAD_SEGS_3X3 2
#endif
constexpr int WGRAD_SEGS_3X3 = AY_WGRAD_SEGS_3X3;   // 3x3 stride 1: 2 segments per K step from a ring of 3 stages (126 KiB)

static long long wgrad_split(const ay_conv_desc* d, int* cob, int* cib, long long* total) {
    const int CIP = (d->cin + 15) / 16;
    const int COP = (d->cout_pad > 0 ? d->cout_pad : (d->cout + 15) / 16 * 16) / 16;
    *cob = (COP + 7) / 8;
    *cib = (CIP + 3) / 4;
    *total = (long long)d->batch * d->hout * ((d->wout + 31) / 32);
    if (d->ksize == 1) *total = (*total + WGRAD_SEGS_1X1 - 1) / WGRAD_SEGS_1X1;   // K steps = groups of segments
    if (d->ksize == 3 && d->stride == 1) *total = (*total + WGRAD_SEGS_3X3 - 1) / WGRAD_SEGS_3X3;
    static const int wg_target = getenv("AY_WGRAD_WGS") ? atoi(getenv("AY_WGRAD_WGS")) : 256;
    long long ks = (wg_target + (long long)*cob * *cib - 1) / ((long long)*cob * *cib);   // ~1 workgroup per CU overall ...
    if (ks > *total / 24) ks = *total / 24;                                            // ... but >= 24 K steps each (pipeline fill, epilogue)
    if (ks > *total) ks = *total;
    if (ks > 65535) ks = 65535;
    if (ks < 1) ks = 1;
    return ks;
}

}  // namespace ay

extern "C" size_t ay_conv_wgrad_workspace_bytes(const ay_conv_desc* d) {
    if (!d) return 0;
    int cob, cib;
    long long total;
    const long long ks = ay::wgrad_split(d, &cob, &cib, &total);
    return (size_t)ks * d->cout * d->cin * d->ksize * d->ksize * sizeof(float);
}

extern "C" int ay_conv_wgrad_bf16(const ay_conv_desc* d, const void* x_blocked, const void* dz_blocked, float* dw_oihw, ay_stream_t stream) {
    return ay_conv_wgrad_bf16_acc(d, x_blocked, dz_blocked, dw_oihw, 0, stream);
}

extern "C" int ay_conv_wgrad_bf16_acc(const ay_conv_desc* d, const void* x_blocked, const void* dz_blocked, float* dw_oihw, int accumulate,
                                      ay_stream_t stream) {
    return ay_conv_wgrad_bf16_ws(d, x_blocked, dz_blocked, dw_oihw, accumulate, nullptr, 0, stream);
}

extern "C" int ay_conv_wgrad_bf16_ws(const ay_conv_desc* d, const void* x_blocked, const void* dz_blocked, float* dw_oihw, int accumulate,
                                     void* workspace, size_t workspace_bytes, ay_stream_t stream) {
    using namespace ay;
    AY_CHECK_ARG(d && x_blocked && dz_blocked && dw_oihw, "ay_conv_wgrad_bf16: null");
    // cin need not be a multiple of 16: x is read as ceil(cin/16) planes (pad channels must be zero), dW rows ci >= cin are skipped
    AY_CHECK_ARG((d->ksize == 3 && (d->stride == 1 || d->stride == 2)) || (d->ksize == 1 && d->stride == 1), "ay_conv_wgrad_bf16: shape");
    hipStream_t st = S(stream);
    WgradArgs a;
    a.x = (const uint8_t*)x_blocked;
    a.dz = (const uint8_t*)dz_blocked;
    a.dw = dw_oihw;
    a.B = d->batch;
    a.cin = d->cin;
    a.cout = d->cout;
    a.CIP = (d->cin + 15) / 16;
    a.COP = (d->cout_pad > 0 ? d->cout_pad : (d->cout + 15) / 16 * 16) / 16;
    a.hin = d->hin;
    a.win = d->win;
    a.ho = d->hout;
    a.wo = d->wout;
    a.nseg_x = (d->wout + 31) / 32;
    const long long total = (long long)d->batch * d->hout * a.nseg_x;
    AY_CHECK_ARG(total > 0 && total < 0x7fffffffLL, "ay_conv_wgrad_bf16: too many segments");
    a.total_segs = (int)total;
    int cob, cib;
    long long total2;
    const long long ks = wgrad_split(d, &cob, &cib, &total2);
    const size_t dw_elems = (size_t)d->cout * d->cin * d->ksize * d->ksize;
    const bool slabs = workspace != nullptr && workspace_bytes >= (size_t)ks * dw_elems * sizeof(float);
    a.slab = slabs ? (float*)workspace : nullptr;
    a.dw_elems = dw_elems;
    // without a workspace the kernel ADDS its split-K partial sums to dw (fp32 atomics): accumulate = 0 starts from zero
    if (!slabs && !accumulate && hipMemsetAsync(dw_oihw, 0, sizeof(float) * dw_elems, st) != hipSuccess) {
        set_error("ay_conv_wgrad_bf16: memset failed");
        return AY_ERR_LAUNCH;
    }
    dim3 grid(cob, cib, (unsigned)ks), block(512);
    if (d->ksize == 3 && d->stride == 1)
        hipLaunchKernelGGL((wgrad_bf16_kernel<3, 1, WGRAD_SEGS_3X3, WGRAD_SEGS_3X3 == 1 ? 4 : 3>), grid, block, 0, st, a);
    else if (d->ksize == 3)
        hipLaunchKernelGGL((wgrad_bf16_kernel<3, 2>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((wgrad_bf16_kernel<1, 1, WGRAD_SEGS_1X1, 3>), grid, block, 0, st, a);
    AY_CHECK_LAUNCH("wgrad_bf16_kernel");
    if (slabs) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((dw_elems + 1023) / 1024)), dim3(256), 0, st, a.slab, dw_oihw, dw_elems, (int)ks,
                           accumulate);
        AY_CHECK_LAUNCH("wgrad_reduce_kernel");
    }
    return AY_OK;
}
