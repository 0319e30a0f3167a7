// The stem (models.py:26-45 for layer 0: Conv2d(3, 32, 3, 1, 1) + BN + LeakyReLU) on the bf16 TRAINING path, straight from the
// fp32 NCHW image:
//     ay_stem_train_fwd_bf16    z = bf16( conv3x3( bf16(x), bf16(w) ) ), fp32 accumulation        -> [B][2][H][W][16] bf16
//     ay_stem_train_wgrad_bf16  dW[co][ci][kh][kw] (+)= sum_{b,y,x} dz[b][co][y][x] * bf16(x)[b][ci][y+kh-1][x+kw-1]
// (the reference gets both from autograd, train.py:113).  Until round 2 the training engine turned the image into a zero-padded
// 16-channel bf16 plane (1.07 GB written and read twice per step at B=32 / 1024^2 for 3 useful channels) and ran the generic
// kernels on it: conversion 1.26 + forward 1.0-1.5 + weight gradient 1.5 ms per step.  Both kernels here read the image once.
//
// Same tile mechanics as the fused inference stem (ay_stem_fused.hip, v2): an item is 8 x 64 stem pixels; its (8+2) x (64+2) x 3
// fp32 window arrives by 16-byte buffer-descriptor DMA into a ring of LDS buffers, stored [row][channel][column] with the window
// starting 4 columns left of the tile (every piece 16-byte aligned, out-of-image lanes read zeros through the descriptor's range
// check; needs W % 4 == 0), so the 27 taps of a pixel are a 9 x 3 grid (rho = 3 kh + ci, kw) of floats.
#include "ay_conv_common.h"

namespace ay {

namespace stemtr {
constexpr int TH = 8, TW = 64;                 // stem pixels per item
constexpr int IH = TH + 3;                      // window rows y0-1 .. y0+9: 10 used + 1 that only zero-filter slots read
constexpr int IW = 72;                          // window columns x0-4 .. x0+67 (used x0-1 .. x0+64)
constexpr int XOFF = 3;                         // window column of image column x-1 for tile column 0
constexpr int ROW_PIECES = IW / 4;
constexpr int IMG_PIECES = IH * 3 * ROW_PIECES; // 594 pieces of 16 bytes
constexpr int IMG_DMAS = (IMG_PIECES + 63) / 64;  // 10 wave-wide DMA instructions
constexpr int IMG_BYTES = IMG_DMAS * 1024;
constexpr unsigned OOB = 0x80000000u;

// K slots of the 32-wide reduction over taps (forward: the MFMA's K; weight gradient: its N): slot = step*16 + half*8 + e holds grid
// tap t = rho*3 + kw (rho = 3 kh + ci), ordered so that half 1 of a step is half 0 moved down 3 (step 0) or 2 (step 1) grid rows;
// the five spare slots of (step 1, half 1) alias real taps and carry zero filters / are dropped.
__host__ __device__ constexpr int slot_tap(int slot) {
    const int step = slot >> 4, half = (slot >> 3) & 1, e = slot & 7;
    if (step == 0) return e + 9 * half;                                   // rows 0,1,(2,0),(2,1) | rows 3,4,(5,0),(5,1)
    const int c = e < 6 ? 18 + e : (e == 6 ? 8 : 17);                     // rows 6,7,(2,2),(5,2)
    return half ? c + 6 : c;                                              // | row 8, row 9 (spare), (4,2) (spare), (7,2) (spare)
}
__host__ __device__ constexpr bool slot_live(int slot) {
    const int step = slot >> 4, half = (slot >> 3) & 1;
    return !(step == 1 && half == 1) || slot_tap(slot) / 3 == 8;
}
__host__ __device__ constexpr int tap_goff(int t) { return (t / 3) * IW + t % 3; }               // float offset of grid tap t from the pixel
__host__ __device__ constexpr int tap_kref(int t) { return ((t / 3) % 3) * 9 + (t / 9) * 3 + t % 3; }  // ci*9 + kh*3 + kw

struct Args {
    const float* x;          // [B][3][H][W]
    int B, H, W, tiles_x, tiles_y;
    unsigned m_tx, m_tpi;    // floor(2^32 / tiles_x), floor(2^32 / tiles per image); 0xffffffff for a divisor of 1
    const uint16_t* w0;      // forward: bf16 [32 cout][32], index ci*9 + kh*3 + kw (27..31 unused)
    uint8_t* z;              // forward: [B][2][H][W][16] bf16
    const uint8_t* dz;       // weight gradient: [B][2][H][W][16] bf16
    float* slab;             // weight gradient: [gridDim.x][864] partial filters in dW order
    float* stat_slab;        // forward, optional: [gridDim.x][64] per-workgroup (sum z[32], sum z^2[32]) of the ROUNDED outputs it stored
};

struct Coord {
    int b, y0, x0;
};
__device__ __forceinline__ Coord coord_of(const Args& a, unsigned it) {
    const unsigned tpi = (unsigned)(a.tiles_x * a.tiles_y);
    unsigned b = __umulhi(it, a.m_tpi), r = it - b * tpi;
    if (r >= tpi) ++b, r -= tpi;
    unsigned ty = __umulhi(r, a.m_tx), tx = r - ty * (unsigned)a.tiles_x;
    if (tx >= (unsigned)a.tiles_x) ++ty, tx -= (unsigned)a.tiles_x;
    return Coord{(int)b, (int)ty * TH, (int)tx * TW};
}
}  // namespace stemtr

// ---------------------------------------------------------------------------------------------------------------------
// forward: 8 waves, each two blocks of 32 pixels per item (v_mfma_f32_32x32x16_bf16: rows = 32 filters, columns = 32 pixels, K = 32
// tap slots in two steps); the result is rounded once and stored through the lane-pair swap of the inference epilogues.
template <bool STATS>
__global__ void __launch_bounds__(512, 4) stem_train_fwd_kernel(stemtr::Args a, int n_items) {
    using namespace stemtr;
    constexpr int NIB = 3;
    constexpr int OFF_SCRATCH = NIB * IMG_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t lds[OFF_SCRATCH + 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hh = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (n_items + 7) >> 3;
    const int first = xcd * per_xcd, last = min(first + per_xcd, n_items);
    if (first + slot >= last) {
        if (STATS && tid < 64) a.stat_slab[(size_t)blockIdx.x * 64 + tid] = 0.f;
        return;
    }
    const int nk = (last - (first + slot) + slots - 1) / slots;
    auto item_at = [&](int k) { return (unsigned)(first + slot + k * slots); };
    const unsigned lds_base = lds_addr_of(lds);
    const int plane_elems = a.H * a.W;

    // tile pieces of this wave: pieces wave and wave + 8 (the second only for waves 0, 1); per lane (row, channel, 4 columns)
    int pu_gofs[2], pu_row[2], pu_col[2];
    bool pu_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pu = (wave + 8 * j) * 64 + lane;
        pu_ok[j] = wave + 8 * j < IMG_DMAS && pu < IMG_PIECES;
        pu_col[j] = (pu % ROW_PIECES) * 4;
        const int ci = (pu / ROW_PIECES) % 3;
        pu_row[j] = pu / (ROW_PIECES * 3);
        pu_gofs[j] = ci * plane_elems + pu_row[j] * a.W + pu_col[j];
    }
    auto issue_image = [&](const Coord& t, int buf) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)t.b * 3 * plane_elems, 0, 3 * plane_elems * 4, 0x00020000);
        const int ybase = t.y0 - 1, xbase = t.x0 - 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = pu_ok[j] && (unsigned)(ybase + pu_row[j]) < (unsigned)a.H && (unsigned)(xbase + pu_col[j]) < (unsigned)a.W;
            const unsigned vo = ok ? (unsigned)((pu_gofs[j] + ybase * a.W + xbase) * 4) : OOB;
            dma16_buf(rs, vo, 0u, lds_base + (wave + 8 * j < IMG_DMAS ? buf * IMG_BYTES + (wave + 8 * j) * 1024 : OFF_SCRATCH));
        }
    };
    // filters of output channel c in slot order
    bf16x8 wa[2];
    {
        const uint16_t* wr = a.w0 + c * 32;
        uint16_t v[2][8];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int s0 = ks * 16 + e, s1 = ks * 16 + 8 + e;
                const uint16_t w_h0 = slot_live(s0) ? wr[tap_kref(slot_tap(s0))] : (uint16_t)0;
                const uint16_t w_h1 = slot_live(s1) ? wr[tap_kref(slot_tap(s1))] : (uint16_t)0;
                v[ks][e] = hh ? w_h1 : w_h0;
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            wa[ks] = __builtin_bit_cast(bf16x8, make_uint4(v[ks][0] | (unsigned)v[ks][1] << 16, v[ks][2] | (unsigned)v[ks][3] << 16,
                                                            v[ks][4] | (unsigned)v[ks][5] << 16, v[ks][6] | (unsigned)v[ks][7] << 16));
    }
    // this wave's blocks: 2 * wave + j -> tile row wave, columns 32 j ..; per lane the window offset of its pixel
    int pbase0[2], pbase1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int px = wave * 3 * IW + 32 * j + c + XOFF;
        pbase0[j] = px + hh * 3 * IW;
        pbase1[j] = px + hh * 2 * IW;
    }
    const unsigned out_plane_bytes = (unsigned)plane_elems * 32u;
    // BatchNorm batch statistics of this layer (models.py:43 in train mode) gathered where the outputs are produced: a workgroup keeps
    // per-lane sums of its 16 channels over ALL its items (a few hundred) and reduces them once at the end -- the separate pass that
    // re-reads the 2.1 GB of z (B=32 / 1024^2) is not needed.  The sums are over the bf16-rounded values, as the separate pass sees them.
    float st1[16], st2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) st1[r] = st2[r] = 0.f;

    Coord cA = coord_of(a, item_at(0)), cB = cA;
    issue_image(cA, 0);
    if (nk > 1) {
        cB = coord_of(a, item_at(1));
        issue_image(cB, 1);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
        // tile k+2 into the buffer tile k-1 was read from (every wave passed the barrier at the end of iteration k-1 since)
        const bool issue2 = k + 2 < nk;
        Coord cC = cB;
        if (issue2) {
            cC = coord_of(a, item_at(k + 2));
            issue_image(cC, (k + 2) % NIB);
        }
        const float* img = reinterpret_cast<const float*>(lds + (k % NIB) * IMG_BYTES);
        const __amdgpu_buffer_rsrc_t orsrc =
            __builtin_amdgcn_make_buffer_rsrc(a.z + (size_t)cA.b * 2 * out_plane_bytes, 0, (int)(2 * out_plane_bytes), 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* p0 = img + pbase0[j];
            const float* p1 = img + pbase1[j];
            f32x2 v0[4], v1[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                v0[jj] = f32x2{p0[tap_goff(slot_tap(2 * jj))], p0[tap_goff(slot_tap(2 * jj + 1))]};
                v1[jj] = f32x2{p1[tap_goff(slot_tap(16 + 2 * jj))], p1[tap_goff(slot_tap(16 + 2 * jj + 1))]};
            }
            const uint4 b0 = make_uint4(pack2bf2(v0[0]), pack2bf2(v0[1]), pack2bf2(v0[2]), pack2bf2(v0[3]));
            const uint4 b1 = make_uint4(pack2bf2(v1[0]), pack2bf2(v1[1]), pack2bf2(v1[2]), pack2bf2(v1[3]));
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], __builtin_bit_cast(bf16x8, b0), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1], __builtin_bit_cast(bf16x8, b1), acc, 0, 0, 0);
            // D: column (pixel) = lane & 31, row (channel) = (reg & 3) + 8 (reg >> 2) + 4 hh; one swap per dword pair turns two quads
            // into the 16 bytes this lane stores: channels 8hh..8hh+7 of plane qp
            const int oy = cA.y0 + wave, ox = cA.x0 + 32 * j + c;
            const bool ok = oy < a.H && ox < a.W;
            const unsigned vo = ok ? ((unsigned)oy * a.W + ox) * 32u + hh * 16u : OOB;
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                const int o = qp * 8;
                const unsigned ax = pack2bf2(f32x2{acc[o + 0], acc[o + 1]}), ay_ = pack2bf2(f32x2{acc[o + 2], acc[o + 3]});
                const unsigned bx = pack2bf2(f32x2{acc[o + 4], acc[o + 5]}), by = pack2bf2(f32x2{acc[o + 6], acc[o + 7]});
                if (STATS && ok) {   // register o + i holds channel (i & 3) + 8 (i >> 2) + 4 hh + 16 qp of this lane's pixel
                    const f32x2 r01 = bf2f2(ax), r23 = bf2f2(ay_), r45 = bf2f2(bx), r67 = bf2f2(by);
                    const float rv[8] = {r01[0], r01[1], r23[0], r23[1], r45[0], r45[1], r67[0], r67[1]};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        st1[o + i] += rv[i];
                        st2[o + i] += rv[i] * rv[i];
                    }
                }
                auto r0 = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(ay_, by, false, false);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{r0[0], r1[0], r0[1], r1[1]}, orsrc, vo + (unsigned)qp * out_plane_bytes, 0, 0);
            }
        }
        // tile k+1 landed: younger operations of this wave are the 4 stores of item k-1 (if any), the 2 DMAs of tile k+2 (if issued) and
        // the 4 stores of this item -- all issued AFTER the DMAs of tile k+1
        if (k + 1 < nk) {
            if (issue2)
                asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        cA = cB;
        cB = cC;
    }
    if constexpr (STATS) {
        // lanes of one half hold the same 16 channels (register r: channel (r & 3) + 8 ((r >> 2) & 1) + 4 hh + 16 (r >> 3)) for 32
        // different pixels: butterfly over the 32 lanes, then the 8 waves through LDS, one row of 64 floats per workgroup
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                st1[r] += __shfl_xor(st1[r], off);
                st2[r] += __shfl_xor(st2[r], off);
            }
        }
        __syncthreads();   // the tile buffers are dead
        float* red = reinterpret_cast<float*>(lds);   // [wave 8][64]
        if (c == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = (r & 3) + 8 * ((r >> 2) & 1) + 4 * hh + 16 * (r >> 3);
                red[wave * 64 + ch] = st1[r];
                red[wave * 64 + 32 + ch] = st2[r];
            }
        }
        __syncthreads();
        if (tid < 64) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[w * 64 + tid];
            a.stat_slab[(size_t)blockIdx.x * 64 + tid] = v;
        }
    }
}

// sums[c] = sum z, sums[32 + c] = sum z^2 over the whole batch, fp64, workgroup rows added in a fixed order
__global__ void stem_stats_reduce_kernel(const float* __restrict__ slab, int n_rows, double* __restrict__ sums) {
    const int i = threadIdx.x;
    if (i >= 64) return;
    double s = 0.0;
    for (int k = 0; k < n_rows; ++k) s += (double)slab[(size_t)k * 64 + i];
    sums[i] = s;
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient: v_mfma_f32_16x16x32_bf16 with K = 32 pixels of one tile row, M = 16 filters (one plane of dz, transposed on the
// way out of LDS by ds_read_b64_tr_b16 as in ay_wgrad_bf16.hip), N = 16 tap slots whose operand a lane builds from 8 consecutive
// floats of the window.  Every wave sums its own chunks over all items of the workgroup; one cross-wave sum at the end, one slab of
// 864 partial filters per workgroup, added up in a fixed order by stem_wgrad_reduce_kernel (no atomics: reproducible).
__device__ __forceinline__ int stw_swz(int px) { return px ^ (((px >> 3) & 1) << 2); }

__global__ void __launch_bounds__(512) stem_train_wgrad_kernel(stemtr::Args a, int n_items) {
    using namespace stemtr;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    constexpr int NB = 2;
    constexpr int DZ_BYTES = 16 * 2 * 32 * 32;          // [chunk 16][plane 2][slot 32][32 B]
    constexpr int DZ_PIECES = DZ_BYTES / 1024;           // 32
    constexpr int STAGE = IMG_BYTES + DZ_BYTES;          // 42 KiB
    constexpr int NPIECE = IMG_DMAS + DZ_PIECES;         // 42
    constexpr int PW = (NPIECE + 7) / 8;                 // 6 per wave (48 slots, 6 dummies)
    constexpr int OFF_SCRATCH = NB * STAGE;
    constexpr int LDS_BYTES = OFF_SCRATCH + 1024;
    static_assert(LDS_BYTES >= 8 * 4 * 256 * 4, "the final cross-wave sum reuses the stage buffers");
    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (n_items + 7) >> 3;
    const int first = xcd * per_xcd, last = min(first + per_xcd, n_items);
    const int nk = first + slot < last ? (last - (first + slot) + slots - 1) / slots : 0;
    auto item_at = [&](int k) { return (unsigned)(first + slot + k * slots); };
    const unsigned lds_base = lds_addr_of(lds);
    const int plane_elems = a.H * a.W;
    const unsigned dz_plane_bytes = (unsigned)plane_elems * 32u;

    // pieces of this wave: i*8 + wave; < IMG_DMAS: image piece, < NPIECE: dz piece, else dummy
    unsigned lane_off[PW];
    int lane_row[PW], lane_col[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int qn = i * 8 + wave;
        lane_off[i] = OOB;
        lane_row[i] = lane_col[i] = 0;
        if (qn < IMG_DMAS) {
            const int pu = qn * 64 + lane;
            if (pu < IMG_PIECES) {
                lane_col[i] = (pu % ROW_PIECES) * 4;
                const int ci = (pu / ROW_PIECES) % 3;
                lane_row[i] = pu / (ROW_PIECES * 3);
                lane_off[i] = (unsigned)((ci * plane_elems + lane_row[i] * a.W + lane_col[i]) * 4);   // from image pixel (y0-1, x0-4)
            }
        } else if (qn < NPIECE) {
            const int u = (qn - IMG_DMAS) * 64 + lane;
            const int half = u & 1, sl = (u >> 1) & 31, pl = (u >> 6) & 1, chunk = u >> 7;
            lane_row[i] = chunk >> 1;
            lane_col[i] = (chunk & 1) * 32 + stw_swz(sl);
            lane_off[i] = (unsigned)pl * dz_plane_bytes + (unsigned)((lane_row[i] * a.W + lane_col[i]) * 32) + half * 16u;   // from pixel (y0, x0)
        }
    }
    auto issue_stage = [&](const Coord& t, int buf) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)t.b * 3 * plane_elems, 0, 3 * plane_elems * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rdz =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.dz) + (size_t)t.b * 2 * dz_plane_bytes, 0, (int)(2 * dz_plane_bytes), 0x00020000);
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int qn = i * 8 + wave;
            if (qn < IMG_DMAS) {
                const int ybase = t.y0 - 1, xbase = t.x0 - 4;
                const bool ok = lane_off[i] != OOB && (unsigned)(ybase + lane_row[i]) < (unsigned)a.H && (unsigned)(xbase + lane_col[i]) < (unsigned)a.W;
                const unsigned vo = ok ? lane_off[i] + (unsigned)((ybase * a.W + xbase) * 4) : OOB;
                dma16_buf(rx, vo, 0u, lds_base + buf * STAGE + qn * 1024);
            } else if (qn < NPIECE) {
                const bool ok = t.y0 + lane_row[i] < a.H && t.x0 + lane_col[i] < a.W;
                const unsigned vo = ok ? lane_off[i] + (unsigned)((t.y0 * a.W + t.x0) * 32) : OOB;
                dma16_buf(rdz, vo, 0u, lds_base + buf * STAGE + IMG_BYTES + (qn - IMG_DMAS) * 1024);
            } else {
                dma16_buf(rx, OOB, 0u, lds_base + OFF_SCRATCH);
            }
        }
    };
    // operand addresses: dz fragments as in wgrad_bf16_kernel (pixel row 8g + 4s + q of the chunk, 8 bytes p of its 32);
    // tap-slot operand: 8 consecutive window floats from column 8g of the chunk
    int aoff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) aoff[s] = stw_swz(8 * g + 4 * s + q) * 32 + p * 8;
    int boff[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        int off = 0;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (s == n) off = tap_goff(slot_tap(nt * 16 + s));
        boff[nt] = off + XOFF + 8 * g;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    Coord cA{0, 0, 0}, cB{0, 0, 0};
    if (nk > 0) {
        cA = coord_of(a, item_at(0));
        issue_stage(cA, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
        const bool issue1 = k + 1 < nk;
        if (issue1) {   // the other buffer: read in iteration k-1, every wave has passed that iteration's barrier
            cB = coord_of(a, item_at(k + 1));
            issue_stage(cB, (k + 1) % NB);
        }
        const uint8_t* L = lds + (k % NB) * STAGE;
        const float* img = reinterpret_cast<const float*>(L);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int chunk = 2 * wave + j;   // tile row `wave`, columns 32 j ..
            const uint8_t* dzc = L + IMG_BYTES + chunk * 2048;
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dzc + mt * 1024 + aoff[0]));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dzc + mt * 1024 + aoff[1]));
                const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                __builtin_memcpy(&af[mt], v, 16);
            }
            const float* px = img + wave * 3 * IW + 32 * j;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float* s = px + boff[nt];
                const uint4 v = make_uint4(pack2bf2(f32x2{s[0], s[1]}), pack2bf2(f32x2{s[2], s[3]}), pack2bf2(f32x2{s[4], s[5]}),
                                           pack2bf2(f32x2{s[6], s[7]}));
                bfr[nt] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // stage k+1 landed (nothing younger in flight)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cA = cB;
    }
    // ---- cross-wave sum.  D of a 16x16 tile: row (filter) = 4 (lane >> 4) + reg, column (slot) = lane & 15
    float* red = reinterpret_cast<float*>(lds);   // [wave 8][mt 2][nt 2][16 x 16]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave * 2 + mt) * 2 + nt) * 256 + (4 * g + r) * 16 + n] = acc[mt][nt][r];
    __syncthreads();
    for (int e = tid; e < 1024; e += 512) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sum += red[w * 1024 + e];
        const int mt = e >> 9, nt = (e >> 8) & 1, row = (e >> 4) & 15, col = e & 15;
        const int co = mt * 16 + row, sl = nt * 16 + col;
        int kref = -1;
#pragma unroll
        for (int s = 0; s < 32; ++s)
            if (s == sl && slot_live(s)) kref = tap_kref(slot_tap(s));
        if (kref >= 0) a.slab[(size_t)blockIdx.x * 864 + co * 27 + kref] = sum;
    }
}

__global__ void stem_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int n_slabs, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 864) return;
    float s = accumulate ? dw[i] : 0.f;
    for (int k = 0; k < n_slabs; ++k) s += slab[(size_t)k * 864 + i];
    dw[i] = s;
}

static bool stem_train_setup(stemtr::Args& a, const float* x, int batch, int h, int w, long long* n_items, unsigned* grid, int wgs_per_cu) {
    using namespace stemtr;
    a.x = x;
    a.B = batch, a.H = h, a.W = w;
    a.tiles_x = (w + TW - 1) / TW;
    a.tiles_y = (h + TH - 1) / TH;
    auto magic = [](unsigned d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ULL / d); };
    a.m_tx = magic((unsigned)a.tiles_x);
    a.m_tpi = magic((unsigned)(a.tiles_x * a.tiles_y));
    *n_items = (long long)a.tiles_x * a.tiles_y * batch;
    if (*n_items <= 0 || *n_items >= 0x7fffffffLL) return false;
    const int per_xcd = (int)((*n_items + 7) / 8);
    const int cu_slots = wgs_per_cu * conv_num_cus() / 8;
    *grid = (unsigned)(8 * (per_xcd < cu_slots ? per_xcd : cu_slots));
    return true;
}

}  // namespace ay

using namespace ay;

static int stem_train_args_ok(const float* x, int batch, int h, int w, const char* who) {
    if (!(x && batch > 0 && h > 0 && w > 0 && w % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && 3LL * h * w * 4 < 0x7fffffffLL &&
          2LL * h * w * 32 < 0x7fffffffLL)) {
        set_error("%s: needs a 16-byte aligned image with W %% 4 == 0 below 2 GiB per image (%dx%d)", who, h, w);
        return AY_ERR_ARG;
    }
    return AY_OK;
}

extern "C" int ay_stem_train_fwd_bf16(const float* x_nchw, const void* w_bf16, void* z_blocked, int batch, int h, int w, ay_stream_t stream) {
    AY_CHECK_ARG(w_bf16 && z_blocked, "ay_stem_train_fwd_bf16: null");
    if (int rc = stem_train_args_ok(x_nchw, batch, h, w, "ay_stem_train_fwd_bf16")) return rc;
    stemtr::Args a{};
    long long n_items;
    unsigned grid;
    AY_CHECK_ARG(stem_train_setup(a, x_nchw, batch, h, w, &n_items, &grid, 2), "ay_stem_train_fwd_bf16: grid");   // 31 KiB of LDS: two per CU
    a.w0 = (const uint16_t*)w_bf16;
    a.z = (uint8_t*)z_blocked;
    hipLaunchKernelGGL(stem_train_fwd_kernel<false>, dim3(grid), dim3(512), 0, S(stream), a, (int)n_items);
    AY_CHECK_LAUNCH("stem_train_fwd_kernel");
    return AY_OK;
}

extern "C" size_t ay_stem_train_stats_workspace_bytes(void) { return (size_t)2 * conv_num_cus() * 64 * sizeof(float); }

extern "C" int ay_stem_train_fwd_stats_bf16(const float* x_nchw, const void* w_bf16, void* z_blocked, double* sums /* 64 doubles */, void* workspace,
                                            size_t workspace_bytes, int batch, int h, int w, ay_stream_t stream) {
    AY_CHECK_ARG(w_bf16 && z_blocked && sums && workspace, "ay_stem_train_fwd_stats_bf16: null");
    if (int rc = stem_train_args_ok(x_nchw, batch, h, w, "ay_stem_train_fwd_stats_bf16")) return rc;
    stemtr::Args a{};
    long long n_items;
    unsigned grid;
    AY_CHECK_ARG(stem_train_setup(a, x_nchw, batch, h, w, &n_items, &grid, 2), "ay_stem_train_fwd_stats_bf16: grid");
    AY_CHECK_ARG(workspace_bytes >= (size_t)grid * 64 * sizeof(float), "ay_stem_train_fwd_stats_bf16: workspace");
    a.w0 = (const uint16_t*)w_bf16;
    a.z = (uint8_t*)z_blocked;
    a.stat_slab = (float*)workspace;
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(stem_train_fwd_kernel<true>, dim3(grid), dim3(512), 0, st, a, (int)n_items);
    AY_CHECK_LAUNCH("stem_train_fwd_kernel");
    hipLaunchKernelGGL(stem_stats_reduce_kernel, dim3(1), dim3(64), 0, st, a.stat_slab, (int)grid, sums);
    AY_CHECK_LAUNCH("stem_stats_reduce_kernel");
    return AY_OK;
}

extern "C" size_t ay_stem_train_wgrad_workspace_bytes(void) { return (size_t)conv_num_cus() * 864 * sizeof(float); }

extern "C" int ay_stem_train_wgrad_bf16(const float* x_nchw, const void* dz_blocked, float* dw_oihw, int accumulate, void* workspace,
                                        size_t workspace_bytes, int batch, int h, int w, ay_stream_t stream) {
    AY_CHECK_ARG(dz_blocked && dw_oihw && workspace, "ay_stem_train_wgrad_bf16: null");
    if (int rc = stem_train_args_ok(x_nchw, batch, h, w, "ay_stem_train_wgrad_bf16")) return rc;
    stemtr::Args a{};
    long long n_items;
    unsigned grid;
    AY_CHECK_ARG(stem_train_setup(a, x_nchw, batch, h, w, &n_items, &grid, 1), "ay_stem_train_wgrad_bf16: grid");
    AY_CHECK_ARG(workspace_bytes >= (size_t)grid * 864 * sizeof(float), "ay_stem_train_wgrad_bf16: workspace %zu < %zu", workspace_bytes,
                 (size_t)grid * 864 * sizeof(float));
    a.dz = (const uint8_t*)dz_blocked;
    a.slab = (float*)workspace;
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(stem_train_wgrad_kernel, dim3(grid), dim3(512), 0, st, a, (int)n_items);
    AY_CHECK_LAUNCH("stem_train_wgrad_kernel");
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(4), dim3(256), 0, st, a.slab, dw_oihw, (int)grid, accumulate);
    AY_CHECK_LAUNCH("stem_wgrad_reduce_kernel");
    return AY_OK;
}
